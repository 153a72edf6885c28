"""gbx_phmm_forward_host on 'large' with a pause in front of every call (the GPU's clocks come down when it idles): python3 dbg_phmm_host_cold.py [pause_s]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from genomicsbench_amd import _native as N
from genomicsbench_amd.datagen import gen_phmm
from genomicsbench_amd.phmm import forward_host
N.check(N.lib().gbx_host_prepare())
b = gen_phmm(20000, 3001)
pause = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
forward_host(b)
ms = []
for _ in range(int(os.environ.get("CALLS", "5"))):
    time.sleep(pause)
    t = time.perf_counter(); forward_host(b); ms.append((time.perf_counter() - t) * 1e3)
print("pause %.1f s before every call: %s ms | env %s" % (pause, " ".join("%.1f" % m for m in ms), {k: v for k, v in os.environ.items() if k.startswith("GBX_")}))
