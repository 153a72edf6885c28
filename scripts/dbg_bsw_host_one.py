"""gbx_bsw_extend_host on the 'large' shard, a few calls, the last one with the GBX_HOST_TRACE timeline; meant to run under
`rocprofv3 --kernel-trace --memory-copy-trace` so that every DMA and kernel of the call has a start and an end.
usage: python3 scripts/dbg_bsw_host_one.py [calls]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from genomicsbench_amd import _native as N
from genomicsbench_amd.bsw import extend_host, make_params
from genomicsbench_amd.datagen import gen_bsw
b = gen_bsw(2_000_000, 1002); p = make_params()
N.check(N.lib().gbx_host_prepare())
out = np.full((b.n, 6), -1, dtype=np.int32)
ms = []
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    t = time.perf_counter(); extend_host(p, b, out); ms.append((time.perf_counter() - t) * 1e3)
print("calls ms:", " ".join("%.2f" % m for m in ms), "| env", {k: v for k, v in os.environ.items() if k.startswith(("GBX_", "GPU_MAX"))}, flush=True)
os.environ["GBX_HOST_TRACE"] = "1"
extend_host(p, b, out)
