"""gbx_bsw_extend_host on the 'large' shard, many calls: the distribution of the call time (median, quartiles, the share of
outliers) - single calls on the pool's boxes stall now and then for 5-8 ms.  usage: dbg_bsw_host_many.py [calls]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from genomicsbench_amd import _native as N
from genomicsbench_amd.bsw import extend_host, make_params
from genomicsbench_amd.datagen import gen_bsw
b = gen_bsw(2_000_000, 1002); p = make_params()
N.check(N.lib().gbx_host_prepare())
out = np.full((b.n, 6), -1, dtype=np.int32)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
ms = []
for k in range(n + 3):
    t = time.perf_counter(); extend_host(p, b, out); ms.append((time.perf_counter() - t) * 1e3)
ms = np.sort(np.array(ms[3:]))
print("calls %d  min %.2f  q1 %.2f  median %.2f  q3 %.2f  max %.2f  over 12 ms: %d | env %s" % (
    n, ms[0], ms[n // 4], ms[n // 2], ms[3 * n // 4], ms[-1], int((ms > 12).sum()),
    {k: v for k, v in os.environ.items() if k.startswith(("GBX_", "GPU_MAX"))}), flush=True)
