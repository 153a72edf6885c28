"""Do the occasional 5-8 ms stalls of gbx_bsw_extend_host coincide with the container's CPU throttling?  cpu.stat of the cgroup
(nr_throttled, throttled_usec) is read around every call; the slow calls are listed with what the counters did across them.
usage: dbg_bsw_host_stalls.py [calls]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from genomicsbench_amd import _native as N
from genomicsbench_amd.bsw import extend_host, make_params
from genomicsbench_amd.datagen import gen_bsw


def stat():
    d = {}
    try:
        for ln in open("/sys/fs/cgroup/cpu.stat"):
            k, v = ln.split()
            d[k] = int(v)
    except OSError:
        pass
    return d.get("nr_throttled", 0), d.get("throttled_usec", 0), d.get("nr_periods", 0)


b = gen_bsw(2_000_000, 1002); p = make_params()
N.check(N.lib().gbx_host_prepare())
out = np.full((b.n, 6), -1, dtype=np.int32)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
time.sleep(0.5)                                   # let the generator's threads' throttling drain
rows = []
for k in range(n + 3):
    s0 = stat(); t = time.perf_counter(); extend_host(p, b, out); ms = (time.perf_counter() - t) * 1e3; s1 = stat()
    rows.append((ms, s1[0] - s0[0], s1[1] - s0[1], s1[2] - s0[2]))
rows = rows[3:]
ms = np.array([r[0] for r in rows]); thr = np.array([r[1] for r in rows])
print("calls %d  median %.2f ms; slow (> 12 ms): %d, of which in a throttled period: %d; throttled periods seen during fast calls: %d of %d" % (
    n, np.median(ms), int((ms > 12).sum()), int(((ms > 12) & (thr > 0)).sum()), int(((ms <= 12) & (thr > 0)).sum()), int((ms <= 12).sum())))
for r in rows:
    if r[0] > 12:
        print("  slow call %.2f ms: nr_throttled +%d, throttled_usec +%d, periods +%d" % r)
