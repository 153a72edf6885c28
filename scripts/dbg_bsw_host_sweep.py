"""gbx_bsw_extend_host on the 'large' shard under a few settings of the host pipeline: best of 6 calls each.
usage: python3 scripts/dbg_bsw_host_sweep.py"""
import os, sys, time, subprocess
sys.path.insert(0, ".")
if len(sys.argv) > 1:
    import numpy as np
    from genomicsbench_amd import _native as N
    from genomicsbench_amd.bsw import extend_host, make_params
    from genomicsbench_amd.datagen import gen_bsw
    b = gen_bsw(2_000_000, 1002); p = make_params()
    N.check(N.lib().gbx_host_prepare())
    out = np.full((b.n, 6), -1, dtype=np.int32)
    ms = []
    for k in range(7):
        t = time.perf_counter(); extend_host(p, b, out); ms.append((time.perf_counter() - t) * 1e3)
    print("%-60s best %.2f ms  median %.2f  (%.0f GCUPS)" % (sys.argv[1], min(ms[1:]), sorted(ms[1:])[3], b.nominal_cells / min(ms[1:]) / 1e6), flush=True)
else:
    for env in ({}, {"GBX_HOST_THREADS": "8"}, {"GBX_HOST_THREADS": "4"}, {"GBX_BSW_HOST_CHUNK": "500000"}, {"GBX_BSW_HOST_CHUNK": "1000000"},
                {"GBX_HOST_THREADS": "8", "GBX_BSW_HOST_CHUNK": "500000"}, {"GBX_BSW_PACK": "0"}, {"GBX_BSW_PACK": "0", "GBX_HOST_THREADS": "8"}):
        e = dict(os.environ); e.update(env)
        subprocess.run([sys.executable, __file__, " ".join("%s=%s" % kv for kv in env.items()) or "default"], env=e)
