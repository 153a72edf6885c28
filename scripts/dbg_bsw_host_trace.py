"""Timeline (GBX_HOST_TRACE) of gbx_bsw_extend_host on the 'large' shard for a few upload-thread counts.
usage: python3 scripts/dbg_bsw_host_trace.py"""
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
    os.environ.pop("GBX_HOST_TRACE", None)
    ms = []
    for k in range(6):
        t = time.perf_counter(); extend_host(p, b, out); ms.append((time.perf_counter() - t) * 1e3)
    print("== %-50s best %.2f ms  median %.2f" % (sys.argv[1], min(ms[1:]), sorted(ms[1:])[2]), flush=True)
    os.environ["GBX_HOST_TRACE"] = "1"
    extend_host(p, b, out)
else:
    for env in ({}, {"GBX_HOST_THREADS": "2"}, {"GBX_HOST_THREADS": "4"}, {"GBX_HOST_THREADS": "8"}, {"GBX_BSW_PACK": "0"}, {"GBX_BSW_PACK": "0", "GBX_HOST_THREADS": "8"}):
        e = dict(os.environ); e.update(env)
        subprocess.run([sys.executable, __file__, " ".join("%s=%s" % kv for kv in env.items()) or "default"], env=e)
