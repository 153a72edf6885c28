"""gbx_bsw_extend_host on the 'large' shard under several builds of the library (GBX_LIB) / settings, alternating: best and median
of 6 calls each.  usage: python3 scripts/ab_bsw_host.py <reps> "<env assignments>" ...   ("-" = none)"""
import os, sys, time, subprocess
sys.path.insert(0, ".")
if sys.argv[1] == "--one":
    import numpy as np
    from genomicsbench_amd import _native as N
    from genomicsbench_amd.bsw import extend_host, make_params
    from genomicsbench_amd.datagen import gen_bsw
    from oracle import oracle_py as O
    b = gen_bsw(2_000_000, 1002); p = make_params()
    N.check(N.lib().gbx_host_prepare())
    out = np.full((b.n, 6), -1, dtype=np.int32)
    ms = []
    for k in range(7):
        t = time.perf_counter(); extend_host(p, b, out); ms.append((time.perf_counter() - t) * 1e3)
    print("== %-60s best %.2f ms  median %.2f  (%.0f GCUPS) checksum %d" % (sys.argv[2], min(ms[1:]), sorted(ms[1:])[3], b.nominal_cells / min(ms[1:]) / 1e6,
                                                                          int(out.astype(np.int64).sum())), flush=True)
else:
    for rep in range(int(sys.argv[1])):
        for spec in sys.argv[2:]:
            e = dict(os.environ)
            if spec != "-":
                for kv in spec.split():
                    k, v = kv.split("=", 1); e[k] = v if k != "GBX_LIB" else os.path.abspath(v)
            subprocess.run([sys.executable, __file__, "--one", spec], env=e)
