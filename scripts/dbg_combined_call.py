#!/usr/bin/env python3
"""One host call of the size the call combiner produces from a reference driver at -t 64 (bsw 64 x 512 pairs, phmm 64
batches, poa 64 windows): wall time over repeats and, with GBX_HOST_TRACE=1 in the environment, the entry's own timeline.
usage: dbg_combined_call.py [bsw|phmm|poa] [units] [repeats]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from genomicsbench_amd import _native as N  # noqa: E402
from genomicsbench_amd.datagen import gen_bsw, gen_phmm, gen_poa  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "phmm"
units = int(sys.argv[2]) if len(sys.argv) > 2 else {"bsw": 32768, "phmm": 64, "poa": 64}[kind]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
N.check(N.lib().gbx_host_prepare())
if kind == "bsw":
    from genomicsbench_amd.bsw import extend_host, make_params
    b, p = gen_bsw(units, 1002), make_params()
    call = lambda: extend_host(p, b)
elif kind == "phmm":
    from genomicsbench_amd.phmm import forward_host
    N.check(N.lib().gbx_phmm_init())
    s = gen_phmm(units, 3001)
    print("pairs", s.n_pairs, "cells %.3g" % s.cells)
    call = lambda: forward_host(s)
else:
    from genomicsbench_amd.poa import consensus_host, make_params
    w, p = gen_poa(units, 4001), make_params()
    call = lambda: consensus_host(p, w)
call()
ts = []
for r in range(reps):
    if r == reps - 1:
        N.profile_begin()
    t0 = time.perf_counter()
    call()
    ts.append((time.perf_counter() - t0) * 1e3)
prof = N.profile_end(256)
ts = np.array(ts)
print("%s %d units: median %.3f ms, min %.3f, max %.3f" % (kind, units, np.median(ts), ts.min(), ts.max()))
print("kernels of the last call (ms, launches):", {k: (round(v[0], 3), v[1]) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])})
