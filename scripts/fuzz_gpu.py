#!/usr/bin/env python3
"""Randomised parity run on the GPU: host entries of the five kernels against the oracle on random jobs of random
sizes (small-job modes, class modes, staged and packed transfers all get hit).  usage: fuzz_gpu.py [seconds] [seed]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np  # noqa: E402

from cases import adversarial_bsw  # noqa: E402
from genomicsbench_amd import _native as N  # noqa: E402
from genomicsbench_amd.bsw import extend_host, fill_scmat, make_params as bsw_params  # noqa: E402
from genomicsbench_amd.chain import chain_host  # noqa: E402
from genomicsbench_amd.abea import align_host  # noqa: E402
from genomicsbench_amd.datagen import gen_abea, gen_bsw, gen_chain, gen_phmm, gen_poa  # noqa: E402
from genomicsbench_amd.phmm import forward_host  # noqa: E402
from genomicsbench_amd.poa import consensus_host, make_params as poa_params  # noqa: E402
from oracle import oracle_py as O  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
N.check(N.lib().gbx_host_prepare())
t_end = time.time() + budget
count = {"bsw": 0, "chain": 0, "phmm": 0, "poa": 0, "abea": 0}
while time.time() < t_end:
    k = rng.choice(["bsw", "bsw", "chain", "phmm", "poa", "abea"])
    seed = int(rng.integers(1, 1 << 30))
    if k == "bsw":
        n = int(rng.choice([1, 7, 64, 513, 4000, 16384, 16385, 40000, 260000]))
        adv = rng.random() < 0.4 and n <= 40000
        b = adversarial_bsw(n, seed) if adv else gen_bsw(n, seed)
        kw = {} if rng.random() < 0.6 else dict(o_del=int(rng.integers(0, 9)), e_del=int(rng.integers(1, 4)), o_ins=int(rng.integers(0, 9)),
                                                  e_ins=int(rng.integers(1, 4)), zdrop=int(rng.choice([0, 20, 100])), w=int(rng.choice([3, 30, 100, 400])),
                                                  mat=fill_scmat(int(rng.integers(1, 4)), int(rng.integers(1, 6)), -int(rng.integers(0, 3))))
        p = bsw_params(**kw)
        ok = np.array_equal(extend_host(p, b), O.bsw_oracle(p, b, 8))
        what = "n=%d adversarial=%s kw=%s" % (n, adv, kw)
    elif k == "chain":
        nc = int(rng.choice([1, 3, 40, 300]))
        case = gen_chain(nc, seed)
        got, want = chain_host(*case), O.chain_oracle(*case, nthreads=8)
        ok = all(np.array_equal(g, w) for g, w in zip(got, want))
        what = "calls=%d" % nc
    elif k == "phmm":
        nb = int(rng.choice([1, 2, 9, 40, 90]))
        bs = gen_phmm(nb, seed)
        want, _ = O.phmm_oracle(bs, 8, True)
        got = forward_host(bs)
        ok = bool(np.all(np.abs(got - want) <= 1e-5 * np.maximum(1, np.abs(want)) + 5e-7))
        what = "batches=%d pairs=%d" % (nb, bs.n_pairs)
    elif k == "abea":
        nr = int(rng.choice([1, 3, 24, 150, 700]))             # 700 reads: the staged transfers and the packed download
        rs = gen_abea(nr, seed % 100000, first=int(rng.integers(0, 5000)))
        if rng.random() < 0.3:                                   # push some reads out of the fast-division range
            ev = rs.event_mean
            for r in rng.choice(nr, size=max(1, nr // 4), replace=False):
                a, b = int(rs.event_off[r]), int(rs.event_off[r + 1])
                ev[a + int(rng.integers(0, b - a))] = np.float32(rng.choice([1e-30, 3e13, 0.0]))
        (go, gn), (wo, wn) = align_host(rs), O.abea_oracle(rs, 16)
        ok = np.array_equal(gn, wn) and all(np.array_equal(g, w) for g, w in zip(rs.split_pairs(go, gn), rs.split_pairs(wo, wn)))
        what = "reads=%d" % nr
    else:
        nw = int(rng.choice([1, 5, 40]))
        ws = gen_poa(nw, seed)
        pp = poa_params()
        ok = consensus_host(pp, ws) == O.poa_oracle(pp, ws, 8)
        what = "windows=%d" % nw
    count[k] += 1
    if not ok:
        print("MISMATCH %s seed=%d %s" % (k, seed, what), flush=True)
        sys.exit(1)
print("fuzz ok:", count, flush=True)
