#!/usr/bin/env python3
"""Concurrency stress on the GPU: T host threads, each in a loop calling a randomly chosen kernel's host entry on a job of
random size (fmi included, multi-device splitting on some calls) and comparing with results computed one at a time beforehand.
The fuzz run is one job at a time; this one is about what the entries share - side streams and their events, lanes, the fmi
index cache, the allocator.  usage: stress_concurrent.py [seconds] [threads] [seed]"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from genomicsbench_amd import _native as N  # noqa: E402
from genomicsbench_amd.abea import align_host  # noqa: E402
from genomicsbench_amd.bsw import extend_host, make_params as bsw_params  # noqa: E402
from genomicsbench_amd.chain import chain_host  # noqa: E402
from genomicsbench_amd.datagen import gen_abea, gen_bsw, gen_chain, gen_fmi_genome, gen_fmi_reads, gen_phmm, gen_poa  # noqa: E402
from genomicsbench_amd.fmi import build_index, default_params as fmi_params, smem_host  # noqa: E402
from genomicsbench_amd.phmm import forward_host  # noqa: E402
from genomicsbench_amd.poa import consensus_host, make_params as poa_params  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
T = int(sys.argv[2]) if len(sys.argv) > 2 else 6
rng = np.random.default_rng(int(sys.argv[3]) if len(sys.argv) > 3 else 5)
N.check(N.lib().gbx_host_prepare())
pb, pp = bsw_params(), poa_params()
g = gen_fmi_genome(300000, 99)
idx = build_index(g)
P = fmi_params(19)


def same(a, b):
    if isinstance(a, (list, tuple)):
        return len(a) == len(b) and all(same(x, y) for x, y in zip(a, b))
    if isinstance(a, np.ndarray):
        if a.dtype.names:
            return all(np.array_equal(a[f], b[f]) for f in a.dtype.names if not f.startswith("pad") and not f.startswith("_"))
        return np.array_equal(a, b)
    return a == b


def make_jobs():
    jobs = []
    for n in (300000, 40000, 3000, 64):
        b = gen_bsw(n, int(rng.integers(1, 1 << 20)))
        jobs.append(("bsw n=%d" % n, lambda b=b: extend_host(pb, b)))
    for n in (300, 40):
        c = gen_chain(n, int(rng.integers(1, 1 << 20)), realistic=bool(n == 40))
        jobs.append(("chain calls=%d" % n, lambda c=c: chain_host(*c)))
    for n in (40, 4):
        s = gen_phmm(n, int(rng.integers(1, 1 << 20)))
        jobs.append(("phmm batches=%d" % n, lambda s=s: forward_host(s)))
    for n in (40, 3):
        w = gen_poa(n, int(rng.integers(1, 1 << 20)))
        jobs.append(("poa windows=%d" % n, lambda w=w: consensus_host(pp, w)))
    for n in (150, 8):
        r = gen_abea(n, int(rng.integers(1, 90000)))
        jobs.append(("abea reads=%d" % n, lambda r=r: (lambda o, k: (r.split_pairs(o, k), k))(*align_host(r))))
    for n in (6000, 200):
        rs = gen_fmi_reads(g, n, int(rng.integers(1, 1 << 20)), read_len=151)
        jobs.append(("fmi reads=%d" % n, lambda rs=rs, n=n: smem_host(idx, rs, P, out_cap=400 * n)))
    return jobs


jobs = make_jobs()
want = [f() for _, f in jobs]
print("stress: %d job kinds, %d threads, %.0f s" % (len(jobs), T, budget), flush=True)
t_end = time.time() + budget
bad = []
count = [0] * T


def work(t):
    r = np.random.default_rng(1000 + t)
    while time.time() < t_end and not bad:
        k = int(r.integers(0, len(jobs)))
        got = jobs[k][1]()
        if not same(got, want[k]):
            bad.append("thread %d: %s differs when run beside the others" % (t, jobs[k][0]))
        count[t] += 1


th = [threading.Thread(target=work, args=(t,)) for t in range(T)]
for x in th:
    x.start()
for x in th:
    x.join()
if bad:
    print("MISMATCH", bad[0], flush=True)
    sys.exit(1)
print("stress ok: %d calls from %d threads" % (sum(count), T), flush=True)
