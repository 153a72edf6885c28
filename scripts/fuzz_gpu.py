#!/usr/bin/env python3
"""Randomised parity run on the GPU: host entries of the six kernels against the oracle on random jobs of random
sizes (small-job modes, class modes, staged and packed transfers all get hit).  usage: fuzz_gpu.py [seconds] [seed]
FUZZ_LOG=<file>: every job's description is appended (and synced) before it runs - after a hang the last line is the culprit;
FUZZ_SKIP_UNTIL=<n>: the first n-1 jobs are drawn but not run (replays the random stream up to a job of interest);
FUZZ_KINDS=phmm,combo: only these job kinds (a run aimed at one kernel after a change to it)."""
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
from genomicsbench_amd.datagen import gen_abea, gen_bsw, gen_chain, gen_fmi_genome, gen_fmi_reads, gen_phmm, gen_poa  # noqa: E402
from genomicsbench_amd.fmi import FmiReadSet, build_index, default_params as fmi_params, smem_host  # noqa: E402
from genomicsbench_amd.phmm import forward_host  # noqa: E402
from genomicsbench_amd.poa import consensus_host, make_params as poa_params  # noqa: E402
from oracle import oracle_py as O  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
N.check(N.lib().gbx_host_prepare())
t_end = time.time() + budget
count = {"bsw": 0, "chain": 0, "phmm": 0, "poa": 0, "abea": 0, "fmi": 0, "combo": 0}
skip_until = int(os.environ.get("FUZZ_SKIP_UNTIL", "0"))       # replay aid: draw the first jobs without running them
job_no = 0


def announce(text):
    """the job about to run, flushed before it runs: a hang leaves its description as the last line (FUZZ_LOG=file)"""
    path = os.environ.get("FUZZ_LOG")
    if path:
        with open(path, "a") as fh:
            fh.write("job %d %s\n" % (job_no, text))
            fh.flush()
            os.fsync(fh.fileno())
    return job_no >= skip_until
fmi_idx = {}                                                    # genome length -> (genome, index): built once per size
n_multi = 0
kinds = os.environ["FUZZ_KINDS"].split(",") if os.environ.get("FUZZ_KINDS") else ["bsw", "bsw", "chain", "phmm", "poa", "poa", "abea", "fmi", "combo"]
while time.time() < t_end:
    k = rng.choice(kinds)
    seed = int(rng.integers(1, 1 << 30))
    job_no += 1
    ok, what = True, ""
    # a third of the jobs through the multi-device layer of the host entries: 2-4 logical devices on this GPU, cut however
    # small the job is (half of those) or only when it is large enough by the kernel's own threshold
    ndev = int(rng.choice([2, 3, 4])) if rng.random() < 0.33 else 0
    for v in ("GBX_DEVICE_MAP", "GBX_SHARD_MIN_UNITS"):
        os.environ.pop(v, None)
    if ndev:
        os.environ["GBX_DEVICE_MAP"] = ",".join(["0"] * ndev)
        if rng.random() < 0.5:
            os.environ["GBX_SHARD_MIN_UNITS"] = "1"
        n_multi += 1
    N.check(N.lib().gbx_host_set_devices(ndev))
    if k == "bsw":
        n = int(rng.choice([1, 7, 64, 513, 4000, 16384, 16385, 40000, 260000]))
        adv = rng.random() < 0.4 and n <= 40000
        b = adversarial_bsw(n, seed) if adv else gen_bsw(n, seed)
        kw = {} if rng.random() < 0.6 else dict(o_del=int(rng.integers(0, 9)), e_del=int(rng.integers(1, 4)), o_ins=int(rng.integers(0, 9)),
                                                  e_ins=int(rng.integers(1, 4)), zdrop=int(rng.choice([0, 20, 100])), w=int(rng.choice([3, 30, 100, 400])),
                                                  mat=fill_scmat(int(rng.integers(1, 4)), int(rng.integers(1, 6)), -int(rng.integers(0, 3))))
        p = bsw_params(**kw)
        lane = rng.random() < 0.5                                # the lane kernels (large jobs take them by default) on this job too
        # a third of the jobs as a pipelined call: staged uploads, two to a dozen chunks (preparing passes on their own streams,
        # chunks with and without row-kernel pairs, packed bases read by the lane kernels)
        piped = n >= 513 and rng.random() < 0.35
        chunk = 64 * int(rng.integers(max(1, n // 768), max(2, n // 128))) if piped else 0
        what = "n=%d adversarial=%s lane=%s chunk=%d kw=%s" % (n, adv, lane, chunk, kw)
        run = announce("bsw seed=%d devices=%d min_units=%s %s" % (seed, ndev, os.environ.get("GBX_SHARD_MIN_UNITS"), what))
        if lane:
            os.environ["GBX_BSW_LANE"] = "1"
        if piped:
            os.environ["GBX_BSW_HOST_CHUNK"] = str(chunk)
            os.environ["GBX_HOST_STAGE_MIN"] = "0"
        try:
            ok = not run or np.array_equal(extend_host(p, b), O.bsw_oracle(p, b, 8))
        finally:
            for v in ("GBX_BSW_LANE", "GBX_BSW_HOST_CHUNK", "GBX_HOST_STAGE_MIN"):
                os.environ.pop(v, None)
    elif k == "chain":
        nc = int(rng.choice([1, 3, 40, 300]))
        real = bool(rng.random() < 0.5)                          # minimap2's strand / reference structure: calls cut into jobs
        case = gen_chain(nc, seed, realistic=real)
        if rng.random() < 0.3:                                   # several segment ids inside the calls
            case[2][:] |= (rng.integers(0, 2, len(case[2])).astype(np.uint64) << np.uint64(48))
            case[3]["n_segs"] = 2
            case[3]["max_dist_y"] = 800
        what = "calls=%d realistic=%s n_segs=%d" % (nc, real, int(case[3]["n_segs"][0]))
        if announce("chain seed=%d devices=%d min_units=%s %s" % (seed, ndev, os.environ.get("GBX_SHARD_MIN_UNITS"), what)):
            got, want = chain_host(*case), O.chain_oracle(*case, nthreads=8)
            ok = all(np.array_equal(g, w) for g, w in zip(got, want))
    elif k == "fmi":
        glen = int(rng.choice([3000, 200000, 3000000]))
        if glen not in fmi_idx:
            g = gen_fmi_genome(glen, 4242 + glen)
            fmi_idx[glen] = (g, build_index(g))
        g, idx = fmi_idx[glen]
        nr = int(rng.choice([1, 17, 500, 6000]))
        rl = int(rng.choice([30, 76, 151, 250, 400, 2500, 8100, 9000]))        # the last two: read in place, not staged in LDS
        if rl > 1000:
            nr = min(nr, 17)
        rs = gen_fmi_reads(g, nr, seed, read_len=min(rl, glen // 4))
        if rng.random() < 0.3:                                   # ragged lengths
            keep = rng.integers(1, rs.read_len[0] + 1, nr).astype(np.int32)
            rs = FmiReadSet(rs.enc, rs.read_off, keep)
        P = fmi_params(int(rng.choice([8, 12, 19, 25])))
        if rng.random() < 0.3:
            P.split_width, P.max_mem_intv = int(rng.integers(1, 30)), int(rng.integers(0, 60))
        os.environ["GBX_FMI_WIDE"] = "1" if rng.random() < 0.3 else "0"     # the 64-bit instance too
        what = "genome=%d reads=%d len=%d minseed=%d wide=%s split_width=%d max_mem_intv=%d ragged=%s" % (
            glen, nr, rl, P.min_seed_len, os.environ["GBX_FMI_WIDE"], P.split_width, P.max_mem_intv, bool((rs.read_len != rs.read_len[0]).any()))
        if announce("fmi seed=%d devices=%d min_units=%s %s" % (seed, ndev, os.environ.get("GBX_SHARD_MIN_UNITS"), what)):
            (go, goff), (wo, woff) = smem_host(idx, rs, P, out_cap=max(64, 200 * nr, nr * (rl + 64))), O.fmi_oracle(idx, rs, P, nthreads=8)
            ok = np.array_equal(goff, woff) and all(np.array_equal(go[f], wo[f]) for f in ("rid", "m", "n", "k", "l", "s"))
    elif k == "phmm":
        nb = int(rng.choice([1, 2, 9, 40, 90]))
        bs = gen_phmm(nb, seed)
        odd = rng.random() < 0.35
        if odd:                                                  # bytes the prior tables do not code (lower case, IUPAC), and N: literal compares
            for arr in (bs.hap, bs.rs):
                nmut = max(1, len(arr) // int(rng.choice([50, 400, 5000])))
                at = rng.integers(0, max(1, len(arr) - 8), nmut)
                arr[at] = rng.choice(np.frombuffer(b"NNacgtRYn", dtype=np.uint8), nmut)
        what = "batches=%d pairs=%d odd_bytes=%s" % (nb, bs.n_pairs, odd)
        if announce("phmm seed=%d devices=%d min_units=%s %s" % (seed, ndev, os.environ.get("GBX_SHARD_MIN_UNITS"), what)):
            want, _ = O.phmm_oracle(bs, 8, True)
            got = forward_host(bs)
            ok = bool(np.all(np.abs(got - want) <= 1e-5 * np.maximum(1, np.abs(want)) + 5e-7))
    elif k == "abea":
        nr = int(rng.choice([1, 3, 24, 150, 700]))             # 700 reads: the staged transfers and the packed download
        first = int(rng.integers(0, 5000))
        rs = gen_abea(nr, seed % 100000, first=first)
        extreme = rng.random() < 0.3
        if extreme:                                              # push some reads out of the fast-division range
            ev = rs.event_mean
            for r in rng.choice(nr, size=max(1, nr // 4), replace=False):
                a, b = int(rs.event_off[r]), int(rs.event_off[r + 1])
                ev[a + int(rng.integers(0, b - a))] = np.float32(rng.choice([1e-30, 3e13, 0.0]))
        what = "reads=%d first=%d extreme_events=%s" % (nr, first, extreme)
        if announce("abea seed=%d devices=%d min_units=%s %s" % (seed, ndev, os.environ.get("GBX_SHARD_MIN_UNITS"), what)):
            (go, gn), (wo, wn) = align_host(rs), O.abea_oracle(rs, 16)
            ok = np.array_equal(gn, wn) and all(np.array_equal(g, w) for g, w in zip(rs.split_pairs(go, gn), rs.split_pairs(wo, wn)))
    elif k == "combo":
        # round 6: small calls of one kernel from several threads at once - the host entries combine them (csrc/host_combine.h);
        # every caller must get the results of its own call
        import threading
        kind = str(rng.choice(["bsw", "phmm", "poa"]))
        T = int(rng.choice([2, 3, 8, 16]))
        rounds = int(rng.choice([1, 3]))
        if kind == "bsw":
            jobs = [gen_bsw(int(rng.choice([1, 64, 512, 3000])), seed + t) for t in range(T)]
            prm = [bsw_params() if rng.random() < 0.7 else bsw_params(zdrop=20, w=30) for _ in range(T)]
            call = lambda t: extend_host(prm[t], jobs[t])
            want = [O.bsw_oracle(prm[t], jobs[t], 4) for t in range(T)]
            same_as = lambda g, w: np.array_equal(g, w)
        elif kind == "phmm":
            jobs = [gen_phmm(int(rng.choice([1, 2, 5])), seed + t) for t in range(T)]
            call = lambda t: forward_host(jobs[t])
            want = [O.phmm_oracle(jobs[t], 4) for t in range(T)]
            same_as = lambda g, w: bool(np.all(np.abs(g - w) <= 1e-5 * np.maximum(1, np.abs(w)) + 5e-7))
        else:
            pps = [poa_params() if rng.random() < 0.7 else poa_params(o1=0, e1=2) for _ in range(T)]
            jobs = [gen_poa(int(rng.choice([1, 2])), seed + t) for t in range(T)]
            call = lambda t: consensus_host(pps[t], jobs[t])
            want = [O.poa_oracle(pps[t], jobs[t], 4) for t in range(T)]
            same_as = lambda g, w: g == w
        what = "combo kind=%s threads=%d rounds=%d" % (kind, T, rounds)
        if announce("combo seed=%d devices=%d min_units=%s %s" % (seed, ndev, os.environ.get("GBX_SHARD_MIN_UNITS"), what)):
            for _ in range(rounds):
                got = [None] * T

                def work(t):
                    try:
                        got[t] = call(t)
                    except Exception as e:      # noqa: BLE001
                        got[t] = e
                th = [threading.Thread(target=work, args=(t,)) for t in range(T)]
                for x in th:
                    x.start()
                for x in th:
                    x.join()
                ok = ok and all(not isinstance(g, Exception) and same_as(g, w) for g, w in zip(got, want))
    else:
        nw = int(rng.choice([1, 5, 40]))
        ws = gen_poa(nw, seed)
        # round 6: spoa's three gap subtypes (linear when g >= e), and the int32 wide path forced onto ordinary windows
        sub = rng.random()
        pp = poa_params() if sub < 0.5 else poa_params(o1=0, e1=int(rng.choice([1, 2, 4]))) if sub < 0.75 else poa_params(o2=4, e2=2)
        if rng.random() < 0.2 and nw <= 5:
            os.environ["GBX_POA_FORCE_WIDE"] = "1"
            if rng.random() < 0.5:
                os.environ["GBX_POA_WIDE_SLOTS"] = "2"
        lock = rng.random() < 0.4                                # the lock-step form (default: the window kernel with the row ring)
        if lock:
            os.environ["GBX_POA_LOCKSTEP"] = "1"
            os.environ["GBX_POA_TB_SERIAL"] = str(int(rng.integers(0, 2)))
            os.environ["GBX_POA_DP_OCC"] = str(int(rng.choice([5, 6])))
        elif rng.random() < 0.4:                                 # (default: the team kernel; here one wavefront per window, or a mix)
            os.environ[str(rng.choice(["GBX_POA_TEAM", "GBX_POA_TEAM_MAX"]))] = "0"
        what = "windows=%d lockstep=%s tb_serial=%s dp_occ=%s team=%s team_max=%s scores=(g %d e %d q %d c %d) wide=%s" % (
            nw, lock, os.environ.get("GBX_POA_TB_SERIAL"), os.environ.get("GBX_POA_DP_OCC"), os.environ.get("GBX_POA_TEAM"), os.environ.get("GBX_POA_TEAM_MAX"),
            pp.g, pp.e, pp.q, pp.c, os.environ.get("GBX_POA_FORCE_WIDE"))
        run = announce("poa seed=%d devices=%d min_units=%s %s" % (seed, ndev, os.environ.get("GBX_SHARD_MIN_UNITS"), what))
        try:
            ok = not run or consensus_host(pp, ws) == O.poa_oracle(pp, ws, 8)
        finally:
            for v in ("GBX_POA_LOCKSTEP", "GBX_POA_TB_SERIAL", "GBX_POA_DP_OCC", "GBX_POA_TEAM", "GBX_POA_TEAM_MAX", "GBX_POA_FORCE_WIDE", "GBX_POA_WIDE_SLOTS"):
                os.environ.pop(v, None)
    what += " devices=%d min_units=%s" % (ndev, os.environ.get("GBX_SHARD_MIN_UNITS"))
    count[k] += 1
    if not ok:
        print("MISMATCH %s seed=%d %s" % (k, seed, what), flush=True)
        sys.exit(1)
print("fuzz ok:", count, "of which through 2-4 logical devices:", n_multi, flush=True)
