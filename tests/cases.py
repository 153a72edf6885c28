"""Seeded test-case builders shared by the CPU and GPU tests."""
import numpy as np

from genomicsbench_amd.bsw import BswBatch


def adversarial_bsw(n, seed, max_q=250, max_t=2000, n_frac=0.02):
    """h0 in 0..200, 0-30 % substitutions, indels up to 12, diverged tails, N bases, ragged lengths."""
    rng = np.random.default_rng(seed)
    ts, qs, h0 = [], [], []
    for _ in range(n):
        ql = int(rng.integers(1, max_q + 1))
        q = rng.integers(0, 4, ql).astype(np.uint8)
        sub = rng.random() * 0.3
        t = []
        j = 0
        cut = int(rng.integers(0, ql + 1)) if rng.random() < 0.3 else ql     # diverge after `cut`
        while j < cut:
            u = rng.random()
            if u < 0.02:
                j += int(rng.integers(1, 13))                                  # deletion from target
                continue
            b = q[j] if rng.random() >= sub else rng.integers(0, 4)
            t.append(b)
            if rng.random() < 0.02:
                t.extend(rng.integers(0, 4, int(rng.integers(1, 13))))         # insertion
            j += 1
        tl = int(rng.integers(1, max_t + 1)) if rng.random() < 0.2 else min(max_t, len(t) + int(rng.integers(0, 220)))
        tl = max(tl, 1)
        t = np.array(t[:tl], dtype=np.uint8)
        if len(t) < tl:
            t = np.concatenate([t, rng.integers(0, 4, tl - len(t)).astype(np.uint8)])
        if rng.random() < 0.3:
            q[rng.random(ql) < n_frac] = 4
            t[rng.random(tl) < n_frac] = 4
        ts.append(t)
        qs.append(q)
        h0.append(int(rng.integers(0, 201)))
    return BswBatch.from_sequences(ts, qs, np.array(h0, dtype=np.int32))


def edge_bsw():
    """Hand-made edge cases: 1x1, all-match, all-mismatch, all-N, h0=0, zero-length, exact class boundaries."""
    ts, qs, h0 = [], [], []
    A = lambda *x: np.array(x, dtype=np.uint8)
    def add(t, q, h):
        ts.append(np.asarray(t, dtype=np.uint8)); qs.append(np.asarray(q, dtype=np.uint8)); h0.append(h)
    add(A(0), A(0), 10)
    add(A(0), A(1), 10)
    add(A(4), A(4), 3)
    add(A(0, 1, 2, 3), A(0, 1, 2, 3), 0)
    add(A(0, 1, 2, 3), A(0, 1, 2, 3), 1)
    add(np.zeros(0), A(0, 1), 7)            # tlen == 0
    add(A(0, 1), np.zeros(0), 7)            # qlen == 0
    add(np.zeros(0), np.zeros(0), 7)
    rng = np.random.default_rng(7)
    for ql in (1, 2, 15, 16, 17, 31, 32, 33, 63, 64, 65, 127, 128, 129, 255, 256, 257, 300, 1023, 1024, 1025, 1500):
        q = rng.integers(0, 4, ql).astype(np.uint8)
        for mode in range(4):
            if mode == 0:
                t = np.concatenate([q, rng.integers(0, 4, 50).astype(np.uint8)])
            elif mode == 1:
                t = (q + 1) % 4                                  # all mismatch
            elif mode == 2:
                t = np.concatenate([q[: ql // 2], rng.integers(0, 4, 7).astype(np.uint8), q[ql // 2:]])
            else:
                t = np.concatenate([q[: ql // 3], q[ql // 3 + min(5, ql // 3):], rng.integers(0, 4, 30).astype(np.uint8)])
            add(t, q, int(rng.integers(0, 150)))
    add(np.full(500, 4), np.full(200, 4), 50)       # all N
    add(np.zeros(2047), np.zeros(255), 200)         # driver maximum sizes, homopolymer
    add(np.zeros(300), np.zeros(255), 0)
    return BswBatch.from_sequences(ts, qs, np.array(h0, dtype=np.int32))


# ------------------------------------------------------------------------- chain
def _chain_call(rng, n, n_segs=1, dense=False, avg_qspan=15.0, dup_x=False, max_dist=5000, bw=500):
    """One synthetic minimap2 chaining call: anchors sorted by x; y = seg<<48 | span<<32 | qpos."""
    ndiag = int(rng.integers(1, 5))
    xs, qs = [], []
    per = max(1, (n * 4 // 5) // ndiag)
    for d in range(ndiag):
        x = int(rng.integers(1000, 3000)) + d * 7919
        y = int(rng.integers(100, 600))
        for _ in range(per):
            dx = int(rng.integers(0, 4 if dense else 61))
            dy = dx + int(round(rng.normal(0, 2 if dense else 8)))
            x += dx
            y += max(dy, 0)
            xs.append(x)
            qs.append(y)
    span = n * (3 if dense else 30) + 1000
    while len(xs) < n:
        xs.append(int(rng.integers(1000, 1000 + span)))
        qs.append(int(rng.integers(100, 100 + span)))
    xs = np.array(xs[:n], dtype=np.uint64)
    qs = np.array(qs[:n], dtype=np.uint64)
    if dup_x:                                   # equal reference positions -> dr == 0 paths
        idx = rng.integers(0, n, n // 10)
        xs[idx] = xs[(idx + 1) % n]
    qspan = rng.integers(10, 20, n).astype(np.uint64)
    seg = rng.integers(0, n_segs, n).astype(np.uint64)
    strand = (rng.random(n) < 0.1).astype(np.uint64) if dup_x else np.zeros(n, dtype=np.uint64)
    ax = (strand << np.uint64(63)) | xs
    ay = (seg << np.uint64(48)) | (qspan << np.uint64(32)) | (qs & np.uint64(0x7fffffff))
    order = np.lexsort((ay, ax))
    hdr = (np.float32(avg_qspan), max_dist, max_dist if n_segs == 1 else 800, bw, n_segs)
    return ax[order], ay[order], hdr


def chain_pack(calls):
    from genomicsbench_amd._native import CHAIN_CALL_DTYPE
    off = np.zeros(len(calls) + 1, dtype=np.int64)
    np.cumsum([len(c[0]) for c in calls], out=off[1:])
    ax = np.concatenate([c[0] for c in calls]) if calls else np.zeros(0, np.uint64)
    ay = np.concatenate([c[1] for c in calls]) if calls else np.zeros(0, np.uint64)
    hdr = np.array([c[2] for c in calls], dtype=CHAIN_CALL_DTYPE)
    return off, np.ascontiguousarray(ax), np.ascontiguousarray(ay), hdr


def chain_cases(seed=99):
    rng = np.random.default_rng(seed)
    cases = {}
    cases["mixed"] = chain_pack([_chain_call(rng, int(n)) for n in (1, 2, 3, 50, 64, 65, 300, 1000, 2500)] +
                                [_chain_call(rng, 700, avg_qspan=q) for q in (14.2857, 12.5, 20.0, 33.3333, 0.5)])
    # anti-diagonal: every predecessor is skipped (dq <= 0), so the look-back runs into max_iter = 5000
    k = np.arange(7000, dtype=np.uint64)
    anti = (np.uint64(1000) + k, (np.uint64(15) << np.uint64(32)) | (np.uint64(100000) - k),
            (np.float32(15.0), 100000, 100000, 500, 1))
    cases["dense_maxiter"] = chain_pack([_chain_call(rng, 7000, dense=True, max_dist=100000, bw=100000),
                                         _chain_call(rng, 5200, dense=True), anti])
    cases["multiseg"] = chain_pack([_chain_call(rng, 800, n_segs=2, dup_x=True),
                                    _chain_call(rng, 1500, n_segs=3, dup_x=True, avg_qspan=17.5),
                                    _chain_call(rng, 400, n_segs=1, dup_x=True)])
    cases.update(chain_cut_cases(seed + 1000))
    return cases


def _chain_clusters(rng, sizes, gap, keys=None, n_segs=1, max_dist=5000, bw=500, avg_qspan=15.0):
    """One call made of clusters of the given sizes: colinear runs `gap` reference bases apart (> max_dist: the chain
    kernel cuts the call there), optionally under different upper x words (strand / reference id) and segment ids."""
    xs, ys = [], []
    x = 1000
    for c, n in enumerate(sizes):
        key = int(keys[c % len(keys)]) if keys is not None else 0
        y = int(rng.integers(100, 4000))
        for _ in range(n):
            dx = int(rng.integers(0, 61))
            x += dx
            y += max(dx + int(round(rng.normal(0, 8))), 0)
            seg = int(rng.integers(0, n_segs))
            xs.append((key << 32) | x)
            ys.append((seg << 48) | (int(rng.integers(10, 20)) << 32) | (y & 0x7fffffff))
        x += gap + int(rng.integers(0, 50))
    ax, ay = np.array(xs, dtype=np.uint64), np.array(ys, dtype=np.uint64)
    order = np.lexsort((ay, ax))
    return ax[order], ay[order], (np.float32(avg_qspan), max_dist, max_dist if n_segs == 1 else 800, bw, n_segs)


def chain_cut_cases(seed=1099):
    """Calls that fall apart into independent pieces (an anchor further than max_dist_x from its predecessor looks back
    at nobody, host_kernel.cpp:56, and nothing after it looks across it): pieces of 1 anchor, pieces ending and starting
    on the 64-anchor block boundaries the kernel records its cuts by, several cuts inside one block, pieces under
    different strands / reference ids and with several segment ids, a gap of exactly max_dist_x (no cut) and one more
    (a cut), pieces long enough to leave the LDS ring, and the realistic generator of the bench."""
    from genomicsbench_amd.datagen import gen_chain
    rng = np.random.default_rng(seed)
    calls = [
        _chain_clusters(rng, [1, 1, 1, 5, 56, 64, 64, 1, 63, 65, 1, 200, 3, 700], 6000),
        _chain_clusters(rng, [30, 1, 1, 1, 1, 30, 2, 2, 2, 700, 1], 5001),
        _chain_clusters(rng, [64] * 6 + [128, 1, 127, 1], 9000),
        _chain_clusters(rng, [40, 300, 17, 1, 90, 1200, 2, 64], 7000, keys=[0, 0, 1, 1, 5, (1 << 31) | 2, (1 << 31) | 2, (1 << 31) | 4]),
        _chain_clusters(rng, [100, 50, 1, 400, 64, 20], 7000, keys=[3, 3, 4], n_segs=2),
        _chain_clusters(rng, [1500, 900, 1, 2000], 5200),
    ]
    # a gap of exactly max_dist_x is still looked across; one base more is not
    x = np.uint64(1000) + np.arange(300, dtype=np.uint64) * np.uint64(10)
    x[100:] += np.uint64(5000 - 10)
    x[200:] += np.uint64(5001 - 10)
    y = (np.uint64(15) << np.uint64(32)) | (np.uint64(100) + np.arange(300, dtype=np.uint64) * np.uint64(10))
    calls.append((x, y, (np.float32(15.0), 5000, 5000, 500, 1)))
    off, ax, ay, hdr = gen_chain(12, 7001, realistic=True,
                                 n_override=[50, 64, 65, 129, 400, 1000, 2000, 3000, 3000, 5000, 7000, 9000])
    real = [(ax[off[c]:off[c + 1]], ay[off[c]:off[c + 1]], tuple(hdr[c])) for c in range(12)]
    return {"cuts": chain_pack(calls), "realistic": chain_pack(real)}
