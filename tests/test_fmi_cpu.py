"""CPU-side checks for fmi.  bwa-mem2 (tools/bwa-mem2) is an empty submodule: nothing pins the restatement against a
compiled reference.  What is checked here instead, independently of bwa-mem2: the index tables against a plain loop and
a brute-force suffix array, every reported (k, l, s) against brute-force suffix-array ranges, the round-1 SMEMs against
a brute-force enumeration of the super-maximal exact matches, the re-seeding filter (that one is in the tree,
fmi.cpp:230-240) and the sort order (rid, m ascending, n descending)."""
import numpy as np
import pytest

from genomicsbench_amd.datagen import gen_fmi_genome, gen_fmi_reads
from genomicsbench_amd.fmi import FmiReadSet, build_index, default_params, suffix_array
from oracle import oracle_py as O


def brute_sa(text):
    """suffix array incl. the sentinel suffix (smallest) by plain sorting of the suffixes"""
    n = len(text)
    t = bytes(int(c) + 1 for c in text) + b"\x00"
    return np.array(sorted(range(n + 1), key=lambda i: t[i:]), dtype=np.int64)


def text_of(ref):
    return np.concatenate([ref, 3 - ref[::-1]]).astype(np.uint8)


def small_genome(n, seed, repeats=True):
    rng = np.random.default_rng(seed)
    g = rng.integers(0, 4, n).astype(np.uint8)
    if repeats:
        for _ in range(max(1, n // 400)):
            L = int(rng.integers(20, 120))
            a, b = (int(v) for v in rng.integers(0, n - L, 2))
            seg = g[a:a + L].copy()
            if rng.random() < 0.5:
                seg = 3 - seg[::-1]
            hit = rng.random(L) < 0.03
            seg[hit] = (seg[hit] + 1) % 4
            g[b:b + L] = seg
    return g


@pytest.mark.parametrize("n,seed", [(1, 1), (2, 2), (63, 3), (64, 4), (700, 5), (5000, 6)])
def test_suffix_array_and_index_tables(n, seed):
    g = small_genome(n, seed, repeats=n > 100)
    text = text_of(g)
    sa = suffix_array(text).numpy()
    assert np.array_equal(sa, brute_sa(text))
    idx = build_index(g)
    plain = O.fmi_build_index_plain(text, sa)
    assert idx.ref_seq_len == plain.ref_seq_len == 2 * n + 1
    assert idx.count == plain.count and idx.sentinel_index == plain.sentinel_index
    assert np.array_equal(idx.cp_occ.view(np.uint8), plain.cp_occ.view(np.uint8))
    assert idx.count[0] == 1 and idx.count[4] == 2 * n + 1


_SUFFIXES = {}


def sa_range(text, sa, pat):
    """rows of the suffix array whose suffix starts with pat (brute force: bisection over the sorted list of all suffixes
    as byte strings, symbols + 1, the sentinel = 0) -> (first row, count)"""
    import bisect
    key = (len(text), text[:64].tobytes(), text[-64:].tobytes())
    if key not in _SUFFIXES:
        t = bytes(int(c) + 1 for c in text) + b"\x00"
        _SUFFIXES.clear()
        _SUFFIXES[key] = [t[i:] for i in sa]
    suf = _SUFFIXES[key]
    p = bytes(int(c) + 1 for c in pat)
    lo, hi = bisect.bisect_left(suf, p), bisect.bisect_left(suf, p + b"\xff")
    return (lo, hi - lo) if hi > lo else (None, 0)


def test_every_interval_is_the_suffix_array_range_of_its_match():
    g = small_genome(1500, 11)
    text, idx = text_of(g), build_index(g)
    sa = brute_sa(text)
    rng = np.random.default_rng(12)
    reads = []
    for _ in range(60):
        p = int(rng.integers(0, len(g) - 80))
        r = g[p:p + 80].copy()
        if rng.random() < 0.5:
            r = 3 - r[::-1]
        hit = rng.random(80) < 0.04
        r[hit] = (r[hit] + 1) % 4
        if rng.random() < 0.2:
            r[int(rng.integers(0, 80))] = 4
        reads.append(r)
    rs = FmiReadSet.fixed(np.array(reads))
    P = default_params(12)
    out, off = O.fmi_oracle(idx, rs, P)
    assert len(out) > 60
    for s in out:
        q = reads[int(s["rid"])][int(s["m"]):int(s["n"]) + 1]
        assert (q < 4).all()
        k, cnt = sa_range(text, sa, q)
        assert cnt == s["s"] and k == s["k"]
        l, cnt2 = sa_range(text, sa, (3 - q[::-1]).astype(np.uint8))
        assert cnt2 == s["s"] and l == s["l"]


def occurs(text_set_cache, text, pat):
    if "text" not in text_set_cache:
        text_set_cache["text"] = text.tobytes()
    return pat.tobytes() in text_set_cache["text"]


def brute_smems(text, q, min_len):
    """super-maximal exact matches of q against text: intervals [m, n] of q that occur in text, cannot be extended to
    either side, and are not contained in another such interval; bounded by ambiguous bases."""
    L, cache, mems = len(q), {}, []
    for m in range(L):
        if q[m] > 3:
            continue
        n = m
        while n + 1 < L and q[n + 1] < 4 and occurs(cache, text, q[m:n + 2]):
            n += 1
        if not occurs(cache, text, q[m:n + 1]):
            continue
        left_ext = m > 0 and q[m - 1] < 4 and occurs(cache, text, q[m - 1:n + 1])
        if not left_ext:
            mems.append((m, n))
    keep = [iv for iv in mems if not any(o != iv and o[0] <= iv[0] and iv[1] <= o[1] for o in mems)]
    return sorted(iv for iv in keep if iv[1] - iv[0] + 1 >= min_len)


def test_round_one_reports_the_super_maximal_exact_matches():
    """getSMEMsAllPosOneThread with min_intv 1: exactly the SMEMs of at least minSeedLen bases (round 1 alone: split
    and LAST rounds switched off through the parameters)."""
    g = small_genome(600, 21)
    text, idx = text_of(g), build_index(g)
    rng = np.random.default_rng(22)
    P = default_params(8)
    P.split_len, P.max_mem_intv = 1 << 20, 0           # no re-seeding, and "s < 0" never holds in the LAST round
    reads = []
    for _ in range(25):
        p = int(rng.integers(0, len(g) - 50))
        r = g[p:p + 50].copy()
        hit = rng.random(50) < 0.06
        r[hit] = (r[hit] + 1) % 4
        if rng.random() < 0.3:
            r[int(rng.integers(0, 50))] = 4
        reads.append(r)
    rs = FmiReadSet.fixed(np.array(reads))
    out, off = O.fmi_oracle(idx, rs, P)
    for r, q in enumerate(reads):
        got = sorted(set((int(s["m"]), int(s["n"])) for s in out[off[r]:off[r + 1]]))
        assert got == brute_smems(text, q, 8), "read %d" % r


def test_sort_order_reseeding_filter_and_last_round():
    g = gen_fmi_genome(60000, 7)
    idx = build_index(g)
    rs = gen_fmi_reads(g, 400, 8)
    P = default_params(19)
    out, off, ext, rounds = O.fmi_oracle(idx, rs, P, nthreads=4, return_stats=True)
    assert off[-1] == len(out) and rounds[0] > 400 and rounds[2] > 0 and ext > 400 * 151
    key = np.stack([out["rid"].astype(np.int64), out["m"].astype(np.int64), -out["n"].astype(np.int64)], 1)
    assert all(tuple(key[i]) <= tuple(key[i + 1]) for i in range(len(key) - 1))
    assert np.all(out["rid"] == np.repeat(np.arange(400), np.diff(off)))
    assert np.all(out["s"] >= 1) and np.all(out["n"] - out["m"] + 1 >= 19)
    # without the two extra rounds a subset comes out, and re-seeding only ever adds SMEMs with more hits than the
    # SMEM they were seeded from
    P1 = default_params(19)
    P1.split_len, P1.max_mem_intv = 1 << 20, 0
    base, boff = O.fmi_oracle(idx, rs, P1)
    assert len(base) == rounds[0]
    assert set(map(tuple, base.tolist())) <= set(map(tuple, out.tolist()))


def test_reads_without_any_valid_base_and_short_reads():
    g = small_genome(400, 31)
    idx = build_index(g)
    enc = np.concatenate([np.full(30, 4), g[10:12], g[50:90], np.array([2])]).astype(np.uint8)
    rs = FmiReadSet(enc, np.array([0, 30, 32, 72]), np.array([30, 2, 40, 1]))
    out, off = O.fmi_oracle(idx, rs, default_params(19))
    assert list(np.diff(off)[:2]) == [0, 0] and off[3] - off[2] >= 1 and off[4] == off[3]


def test_frozen_fixture_equals_the_oracle_and_brute_force():
    """tests/golden/fmi_small.* (written by the oracle, frozen): the oracle still produces it, and every record in it is
    right by brute force - (k, s) and (l, s) are the suffix-array ranges of the match and of its reverse complement, no
    record can be extended to either side, and the round-1 SMEMs of every read are among them."""
    from util import load_fmi_golden
    g, rs, want = load_fmi_golden()
    idx = build_index(g)
    out, off = O.fmi_oracle(idx, rs, default_params(19))
    got = np.stack([out[f].astype(np.int64) for f in ("rid", "m", "n", "k", "l", "s")], 1)
    assert np.array_equal(got, want)
    text = text_of(g)
    sa = brute_sa(text)
    cache = {}
    for rid, m, n, k, l, s in want[::3]:
        a = int(rs.read_off[rid])
        q = rs.enc[a:a + int(rs.read_len[rid])]
        pat = q[m:n + 1]
        assert sa_range(text, sa, pat) == (k, s) and sa_range(text, sa, (3 - pat[::-1]).astype(np.uint8)) == (l, s)
    for r in range(0, rs.n_reads, 5):
        a = int(rs.read_off[r])
        q = rs.enc[a:a + int(rs.read_len[r])]
        mine = set((int(x[1]), int(x[2])) for x in want[want[:, 0] == r])
        assert set(brute_smems(text, q, 19)) <= mine, "read %d" % r
