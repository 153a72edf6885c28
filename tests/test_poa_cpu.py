"""CPU-side checks for poa: the oracle against known answers, and the product's serial graph code
(host build of genomicsbench_amd/csrc/poa_graph.h, tests/hostcheck) against the oracle.

The reference arithmetic (spoa) is un-vendored: parity with spoa itself is UNPINNED (see oracle/poa_oracle.c).
"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from genomicsbench_amd.datagen import gen_poa
from genomicsbench_amd.poa import PoaWindowSet, make_params
from oracle import oracle_py as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_known_answers():
    p = make_params()
    ws = PoaWindowSet.from_lists([
        ["ACGTACGTAC"] * 3,                                   # identical reads -> the read
        ["ACGTACGTAC", "ACGTACGTAC", "ACGAACGTAC"],           # one substitution outvoted
        ["ACGTACGTAC", "ACGTCGTAC", "ACGTACGTAC"],            # one deletion outvoted
        ["AAAA", "AATAA", "AATAA"],                           # insertion wins 2:1
        ["ACGT"],                                             # single read
        ["GATTACA", "GATTACA", "GATTTACA", "GATTACA"],
    ])
    assert O.poa_oracle(p, ws) == ["ACGTACGTAC", "ACGTACGTAC", "ACGTACGTAC", "AATAA", "ACGT", "GATTACA"]


def test_params_follow_driver_cli():
    p = make_params()
    assert (p.m, p.n, p.g, p.e, p.q, p.c) == (2, -4, -6, -2, -25, -1)       # msa_spoa_omp.cpp:156-162,184
    p = make_params(m=3, x=5, o1=5, e1=3, o2=30, e2=2)
    assert (p.m, p.n, p.g, p.e, p.q, p.c) == (3, -5, -8, -3, -32, -2)


def test_consensus_recovers_backbone_and_is_order_stable():
    ws = gen_poa(6, 4001)
    p = make_params()
    cons = O.poa_oracle(p, ws, 4)
    for w, c in enumerate(cons):
        assert 440 <= len(c) <= 540 and set(c) <= set("ACGT")
    # same windows again give the same answer (no hidden state between windows)
    assert O.poa_oracle(p, ws, 1) == cons
    # a window of one read returns the read
    one = PoaWindowSet.from_lists([[ws.window(0)[0]]])
    assert O.poa_oracle(p, one) == [ws.window(0)[0]]


@pytest.fixture(scope="module")
def hostcheck():
    src = os.path.join(ROOT, "tests", "hostcheck", "poa_hostcheck.cpp")
    out = os.path.join(ROOT, "tests", "hostcheck", "libpoa_hostcheck.so")
    hdr = os.path.join(ROOT, "genomicsbench_amd", "csrc", "poa_graph.h")
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", src, "-o", out], check=True)
    return C.CDLL(out)


def _host_window(L, p, seqs, ncap=4096, deg=64, cap=4096):
    arr = (C.c_char_p * len(seqs))(*[s.encode() for s in seqs])
    lens = (C.c_int32 * len(seqs))(*[len(s) for s in seqs])
    buf = C.create_string_buffer(cap)
    st = (C.c_int64 * 2)()
    n = L.hostcheck_poa_window(C.byref(p), len(seqs), arr, lens, buf, cap, ncap, deg, st)
    return buf.raw[:min(n, cap)].decode(), st[1]


def test_product_graph_code_matches_oracle(hostcheck):
    """add_alignment / topological sort / traceback / consensus of the product (host build) == oracle."""
    p = make_params()
    ws = gen_poa(8, 77)
    want = O.poa_oracle(p, ws, 4)
    for w in range(ws.n_windows):
        got, err = _host_window(hostcheck, p, ws.window(w))
        assert err == 0 and got == want[w]
    rng = np.random.default_rng(1)
    wins = []
    for _ in range(30):
        base = "".join(rng.choice(list("ACGTN"), int(rng.integers(5, 60)), p=[.24, .24, .24, .24, .04]))
        reads = []
        for _ in range(int(rng.integers(1, 9))):
            r = [c for c in base if rng.random() > 0.1]
            r = [c if rng.random() > 0.1 else "ACGT"[int(rng.integers(4))] for c in r]
            for _ in range(int(rng.integers(0, 3))):
                r.insert(int(rng.integers(0, len(r) + 1)), "ACGT"[int(rng.integers(4))])
            reads.append("".join(r) or "A")
        wins.append(reads)
    ws = PoaWindowSet.from_lists(wins)
    want = O.poa_oracle(p, ws)
    for w in range(ws.n_windows):
        got, err = _host_window(hostcheck, p, ws.window(w))
        assert err == 0 and got == want[w], (w, wins[w])
    # affine scoring (second piece disabled) goes through the same machinery
    pa = make_params(o2=4, e2=2)
    want = O.poa_oracle(pa, ws)
    for w in range(ws.n_windows):
        assert _host_window(hostcheck, pa, ws.window(w))[0] == want[w]


def _noisy_windows(seed, n=30):
    rng = np.random.default_rng(seed)
    wins = []
    for _ in range(n):
        base = "".join(rng.choice(list("ACGTN"), int(rng.integers(5, 60)), p=[.24, .24, .24, .24, .04]))
        reads = []
        for _ in range(int(rng.integers(1, 9))):
            r = [c for c in base if rng.random() > 0.1]
            r = [c if rng.random() > 0.1 else "ACGT"[int(rng.integers(4))] for c in r]
            for _ in range(int(rng.integers(0, 3))):
                r.insert(int(rng.integers(0, len(r) + 1)), "ACGT"[int(rng.integers(4))])
            reads.append("".join(r) or "A")
        wins.append(reads)
    return wins


def test_linear_gap_subtype_known_answers():
    """spoa's linear subtype (g >= e; the driver's -o 0,... reaches it: msa_spoa_omp.cpp:170-196): one gap cost.  Identical
    reads give the read; a majority wins; and where linear and affine costs disagree about an alignment the consensus shows
    which one ran: with g = -2 per gap base two separate one-base gaps cost what one two-base gap costs."""
    pl = make_params(o1=0, e1=2)                      # g = e = -2
    assert pl.g >= pl.e
    for seqs in (["ACGTACGTAC"] * 4, ["ACGTTGCA", "ACGTTGCA", "ACGATGCA"], ["AAAACCCCGGGG", "AAAACCCCGGGG", "AAAAGGGG"]):
        want = seqs[0]
        assert O.poa_oracle(pl, PoaWindowSet.from_lists([seqs]))[0] == want


def test_linear_gap_subtype_product_graph_code_matches_oracle(hostcheck):
    """The device path runs the linear subtype as the affine DP with e = q = c = g (the same score matrix) and a backtrack that
    takes single cells only (poa_graph.h: PoaScore::linear): the product's serial code (host build) against the oracle's
    one-matrix restatement, on noisy windows where ties between diagonal, vertical and horizontal moves are common."""
    for pl in (make_params(o1=0, e1=2), make_params(o1=0, e1=4, o2=2, e2=1), make_params(m=1, x=1, o1=0, e1=1)):
        assert pl.g >= pl.e
        for seed in (3, 4):
            wins = _noisy_windows(seed)
            ws = PoaWindowSet.from_lists(wins)
            want = O.poa_oracle(pl, ws)
            for w in range(ws.n_windows):
                got, err = _host_window(hostcheck, pl, ws.window(w))
                assert err == 0 and got == want[w], (w, wins[w])
        ws = gen_poa(4, 78)
        want = O.poa_oracle(pl, ws, 4)
        for w in range(ws.n_windows):
            assert _host_window(hostcheck, pl, ws.window(w)) == (want[w], 0)
    # and the two subtypes are different algorithms: on the same noisy windows some consensus differs
    pa, pl = make_params(), make_params(o1=0, e1=2)
    ws = PoaWindowSet.from_lists(_noisy_windows(5, 60))
    assert O.poa_oracle(pa, ws) != O.poa_oracle(pl, ws)


def test_capacity_overflow_is_reported(hostcheck):
    p = make_params()
    ws = gen_poa(1, 5)
    got, err = _host_window(hostcheck, p, ws.window(0), ncap=600)
    assert err & 1                                                      # POA_ERR_NODES


def test_incremental_topological_sort_model_equals_the_full_sort():
    """The device's topological sort (csrc/poa_kernels.hip: poa_topo_sort_lds) no longer walks the whole graph after every
    sequence: the previous order is cut into the blocks the walks from the roots emitted, untouched blocks are copied and
    the walk is repeated only from the roots of blocks that hold a node with a new in-edge or aligned node.  Here a host
    model of exactly that scheme (tests/hostcheck/poa_topo_inc_check.cpp) runs beside the full sort (poa_graph.h:
    poa_topo_sort, spoa's order) after every add_alignment of real windows: the two orders must be the same, sort after
    sort.  (On the device the same comparison is a -DGBX_POA_TOPO_CHECK build: scripts/dbg_poa_phases.py.)"""
    import subprocess
    src = os.path.join(ROOT, "tests", "hostcheck", "poa_topo_inc_check.cpp")
    out = os.path.join(ROOT, "tests", "hostcheck", "libpoa_topo_inc_check.so")
    subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-I", os.path.join(ROOT, "include"), src, "-o", out], check=True)
    L = C.CDLL(out)
    p = make_params()
    sorts = walked = blocks = 0
    for seed, nw in ((4001, 40), (99, 20), (123456, 20)):
        ws = gen_poa(nw, seed)
        for w in range(ws.n_windows):
            seqs = ws.window(w)
            arr = (C.c_char_p * len(seqs))(*[s.encode() if isinstance(s, str) else bytes(s) for s in seqs])
            lens = (C.c_int32 * len(seqs))(*[len(s) for s in seqs])
            st = (C.c_int64 * 4)()
            ncap = min(sum(len(s) for s in seqs), 6 * max(len(s) for s in seqs) + 256)
            assert L.hostcheck_poa_topo_inc(C.byref(p), len(seqs), arr, lens, ncap, 40, st) == 0, (seed, w)
            sorts += st[0]; walked += st[1]; blocks += st[2]
    assert sorts > 2000 and walked < 0.25 * blocks           # and the point of it: a seventh of the blocks are walked
