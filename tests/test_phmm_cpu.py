"""CPU-side checks for phmm: the oracle against hand-computable answers and an independent evaluation.

The reference arithmetic (GKL) is un-vendored: parity with GKL itself is UNPINNED (see oracle/phmm_oracle.c);
these tests pin the oracle to the published recurrences instead.
"""
import math

import numpy as np
import pytest

from genomicsbench_amd import io as gio
from genomicsbench_amd.datagen import gen_phmm
from oracle import oracle_py as O


def ph(x):
    return 10.0 ** (-x / 10.0)


def slow_forward(rs, hap, q, qi, qd, qc):
    """Independent float64 evaluation with the closed-form match-to-match probability 1-(P_i+P_d)."""
    R, H = len(rs), len(hap)
    M = [[0.0] * (H + 1) for _ in range(R + 1)]
    X = [[0.0] * (H + 1) for _ in range(R + 1)]
    Y = [[0.0] * (H + 1) for _ in range(R + 1)]
    for c in range(H + 1):
        Y[0][c] = 1.0 / H
    for r in range(1, R + 1):
        i, d, cc, qq = qi[r - 1], qd[r - 1], qc[r - 1], q[r - 1]
        mm, gapm, mx, xx, my, yy = 1 - (ph(i) + ph(d)), 1 - ph(cc), ph(i), ph(cc), ph(d), ph(cc)
        for c in range(1, H + 1):
            a, b = rs[r - 1], hap[c - 1]
            prior = 1 - ph(qq) if (a == b or a == "N" or b == "N") else ph(qq) / 3
            M[r][c] = prior * (M[r - 1][c - 1] * mm + (X[r - 1][c - 1] + Y[r - 1][c - 1]) * gapm)
            X[r][c] = M[r - 1][c] * mx + X[r - 1][c] * xx
            Y[r][c] = M[r][c - 1] * my + Y[r][c - 1] * yy
    return math.log10(sum(M[R][c] + X[R][c] for c in range(H + 1)))


def test_known_answers_1x1_and_2x2():
    v, ud = O.phmm_pair("A", "A", [30], [45], [45], [10])
    assert ud == 0 and v == pytest.approx(math.log10((1 - 1e-3) * (1 - 0.1)), rel=1e-5)
    v, _ = O.phmm_pair("A", "C", [30], [45], [45], [10])
    assert v == pytest.approx(math.log10(1e-3 / 3 * 0.9), rel=1e-5)
    v, _ = O.phmm_pair("N", "C", [30], [45], [45], [10])           # N matches anything
    assert v == pytest.approx(math.log10((1 - 1e-3) * 0.9), rel=1e-5)
    v, _ = O.phmm_pair("AC", "AC", [30, 30], [45, 45], [45, 45], [10, 10])
    assert v == pytest.approx(slow_forward("AC", "AC", [30, 30], [45, 45], [45, 45], [10, 10]), rel=1e-5)


def test_oracle_vs_independent_float64():
    rng = np.random.default_rng(5)
    for _ in range(40):
        R, H = int(rng.integers(1, 40)), int(rng.integers(1, 60))
        hap = "".join(rng.choice(list("ACGTN"), H, p=[.24, .24, .24, .24, .04]))
        st = int(rng.integers(0, max(1, H - R + 1)))
        rs = "".join(c if rng.random() > 0.05 else "ACGT"[int(rng.integers(4))] for c in (hap * 3)[st:st + R])
        q = rng.integers(6, 42, R).tolist()
        qi = rng.integers(30, 50, R).tolist()
        qd = rng.integers(30, 50, R).tolist()
        qc = [10] * R
        want = slow_forward(rs, hap, q, qi, qd, qc)
        got, _ = O.phmm_pair(rs, hap, q, qi, qd, qc)
        assert got == pytest.approx(want, rel=2e-5), (rs, hap)
        assert O.phmm_pair(rs, hap, q, qi, qd, qc, f64_only=True) == pytest.approx(want, rel=1e-6)


def test_fp64_fallback_triggers_and_agrees():
    """A read that mismatches everywhere drives the fp32 result below 1e-28 and takes the fp64 path."""
    R = 60
    rs, hap = "A" * R, "C" * 80
    q, qi, qd, qc = [40] * R, [45] * R, [45] * R, [10] * R
    v, ud = O.phmm_pair(rs, hap, q, qi, qd, qc)
    assert ud == 1
    assert v == pytest.approx(O.phmm_pair(rs, hap, q, qi, qd, qc, f64_only=True), rel=1e-12)
    assert v == pytest.approx(slow_forward(rs, hap, q, qi, qd, qc), rel=1e-5)


def test_generated_batches_f32_close_to_f64_and_file_roundtrip(tmp_path):
    bs = gen_phmm(6, 3001)
    out = O.phmm_oracle(bs, 4)
    assert np.isfinite(out).all() and (out < 0).all()
    path = str(tmp_path / "p.in")
    gio.write_phmm_batches(path, bs)
    back = gio.read_phmm_batches(path)
    assert back.n_pairs == bs.n_pairs and np.array_equal(back.read_len, bs.read_len)
    assert np.array_equal(O.phmm_oracle(back, 4), out)
    # pair order is read-major / hap-minor inside a batch (PairHMMUnitTest.cpp:232-244)
    assert bs.pair_read[0] == 0 and bs.pair_hap[1] == 1 and bs.pair_read[bs.n_haps[0]] == 1
