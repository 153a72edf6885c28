"""CPU-side checks for chain: oracle vs the reference's goldens, file format round trip."""
import numpy as np
import pytest

from cases import chain_cases
from genomicsbench_amd import io as gio
from genomicsbench_amd.datagen import gen_chain
from oracle import oracle_py as O
from util import load_chain_golden


@pytest.mark.parametrize("name", ["mixed", "dense_maxiter", "multiseg", "cuts", "realistic"])
def test_oracle_matches_reference_goldens(name):
    case, g = load_chain_golden(name)
    s, p, t, k = O.chain_oracle(*case)
    assert np.array_equal(s, g[:, 0]) and np.array_equal(p, g[:, 1])
    assert np.array_equal(t, g[:, 2]) and np.array_equal(k, g[:, 3])


@pytest.mark.skipif(O.ref_lib("chain") is None, reason="compiled reference only exists in the build container")
def test_oracle_matches_live_reference():
    for case in list(chain_cases(seed=5).values()) + [gen_chain(40, 2001), gen_chain(40, 2001, realistic=True)]:
        for a, b in zip(O.chain_oracle(*case), O.chain_ref(*case)):
            assert np.array_equal(a, b)


def test_known_answers():
    """Two colinear anchors 10 apart with span 15: second chains onto the first with score 15+10."""
    from genomicsbench_amd._native import CHAIN_CALL_DTYPE
    off = np.array([0, 2], dtype=np.int64)
    ax = np.array([100, 110], dtype=np.uint64)
    ay = np.array([(15 << 32) | 50, (15 << 32) | 60], dtype=np.uint64)
    hdr = np.array([(15.0, 5000, 5000, 500, 1)], dtype=CHAIN_CALL_DTYPE)
    s, p, t, k = O.chain_oracle(off, ax, ay, hdr)
    assert list(s) == [15, 25] and list(p) == [-1, 0] and list(k) == [15, 25] and list(t) == [0, 0]
    empty = O.chain_oracle(np.array([0, 0], dtype=np.int64), ax[:0], ay[:0], hdr)
    assert all(len(a) == 0 for a in empty)


def test_chain_file_roundtrip(tmp_path):
    case = chain_cases()["multiseg"]
    path = str(tmp_path / "c.in")
    gio.write_chain_calls(path, *case)
    back = gio.read_chain_calls(path)
    assert np.array_equal(back[0], case[0]) and np.array_equal(back[1], case[1]) and np.array_equal(back[2], case[2])
    for a, b in zip(O.chain_oracle(*case), O.chain_oracle(*back)):
        assert np.array_equal(a, b)
    out = str(tmp_path / "c.out")
    s, p, _, _ = O.chain_oracle(*case)
    gio.write_chain_returns(out, case[0], s, p)
    lines = open(out).read().split("\n")
    assert lines[0] == str(case[0][1]) and lines[1] == "%d\t%d" % (s[0], p[0]) and "EOR" in lines
