"""Drop-in boundary, end to end: the reference's own, UNMODIFIED drivers (bsw main_banded.cpp, chain main.cpp,
phmm PairHMMUnitTest.cpp, poa msa_spoa_omp.cpp), compiled from /root/reference by oracle/build_ref.sh against our
libraries instead of the reference kernels (csrc/shims/*.cpp and the product's include/spoa/*.hpp are the
bindings), run on the GPU and must reproduce the oracle.  The binaries live in oracle/_ref (built where the
reference exists, shipped to the GPU box)."""
import os
import subprocess

import numpy as np
import pytest

from genomicsbench_amd import io as gio
from genomicsbench_amd.bsw import make_params
from genomicsbench_amd.datagen import gen_bsw, gen_chain, gen_phmm, gen_poa
from oracle import oracle_py as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref")


def driver(name):
    path = os.path.join(REF, name)
    if not os.path.exists(path):
        pytest.skip("%s not built (no /root/reference at build time)" % name)
    return path


def run(args, env=None):
    e = dict(os.environ)
    e.update(env or {})
    return subprocess.run(args, capture_output=True, text=True, timeout=900, env=e)


def test_reference_bsw_driver_on_gbx(tmp_path):
    b = gen_bsw(3000, 21)
    pairs, dump = str(tmp_path / "pairs.txt"), str(tmp_path / "dump.txt")
    gio.write_bsw_pairs(pairs, b)
    # one thread: the batches then reach getScores16 in file order (the ids in SeqPair are batch-local)
    r = run([driver("bsw_refdriver_gbx"), "-pairs", pairs, "-t", "1", "-b", "512"], {"GBX_SHIM_DUMP": dump})
    assert "Number of input pairs: 3000" in r.stdout, r.stdout + r.stderr      # the driver exits 1 by design (:352)
    assert os.path.exists(dump), "getScores16 shim wrote nothing:\n" + r.stdout[-600:] + r.stderr[-600:]
    got = np.loadtxt(dump, dtype=np.int64)
    want = O.bsw_oracle(make_params(), b, 2)
    assert got.shape[0] == b.n and np.array_equal(got[:, 1:], want)


def test_reference_chain_driver_on_gbx(tmp_path):
    case = gen_chain(12, 22)
    inp, out = str(tmp_path / "chain.in"), str(tmp_path / "chain.out")
    gio.write_chain_calls(inp, *case)
    r = run([driver("chain_refdriver_gbx"), "-i", inp, "-o", out, "-t", "1"])
    assert r.returncode == 0, r.stderr
    score, parent = O.chain_oracle(*case)[:2]
    got_s, got_p = [], []
    for line in open(out):
        f = line.split()
        if len(f) == 2:
            got_s.append(int(f[0])); got_p.append(int(f[1]))
    assert np.array_equal(np.array(got_s), score) and np.array_equal(np.array(got_p), parent)


def test_reference_phmm_driver_on_our_gkl_library(tmp_path):
    bs = gen_phmm(6, 23)
    inp = str(tmp_path / "phmm.in")
    gio.write_phmm_batches(inp, bs)
    r = run([driver("phmm_refdriver_gbx"), "-f", inp, "-t", "2"])
    assert r.returncode == 0 and "PairHMM completed" in r.stdout, r.stdout + r.stderr
    got = []
    for line in r.stdout.splitlines():
        try:
            got.append(float(line))
        except ValueError:
            pass
    got = np.array(got[-bs.n_pairs:])
    want = O.phmm_oracle(bs, 4)
    assert len(got) == bs.n_pairs
    assert np.all(np.abs(got - want) <= 1e-5 * np.maximum(1.0, np.abs(want)) + 1e-6)      # printed with 6 decimals


@pytest.mark.parametrize("threads", [1, 4])
def test_reference_poa_driver_on_the_spoa_facade(tmp_path, threads):
    """msa_spoa_omp.cpp (-DPRINT_OUTPUT) over include/spoa/spoa.hpp: createAlignmentEngine / createGraph / align /
    add_alignment / generate_consensus (:189-190,237-252) bind to gbx_poa_consensus_host, one window per call."""
    from genomicsbench_amd.poa import make_params
    ws = gen_poa(10, 24)
    inp = str(tmp_path / "poa.fa")
    gio.write_poa_windows(inp, ws)
    r = run([driver("poa_refdriver_gbx"), "-s", inp, "-t", str(threads)])
    assert r.returncode == 0, r.stdout[-600:] + r.stderr[-600:]
    lines = r.stdout.splitlines()
    got = [lines[k + 1] for k in range(len(lines) - 1) if lines[k] == ">Consensus_sequence"]
    assert got == O.poa_oracle(make_params(), ws, 4)


def test_reference_poa_driver_rejects_positive_gap_penalties(tmp_path):
    """createAlignmentEngine throws std::invalid_argument, the driver prints it and returns 1 (:191-194)."""
    ws = gen_poa(1, 24)
    inp = str(tmp_path / "poa.fa")
    gio.write_poa_windows(inp, ws)
    r = run([driver("poa_refdriver_gbx"), "-s", inp, "-o", "-9,24"])         # -o negates: o1 = +9 -> g = o1+e1 > 0
    assert r.returncode == 1 and "non-positive" in r.stderr


def test_every_public_member_of_the_bsw_class_on_gbx(tmp_path):
    """bandedSWA.h:116-315: scalarBandedSWA (one pair per call, here from eight threads at once: the calls share device
    calls), scalarBandedSWAWrapper, getScores8 / getScores16 and their batch wrappers, all on the shim, all six output
    fields against the oracle - and against the reference's own scalar members run from the same harness on the CPU."""
    b = gen_bsw(300, 31)
    pairs = str(tmp_path / "pairs.txt")
    gio.write_bsw_pairs(pairs, b)
    want = O.bsw_oracle(make_params(), b, 2)
    r = run([driver("bsw_members_gbx"), pairs, "sw8bB6", "8"])
    assert r.returncode == 0, r.stdout[-400:] + r.stderr[-400:]
    got = {}
    for line in r.stdout.splitlines():
        f = line.split()
        got.setdefault(f[0], []).append([int(x) for x in f[1:]])
    members = ["scalarBandedSWA", "scalarBandedSWAWrapper", "getScores8", "smithWatermanBatchWrapper8", "smithWatermanBatchWrapper16", "getScores16"]
    assert sorted(got) == sorted(members)
    for m in members:
        a = np.array(got[m], dtype=np.int64)
        a = a[np.argsort(a[:, 0], kind="stable")]
        assert a.shape[0] == b.n and np.array_equal(a[:, 0], np.arange(b.n)), m
        assert np.array_equal(a[:, 1:], want), m
    ref = os.path.join(REF, "bsw_members_ref")
    if os.path.exists(ref):
        rr = run([ref, pairs, "sw"])
        assert rr.returncode == 0, rr.stderr[-400:]
        for line in rr.stdout.splitlines():
            f = line.split()
            assert [int(x) for x in f[2:]] == list(want[int(f[1])]), line
