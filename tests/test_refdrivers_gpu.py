"""Drop-in boundary, end to end: the reference's own, UNMODIFIED drivers (bsw main_banded.cpp, chain main.cpp,
phmm PairHMMUnitTest.cpp), compiled from /root/reference by oracle/build_ref.sh against our libraries instead
of the reference kernels (csrc/shims/*.cpp are the bindings), run on the GPU and must reproduce the oracle.
The binaries live in oracle/_ref (built where the reference exists, shipped to the GPU box); the poa driver
cannot be built (it includes spoa's headers, an empty submodule)."""
import os
import subprocess

import numpy as np
import pytest

from genomicsbench_amd import io as gio
from genomicsbench_amd.bsw import make_params
from genomicsbench_amd.datagen import gen_bsw, gen_chain, gen_phmm
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
