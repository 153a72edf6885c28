"""GPU parity tests for chain: HIP kernel (through the C-ABI) vs the oracle, bit-exact on all four outputs."""
import numpy as np
import pytest

from cases import chain_cases, chain_pack, _chain_call
from genomicsbench_amd.chain import DeviceChainBatch, chain_host
from genomicsbench_amd.datagen import gen_chain
from oracle import oracle_py as O
from util import load_chain_golden

pytestmark = pytest.mark.gpu
NAMES = ("score", "parent", "target", "peak")


def assert_same(got, want):
    for name, g, w in zip(NAMES, got, want):
        if not np.array_equal(g, w):
            k = int(np.nonzero(g != w)[0][0])
            raise AssertionError("%s differs at %d anchors; first at %d: got %d want %d" %
                                 (name, int((g != w).sum()), k, g[k], w[k]))


@pytest.mark.parametrize("ring", ["auto", "short", "long"])
@pytest.mark.parametrize("name", ["mixed", "dense_maxiter", "multiseg", "cuts", "realistic"])
def test_goldens(name, ring, monkeypatch):
    """Both instances of the DP kernel (512- and 768-anchor ring; a job picks one on the device) against the goldens."""
    if ring != "auto":
        monkeypatch.setenv("GBX_CHAIN_RING", ring)
    case, g = load_chain_golden(name)
    assert_same(chain_host(*case), [g[:, 0], g[:, 1], g[:, 2], g[:, 3]])


def test_many_short_calls_take_the_short_ring_and_long_jobs_the_long_one(monkeypatch):
    """The device-side choice: a job as long as its longest call gets the 768-anchor ring, a job of many short calls the
    512-anchor one (more calls in flight per CU); forced either way the four arrays are the same."""
    case = gen_chain(2500, 77)
    want = O.chain_oracle(*case, nthreads=8)
    for ring in ("short", "long", None):
        if ring:
            monkeypatch.setenv("GBX_CHAIN_RING", ring)
        else:
            monkeypatch.delenv("GBX_CHAIN_RING", raising=False)
        assert_same(chain_host(*case), want)


@pytest.mark.parametrize("mode", ["GBX_CHAIN_NOSPLIT", "GBX_CHAIN_WIDE"])
@pytest.mark.parametrize("name", ["cuts", "realistic", "multiseg"])
def test_goldens_without_cuts_and_on_the_general_path(name, mode, monkeypatch):
    """The same goldens with every call as one job (the cuts are an optimisation, not a semantic), and with every job on
    the general 64-bit / multi-segment path (a piece under one strand / reference id normally takes the narrow one)."""
    monkeypatch.setenv(mode, "1")
    case, g = load_chain_golden(name)
    assert_same(chain_host(*case), [g[:, 0], g[:, 1], g[:, 2], g[:, 3]])


def test_realistic_calls_are_cut_into_jobs_and_equal_the_oracle():
    """400 calls of the realistic generator (20 % isolated hits, repeat copies on other strands / references): identical
    to the oracle, and the evaluated-pair count (the benchmark's work unit) equals the oracle's too."""
    import torch
    case = gen_chain(400, 4242, realistic=True)
    want = O.chain_oracle(*case, nthreads=8, return_pairs=True)
    d = DeviceChainBatch(*case, torch.device("cuda:0"))
    d.run(torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert_same(d.results(), want[:4])
    assert d.evaluated_pairs() == int(want[4])


def test_extreme_cut_structures():
    """Every anchor isolated (a cut at every anchor: only one per 64 is taken, the job list stays bounded), one long run
    followed by thousands of isolated anchors, cuts exactly at the 64-anchor block boundaries, 3 000 one-anchor calls, and
    an unsorted call between sorted ones (never cut)."""
    rng = np.random.default_rng(17)
    span = (np.uint64(15) << np.uint64(32))

    def call(x, q):
        return (np.asarray(x, dtype=np.uint64), span | np.asarray(q, dtype=np.uint64), (np.float32(15.0), 5000, 5000, 500, 1))
    iso = call(1000 + np.arange(5000, dtype=np.uint64) * np.uint64(6000), 100 + np.arange(5000) * 7)
    run = np.cumsum(rng.integers(1, 40, 3000)).astype(np.uint64) + np.uint64(1000)
    tail = run[-1] + np.uint64(7000) + np.arange(4000, dtype=np.uint64) * np.uint64(5001)
    mixed = call(np.concatenate([run, tail]), np.concatenate([100 + np.cumsum(rng.integers(1, 40, 3000)), 500 + np.arange(4000) * 3]))
    blocks = []
    for b in range(40):                                      # 40 runs of exactly 64 anchors, 9 000 apart
        blocks.append(np.uint64(1000 + b * 9000) + np.sort(rng.integers(0, 2000, 64)).astype(np.uint64))
    aligned = call(np.concatenate(blocks), 100 + np.arange(64 * 40) * 5)
    ux = np.cumsum(rng.integers(1, 40, 500)).astype(np.uint64) + np.uint64(1000)
    ux[[100, 300]] = ux[[300, 100]]                          # not sorted: one job, the kernel walks st itself
    unsorted = call(ux, 100 + np.arange(500) * 11)
    singles = [call([1000 + 13 * k], [200]) for k in range(3000)]
    case = chain_pack([iso, mixed, unsorted, aligned] + singles + [iso])
    want = O.chain_oracle(*case, nthreads=8)
    assert_same(chain_host(*case), want)


@pytest.mark.parametrize("seed", [1, 2])
def test_case_families_vs_oracle(seed):
    for case in chain_cases(seed=seed).values():
        assert_same(chain_host(*case), O.chain_oracle(*case))


def test_empty_and_tiny_calls():
    rng = np.random.default_rng(3)
    calls = [_chain_call(rng, n) for n in (1, 1, 2, 63, 64, 65, 128, 129)]
    off, ax, ay, hdr = chain_pack(calls)
    # splice two empty calls in
    off = np.concatenate([[0, 0], off[1:4], [off[3]], off[4:]]).astype(np.int64)
    hdr = np.concatenate([hdr[:1], hdr[:3], hdr[:1], hdr[3:]])
    assert_same(chain_host(off, ax, ay, hdr), O.chain_oracle(off, ax, ay, hdr))


def test_nullable_outputs():
    case = gen_chain(20, 7)
    s, p, t, k = chain_host(*case, want_target=False, want_peak=False)
    ws, wp, _, _ = O.chain_oracle(*case)
    assert t is None and k is None and np.array_equal(s, ws) and np.array_equal(p, wp)


def test_generated_calls_and_device_entry():
    import torch
    case = gen_chain(300, 2001)
    want = O.chain_oracle(*case, nthreads=8)
    assert_same(chain_host(*case), want)
    d = DeviceChainBatch(*case, torch.device("cuda:0"))
    s = torch.cuda.current_stream().cuda_stream
    d.run(s)
    torch.cuda.synchronize()
    first = [a.copy() for a in d.results()]
    d.run(s)                       # outputs double as working state: a re-run must give the same answer
    torch.cuda.synchronize()
    assert_same(d.results(), first)
    assert_same(first, want)


def test_host_entry_staged_transfers(monkeypatch):
    """The staged path of the host entry (pinned slabs, upload workers, downloader thread; normally for inputs of
    8 MiB and more) gives the same four arrays as the pageable path and the oracle."""
    case = gen_chain(120, 77)
    want = O.chain_oracle(*case, nthreads=8)
    monkeypatch.setenv("GBX_HOST_STAGE_MIN", "0")
    assert_same(chain_host(*case), want)
    monkeypatch.setenv("GBX_HOST_THREADS", "1")
    assert_same(chain_host(*case), want)
    monkeypatch.setenv("GBX_HOST_PAGEABLE", "1")
    assert_same(chain_host(*case), want)


@pytest.mark.parametrize("top", ["1", "5", "64"])
def test_host_entry_longest_calls_in_a_launch_of_their_own(top, monkeypatch):
    """A large staged call uploads and launches its longest calls ahead of everything else, runs the rest as a second launch
    and downloads that one's results while the longest calls are still at work (capi_chain.hip; default from 4 Mi anchors on):
    forced onto small jobs here - every output array, with and without the optional ones, call lists that fall apart into
    many jobs, fewer calls than the split takes, small download pieces."""
    monkeypatch.setenv("GBX_HOST_STAGE_MIN", "0")
    monkeypatch.setenv("GBX_CHAIN_SPLIT_MIN", "1")
    monkeypatch.setenv("GBX_CHAIN_SPLIT_TOP", top)
    case = gen_chain(150, 4242)
    want = O.chain_oracle(*case, nthreads=8)
    assert_same(chain_host(*case), want)
    s, p, t, k = chain_host(*case, want_target=False, want_peak=False)
    assert t is None and k is None and np.array_equal(s, want[0]) and np.array_equal(p, want[1])
    s, p, t, k = chain_host(*case, want_target=True, want_peak=False)
    assert k is None and np.array_equal(s, want[0]) and np.array_equal(p, want[1]) and np.array_equal(t, want[2])
    monkeypatch.setenv("GBX_HOST_DOWN_PIECE", "4096")
    assert_same(chain_host(*case), want)
    for name in ("cuts", "realistic", "multiseg"):            # calls that fall apart into jobs: a job table of pieces
        gcase, g = load_chain_golden(name)
        assert_same(chain_host(*gcase), [g[:, 0], g[:, 1], g[:, 2], g[:, 3]])
    few = gen_chain(3, 9)                                      # fewer calls than the split takes: one launch as before
    assert_same(chain_host(*few), O.chain_oracle(*few))
    monkeypatch.setenv("GBX_CHAIN_SPLIT_TOP", "0")
    assert_same(chain_host(*case), want)
