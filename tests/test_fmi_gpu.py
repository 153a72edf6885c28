"""GPU parity tests for fmi: the HIP SMEM kernels (through the C-ABI) vs oracle/fmi_oracle.c, bit-exact on every field of every
record (rid, m, n, k, l, s) and on the per-read offsets."""
import numpy as np
import pytest

from genomicsbench_amd.datagen import gen_fmi_genome, gen_fmi_reads
from genomicsbench_amd.fmi import DeviceFmi, FmiReadSet, build_index, default_params, smem_host
from oracle import oracle_py as O

pytestmark = pytest.mark.gpu
FIELDS = ("rid", "m", "n", "k", "l", "s")


def assert_same(got, want):
    (g, goff), (w, woff) = got, want
    assert np.array_equal(goff, woff), "per-read offsets differ, first at read %d" % int(np.nonzero(goff != woff)[0][0] - 1)
    assert len(g) == len(w)
    for f in FIELDS:
        if not np.array_equal(g[f], w[f]):
            k = int(np.nonzero(g[f] != w[f])[0][0])
            raise AssertionError("%s differs in %d records; first at %d (read %d): got %d want %d"
                                 % (f, int((g[f] != w[f]).sum()), k, int(w["rid"][k]), g[f][k], w[f][k]))


@pytest.fixture(scope="module")
def small():
    g = gen_fmi_genome(300_000, 6001)
    return g, build_index(g)


def test_frozen_fixture():
    """tests/golden/fmi_small.*: ragged reads against a genome with a three-copy repeat (one copy reversed)."""
    from util import load_fmi_golden
    g, rs, want = load_fmi_golden()
    for wide in ("0", "1"):
        import os
        os.environ["GBX_FMI_WIDE"] = wide
        try:
            out, off = smem_host(build_index(g), rs, default_params(19))
        finally:
            del os.environ["GBX_FMI_WIDE"]
        got = np.stack([out[f].astype(np.int64) for f in FIELDS], 1)
        assert np.array_equal(got, want), "instance wide=%s" % wide


def test_reads_of_the_bench_generator(small):
    g, idx = small
    rs = gen_fmi_reads(g, 3000, 6002)
    P = default_params(19)
    assert_same(smem_host(idx, rs, P), O.fmi_oracle(idx, rs, P, nthreads=8))


@pytest.mark.parametrize("min_seed_len", [10, 19, 30])
def test_seed_lengths_and_split_parameters(small, min_seed_len):
    g, idx = small
    rs = gen_fmi_reads(g, 1000, 77 + min_seed_len)
    P = default_params(min_seed_len)
    assert_same(smem_host(idx, rs, P), O.fmi_oracle(idx, rs, P, nthreads=8))
    P.split_width, P.max_mem_intv = 3, 5
    assert_same(smem_host(idx, rs, P), O.fmi_oracle(idx, rs, P, nthreads=8))


def test_repetitive_genome_many_hits_and_reseeding():
    """a genome made of diverged copies of one element: intervals with hundreds of hits, long prev[] arrays, the
    re-seeding round busy."""
    rng = np.random.default_rng(5)
    elem = rng.integers(0, 4, 700).astype(np.uint8)
    parts = []
    for _ in range(60):
        c = elem.copy()
        hit = rng.random(700) < 0.02
        c[hit] = (c[hit] + 1) % 4
        parts += [c, rng.integers(0, 4, int(rng.integers(5, 60))).astype(np.uint8)]
    g = np.concatenate(parts)
    idx = build_index(g)
    reads = []
    for _ in range(600):
        p = int(rng.integers(0, len(g) - 151))
        r = g[p:p + 151].copy()
        if rng.random() < 0.5:
            r = 3 - r[::-1]
        hit = rng.random(151) < 0.015
        r[hit] = (r[hit] + 1) % 4
        reads.append(r)
    rs = FmiReadSet.fixed(np.array(reads))
    P = default_params(19)
    want = O.fmi_oracle(idx, rs, P, nthreads=8, return_stats=True)
    assert want[3][1] > 0, "the re-seeding round must be exercised"
    assert_same(smem_host(idx, rs, P), want[:2])


def test_ragged_reads_ambiguous_bases_and_empty_reads(small):
    g, idx = small
    rng = np.random.default_rng(9)
    lens = [0, 1, 2, 18, 19, 20, 21, 40, 151, 152, 255, 256, 300, 0, 151]
    chunks, off = [], []
    at = 0
    for L in lens * 8:
        p = int(rng.integers(0, len(g) - max(L, 1)))
        r = g[p:p + L].copy()
        if L and rng.random() < 0.5:
            r[rng.integers(0, L, max(1, L // 30))] = 4
        chunks.append(r)
        off.append(at)
        at += L + int(rng.integers(0, 3))                    # gaps between reads: any offset layout
        chunks.append(np.full(at - off[-1] - L, 4, dtype=np.uint8))
    chunks.append(np.full(200, 4, dtype=np.uint8))           # reads made of N only
    off.append(at)
    rs = FmiReadSet(np.concatenate(chunks), np.array(off), np.array(lens * 8 + [200]))
    P = default_params(19)
    assert_same(smem_host(idx, rs, P), O.fmi_oracle(idx, rs, P))


def _long_reads(g, L, n, rng):
    reads = []
    for _ in range(n):
        p = int(rng.integers(0, len(g) - L))
        r = g[p:p + L].copy()
        hit = rng.random(L) < 0.02
        r[hit] = (r[hit] + 1) % 4
        if rng.random() < 0.3:
            r[rng.integers(0, L, 3)] = 4                       # ambiguous bases: the in-place and the staged path must agree on them
        reads.append(r)
    return FmiReadSet.fixed(np.array(reads))


@pytest.mark.parametrize("wide", ["0", "1"])
@pytest.mark.parametrize("L", [7650, 7700, 8100, 8200, 9000, 9999])
def test_reads_beyond_the_lds_staging_are_read_in_place(small, monkeypatch, L, wide):
    """Reads of up to ~8000 bases are staged in LDS four bits per base; longer ones (the reference accepts 9999,
    fmi.cpp:93) are read in place (fmi_smem_kernel<false, *>).  The lengths straddle the switch (the whole LDS request -
    staged reads + prev[] slab + static - must fit 64 KB); both SA-row widths."""
    g, idx = small
    monkeypatch.setenv("GBX_FMI_WIDE", wide)
    rs = _long_reads(g, L, 12, np.random.default_rng(L))
    P = default_params(19)
    assert_same(smem_host(idx, rs, P, out_cap=12 * 1500), O.fmi_oracle(idx, rs, P, nthreads=8))


@pytest.mark.parametrize("wide", ["0", "1"])
def test_in_place_path_on_ordinary_reads(small, monkeypatch, wide):
    """GBX_FMI_INPLACE=1 forces the in-place instance for any length: the ragged / N-rich set and 151-bp reads."""
    g, idx = small
    monkeypatch.setenv("GBX_FMI_WIDE", wide)
    monkeypatch.setenv("GBX_FMI_INPLACE", "1")
    P = default_params(19)
    rs = gen_fmi_reads(g, 3000, 6011)
    assert_same(smem_host(idx, rs, P), O.fmi_oracle(idx, rs, P, nthreads=8))
    rs = _long_reads(g, 700, 30, np.random.default_rng(3))
    assert_same(smem_host(idx, rs, P, out_cap=30 * 400), O.fmi_oracle(idx, rs, P, nthreads=8))


def test_long_reads_on_the_lds_path(small):
    g, idx = small
    rng = np.random.default_rng(10)
    for L in (700, 2500):
        reads = []
        for _ in range(40):
            p = int(rng.integers(0, len(g) - L))
            r = g[p:p + L].copy()
            hit = rng.random(L) < 0.02
            r[hit] = (r[hit] + 1) % 4
            reads.append(r)
        rs = FmiReadSet.fixed(np.array(reads))
        P = default_params(19)
        assert_same(smem_host(idx, rs, P, out_cap=40 * 400), O.fmi_oracle(idx, rs, P))


def test_device_entry_rerun_and_extension_count(small):
    import torch
    g, idx = small
    rs = gen_fmi_reads(g, 5000, 6003)
    P = default_params(19)
    want = O.fmi_oracle(idx, rs, P, nthreads=8, return_stats=True)
    d = DeviceFmi(idx, rs, torch.device("cuda:0"), P)
    for _ in range(2):                                        # a second run on the same workspace gives the same
        d.run(torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert_same(d.results(), want[:2])
        assert d.extensions() == want[2]


def test_reads_with_more_smems_than_a_slot_holds_run_again_with_larger_slots(small, monkeypatch):
    """A read's SMEMs wait for the pack pass in a fixed-size slot (48 records for 151-bp reads at minSeedLen 19; the bench's
    reads use 18 at most).  With slots of 6 records most reads overflow: the host entry must notice and run the job again
    with slots of the size the worst read asked for."""
    g, idx = small
    rs = gen_fmi_reads(g, 1500, 6005)
    P = default_params(19)
    want = O.fmi_oracle(idx, rs, P, nthreads=8)
    assert np.diff(want[1]).max() > 6
    monkeypatch.setenv("GBX_FMI_RAW_CAP", "6")
    assert_same(smem_host(idx, rs, P), want)


def test_unmappable_long_reads_with_short_seeds_need_more_than_one_resize(monkeypatch):
    """Found by scripts/fuzz_gpu.py: 400-base reads at minSeedLen 8 - a random (unmappable) read yields 286 SMEMs, more than
    its default slot of 216; the count the overflowing pass reports is only a lower bound (its re-seeding round sees the
    kept records only), so one resize to that count is not enough."""
    g = gen_fmi_genome(200_000, 4242 + 200_000)
    idx = build_index(g)
    rs = gen_fmi_reads(g, 500, 803022771, read_len=400)
    P = default_params(8)
    want = O.fmi_oracle(idx, rs, P, nthreads=8)
    assert np.diff(want[1]).max() > 216
    assert_same(smem_host(idx, rs, P, out_cap=200 * 500), want)
    monkeypatch.setenv("GBX_FMI_RAW_CAP", "40")
    assert_same(smem_host(idx, rs, P, out_cap=200 * 500), want)


def test_device_entry_reports_reads_that_do_not_fit_their_slot():
    """The device entry cannot resize its caller's workspace: it flags the job (gbx_fmi_overflow) instead."""
    import torch
    g = gen_fmi_genome(200_000, 4242 + 200_000)
    idx = build_index(g)
    rs = gen_fmi_reads(g, 500, 803022771, read_len=400)        # one read of this set yields 286 SMEMs at minSeedLen 8: slots hold 216
    d = DeviceFmi(idx, rs, torch.device("cuda:0"), default_params(8), out_cap=200 * 500)
    d.run(torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert d.overflow() > 216
    with pytest.raises(RuntimeError, match="slot"):
        d.results()


def test_device_entry_flags_a_read_longer_than_max_read_len(small):
    import torch
    from genomicsbench_amd._native import GbxError
    g, idx = small
    rs = gen_fmi_reads(g, 64, 6006)
    d = DeviceFmi(idx, rs, torch.device("cuda:0"))
    d.max_len = 100                                           # the caller lies about its longest read (151)
    d.run(torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    with pytest.raises(GbxError, match="exceeds max_read_len"):
        d.overflow()


def test_output_capacity_too_small_is_reported(small):
    g, idx = small
    rs = gen_fmi_reads(g, 200, 6004)
    with pytest.raises(RuntimeError, match="do not fit"):
        smem_host(idx, rs, default_params(19), out_cap=100)


def test_index_built_on_the_gpu_equals_the_cpu_build():
    import torch
    g = gen_fmi_genome(200_000, 11)
    a, b = build_index(g), build_index(g, device="cuda:0").host()
    assert a.count == b.count and a.sentinel_index == b.sentinel_index and a.ref_seq_len == b.ref_seq_len
    assert np.array_equal(a.cp_occ.view(np.uint8), b.cp_occ.view(np.uint8))


def test_host_index_cache_is_keyed_by_content_not_by_address():
    """gbx_fmi_smem_host keeps the device copy of an index between calls.  A caller that reuses the same host buffer for
    another index of the same length (rebuilt in place) must get that index's SMEMs, not the cached one's; the cache
    release is safe beside other entries and repeated."""
    from genomicsbench_amd import _native as N
    from genomicsbench_amd.fmi import FmiIndex
    ga, gb = gen_fmi_genome(120_000, 6101), gen_fmi_genome(120_000, 6102)
    ia, ib = build_index(ga), build_index(gb)
    assert ia.ref_seq_len == ib.ref_seq_len
    P = default_params(19)
    ra, rb = gen_fmi_reads(ga, 800, 6103), gen_fmi_reads(gb, 800, 6104)
    assert_same(smem_host(ia, ra, P), O.fmi_oracle(ia, ra, P, nthreads=4))
    shared = FmiIndex(ia.ref_seq_len, ib.count, ib.sentinel_index, ia.cp_occ)     # A's buffer ...
    shared.cp_occ[:] = ib.cp_occ                                                   # ... now holds B's tables: same address
    assert_same(smem_host(shared, rb, P), O.fmi_oracle(ib, rb, P, nthreads=4))
    for _ in range(2):
        N.check(N.lib().gbx_fmi_host_release())
    assert_same(smem_host(shared, rb, P), O.fmi_oracle(ib, rb, P, nthreads=4))
    for seed in range(6):                                                          # more indexes than the cache keeps idle
        g = gen_fmi_genome(40_000 + 1000 * seed, 6200 + seed)
        ix, rs = build_index(g), gen_fmi_reads(g, 100, 6300 + seed)
        assert_same(smem_host(ix, rs, P), O.fmi_oracle(ix, rs, P, nthreads=4))
    N.check(N.lib().gbx_fmi_host_release())


def test_threads_that_meet_an_index_for_the_first_time_together():
    """Round 5: the device copy of an index is uploaded and re-laid OUTSIDE the cache's lock behind a place-holder entry; callers that
    ask for the very index under construction wait for it, callers with another index build theirs beside it, and a release in between
    leaves the entries in use alone.  Six threads, two indexes, twice (the second time after a release)."""
    import threading
    from genomicsbench_amd import _native as N
    ga, gb = gen_fmi_genome(300_000, 7101), gen_fmi_genome(260_000, 7102)
    ia, ib = build_index(ga), build_index(gb)
    P = default_params(19)
    sets = [(ia, gen_fmi_reads(ga, 400, 7200 + k)) if k % 2 == 0 else (ib, gen_fmi_reads(gb, 400, 7200 + k)) for k in range(6)]
    want = [O.fmi_oracle(ix, rs, P, nthreads=4) for ix, rs in sets]
    for rnd in range(2):
        N.check(N.lib().gbx_fmi_host_release())
        got, err = [None] * len(sets), []

        def work(k):
            try:
                got[k] = smem_host(sets[k][0], sets[k][1], P)
                if k == 3:
                    N.check(N.lib().gbx_fmi_host_release())                        # beside calls that hold their entries
            except Exception as e:                                                 # noqa: BLE001
                err.append((k, repr(e)))

        th = [threading.Thread(target=work, args=(k,)) for k in range(len(sets))]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert not err, err
        for k in range(len(sets)):
            assert_same(got[k], want[k])
    N.check(N.lib().gbx_fmi_host_release())
