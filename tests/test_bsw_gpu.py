"""GPU parity tests for bsw: HIP kernels (through the C-ABI) vs the oracle, bit-exact on all six outputs."""
import numpy as np
import pytest

from cases import adversarial_bsw, edge_bsw
from genomicsbench_amd import _native as N
from genomicsbench_amd.bsw import BandedPairWiseSW, BswBatch, DeviceBswBatch, extend_host, fill_scmat, make_params
from genomicsbench_amd.datagen import gen_bsw
from oracle import oracle_py as O
from util import load_bsw_golden

pytestmark = pytest.mark.gpu
FIELDS = N.BSW_RESULT_FIELDS


def assert_same(got, want, batch=None):
    if not np.array_equal(got, want):
        rows = np.nonzero((got != want).any(1))[0]
        k = int(rows[0])
        msg = "%d/%d pairs differ; first k=%d got=%s want=%s" % (len(rows), len(want), k, got[k], want[k])
        if batch is not None:
            msg += " (len1=%d len2=%d h0=%d)" % (batch.len1[k], batch.len2[k], batch.h0[k])
        raise AssertionError(msg)


@pytest.fixture(params=["0", "1", "2"])
def classmode(request, monkeypatch):
    """The library picks fewer, wider query classes for small jobs (GBX_BSW_CLASSMODE: 0 = all twenty class
    kernels, 1 = six, 2 = three); the small parity cases run in every mode so that each kernel is covered."""
    monkeypatch.setenv("GBX_BSW_CLASSMODE", request.param)
    monkeypatch.setenv("GBX_BSW_DIRECT", "0")        # small plain jobs would otherwise skip the class kernels altogether
    return request.param


@pytest.mark.parametrize("name", ["realistic", "adversarial", "edge"])
def test_goldens(name, classmode):
    b, scalar, avx2 = load_bsw_golden(name)
    got = extend_host(make_params(), b)
    assert_same(got, scalar, b)
    assert np.array_equal(got[:, [0, 1, 3, 5]], avx2[:, [0, 1, 3, 5]])


def test_edge_cases_including_empty_sequences(classmode):
    b = edge_bsw()
    assert_same(extend_host(make_params(), b), O.bsw_oracle(make_params(), b, 4), b)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_adversarial_random(seed, classmode):
    b = adversarial_bsw(6000, seed)
    assert_same(extend_host(make_params(), b), O.bsw_oracle(make_params(), b, 4), b)


def test_long_queries_all_kernel_classes(classmode):
    b = adversarial_bsw(300, 9, max_q=3000, max_t=4000)
    assert_same(extend_host(make_params(), b), O.bsw_oracle(make_params(), b, 4), b)


@pytest.mark.parametrize("kw", [
    dict(o_del=5, e_del=2, o_ins=7, e_ins=3, zdrop=50, end_bonus=9, w=37, mat=fill_scmat(2, 5, -2)),
    dict(zdrop=0, w=5),
    dict(w=1000, zdrop=10),
    dict(o_del=0, e_del=1, o_ins=0, e_ins=1, mat=fill_scmat(3, 1, 0)),
])
def test_non_default_scoring(kw, classmode):
    p = make_params(**kw)
    b = adversarial_bsw(3000, 21)
    assert_same(extend_host(p, b), O.bsw_oracle(p, b, 4), b)


def test_bsw_small_config_100k():
    """BASELINE config[0] workload (bsw 'small', 100k pairs) against the oracle (default class mode for that size)."""
    b = gen_bsw(100_000, 1001)
    assert_same(extend_host(make_params(), b), O.bsw_oracle(make_params(), b, 8), b)


def test_seqpair_dropin_matches_flat_entry():
    b = gen_bsw(5000, 3)
    pairs = np.zeros(b.n, dtype=N.SEQPAIR_DTYPE)
    pairs["idr"], pairs["idq"], pairs["id"] = b.idr, b.idq, np.arange(b.n)
    pairs["len1"], pairs["len2"], pairs["h0"] = b.len1, b.len2, b.h0
    for f in FIELDS:
        pairs[f] = -1
    mat = fill_scmat(1, 4, -1)
    sw = BandedPairWiseSW(6, 1, 6, 1, 100, 5, mat, 1, 4, 1)
    sw.getScores16(pairs, b.ref, b.qer, b.n, 1, 100)
    want = O.bsw_oracle(make_params(), b, 4)
    got = np.stack([pairs[f] for f in FIELDS], axis=1)
    assert_same(got, want, b)


def test_device_resident_entry_and_rerun_idempotence():
    import torch
    b = gen_bsw(20000, 8)
    d = DeviceBswBatch(b, torch.device("cuda:0"))
    p = make_params()
    s = torch.cuda.current_stream().cuda_stream
    d.run(p, s)
    torch.cuda.synchronize()
    first = d.results().copy()
    d.out.fill_(-7)
    d.run(p, s)
    torch.cuda.synchronize()
    assert_same(d.results(), first)
    assert_same(first, O.bsw_oracle(p, b, 4), b)


def test_permutation_and_shard_equivalence_large():
    """Size-independent properties at the bsw 'large' scale: results do not depend on pair order or sharding."""
    b = gen_bsw(400_000, 1002)
    p = make_params()
    full = extend_host(p, b)
    rng = np.random.default_rng(0)
    perm = rng.permutation(b.n)
    pb = BswBatch(b.ref, b.qer, b.idr[perm], b.idq[perm], b.len1[perm], b.len2[perm], b.h0[perm])
    assert_same(extend_host(p, pb), full[perm])
    cut = b.n // 3
    parts = np.concatenate([extend_host(p, b.slice(0, cut)), extend_host(p, b.slice(cut, b.n))])
    assert_same(parts, full)
    # anchor a sample on the oracle
    idx = rng.choice(b.n, 20000, replace=False)
    sb = BswBatch(b.ref, b.qer, b.idr[idx], b.idq[idx], b.len1[idx], b.len2[idx], b.h0[idx])
    assert_same(full[idx], O.bsw_oracle(p, sb, 8), sb)


def test_host_entry_pipeline_chunks(monkeypatch):
    """gbx_bsw_extend_host uploads and computes in chunks of pairs: many small chunks over permuted
    (non-monotone) arena offsets give the same results as one chunk and as the oracle."""
    b = gen_bsw(6000, 21)
    perm = np.random.default_rng(5).permutation(b.n)
    pb = BswBatch(b.ref, b.qer, b.idr[perm], b.idq[perm], b.len1[perm], b.len2[perm], b.h0[perm])
    p = make_params()
    want = O.bsw_oracle(p, pb, 4)
    for chunk in ("64", "1000", "100000"):
        monkeypatch.setenv("GBX_BSW_HOST_CHUNK", chunk)
        assert_same(extend_host(p, pb), want, pb)


def test_concurrent_host_threads():
    """The reference calls one aligner object per OpenMP thread concurrently (main_banded.cpp:279-289): host
    threads calling the C-ABI at the same time get the same results as one after the other."""
    import threading
    p = make_params()
    batches = [gen_bsw(30000, 100 + t) for t in range(4)]
    want = [extend_host(p, b) for b in batches]
    for _ in range(3):
        got = [None] * len(batches)

        def work(t):
            got[t] = extend_host(p, batches[t])

        th = [threading.Thread(target=work, args=(t,)) for t in range(len(batches))]
        for x in th:
            x.start()
        for x in th:
            x.join()
        for t in range(len(batches)):
            assert_same(got[t], want[t], batches[t])


def test_host_entry_staged_small_and_error_in_late_chunk(monkeypatch):
    """Staging forced on a small input (one and many chunks), and an invalid pair in the last chunk: the call
    returns the argument error, names the pair, and the next call works."""
    b = gen_bsw(4000, 33)
    p = make_params()
    want = O.bsw_oracle(p, b, 4)
    monkeypatch.setenv("GBX_HOST_STAGE_MIN", "0")
    assert_same(extend_host(p, b), want, b)
    monkeypatch.setenv("GBX_BSW_HOST_CHUNK", "512")
    assert_same(extend_host(p, b), want, b)
    bad = BswBatch(b.ref, b.qer, b.idr.copy(), b.idq, b.len1, b.len2, b.h0)
    bad.idr[3999] = b.ref.size            # beyond the arena
    with pytest.raises(N.GbxError) as e:
        extend_host(p, bad)
    assert e.value.code == N.GBX_ERR_ARG and "3999" in str(e.value)
    assert_same(extend_host(p, b), want, b)


def test_seqpair_dropin_strided_slots():
    """The reference driver's buffer layout (main_banded.cpp:56-58,160-172): pair k owns a fixed-stride slot, the
    arenas are mostly holes.  The SeqPair entry packs the bases before upload; results as for the packed layout."""
    b = gen_bsw(6000, 44)
    SR, SQ = 2048, 256
    ref = np.full(b.n * SR, 9, dtype=np.uint8)
    qer = np.full(b.n * SQ, 9, dtype=np.uint8)
    pairs = np.zeros(b.n, dtype=N.SEQPAIR_DTYPE)
    for k in range(b.n):
        ref[k * SR:k * SR + b.len1[k]] = b.ref[b.idr[k]:b.idr[k] + b.len1[k]]
        qer[k * SQ:k * SQ + b.len2[k]] = b.qer[b.idq[k]:b.idq[k] + b.len2[k]]
    pairs["idr"], pairs["idq"], pairs["id"] = np.arange(b.n) * SR, np.arange(b.n) * SQ, np.arange(b.n)
    pairs["len1"], pairs["len2"], pairs["h0"] = b.len1, b.len2, b.h0
    for f in FIELDS:
        pairs[f] = -1
    sw = BandedPairWiseSW(6, 1, 6, 1, 100, 5, fill_scmat(1, 4, -1), 1, 4, 1)
    sw.getScores16(pairs, ref, qer, b.n, 1, 100)
    got = np.stack([pairs[f] for f in FIELDS], axis=1)
    assert_same(got, O.bsw_oracle(make_params(), b, 4), b)
    assert np.array_equal(pairs["idr"], np.arange(b.n) * SR)       # the caller's records keep their offsets


def test_direct_launch_for_small_plain_jobs(monkeypatch):
    """Host-entry jobs of up to 16 Ki pairs whose queries are 1..256 long (no empty sequence) run as a single
    launch of the 8x16 or 16x16 kernel in input order, without the binning passes: same results as the class path and the
    oracle; a job with one empty or long query takes the class path."""
    p = make_params()
    for n, seed in ((1, 5), (63, 6), (512, 7), (5000, 8), (16384, 9)):
        b = gen_bsw(n, seed)
        assert b.len2.max() <= 256 and b.len2.min() >= 1 and b.len1.min() >= 1
        want = O.bsw_oracle(p, b, 4)
        monkeypatch.setenv("GBX_BSW_DIRECT", "1")
        assert_same(extend_host(p, b), want, b)
        monkeypatch.setenv("GBX_BSW_DIRECT", "0")
        assert_same(extend_host(p, b), want, b)
    monkeypatch.setenv("GBX_BSW_DIRECT", "1")
    e = edge_bsw()                                     # empty sequences and long queries: not plain
    assert_same(extend_host(p, e), O.bsw_oracle(p, e, 4), e)
    pq = make_params(o_del=5, e_del=2, o_ins=7, e_ins=3, zdrop=50, end_bonus=9, w=37, mat=fill_scmat(2, 5, -2))
    b = gen_bsw(3000, 10)
    assert_same(extend_host(pq, b), O.bsw_oracle(pq, b, 4), b)      # asymmetric gaps through the direct launch


# ---- lane path (bsw_lane_kernel: one pair per lane, pairs sorted by (query length, seed score)) -------------------------
# Large jobs take it by default (tests/test_fullsize_gpu.py runs the 2 M-pair job through it); GBX_BSW_LANE=1 forces it
# on the small parity sets, where pairs that do not qualify (queries over 159, scores of 8192 and more, scorings
# outside six bits) share the call with the row kernels.
@pytest.fixture
def lane(monkeypatch):
    monkeypatch.setenv("GBX_BSW_LANE", "1")
    monkeypatch.setenv("GBX_BSW_DIRECT", "0")


@pytest.mark.parametrize("name", ["realistic", "adversarial", "edge"])
def test_lane_goldens(name, lane):
    b, scalar, avx2 = load_bsw_golden(name)
    got = extend_host(make_params(), b)
    assert_same(got, scalar, b)
    assert np.array_equal(got[:, [0, 1, 3, 5]], avx2[:, [0, 1, 3, 5]])


def test_lane_edge_cases(lane):
    b = edge_bsw()
    assert_same(extend_host(make_params(), b), O.bsw_oracle(make_params(), b, 4), b)


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_lane_adversarial_random(seed, lane):
    b = adversarial_bsw(6000, seed)
    assert_same(extend_host(make_params(), b), O.bsw_oracle(make_params(), b, 4), b)


def test_lane_mixed_with_long_queries(lane):
    b = adversarial_bsw(600, 9, max_q=3000, max_t=4000)
    assert_same(extend_host(make_params(), b), O.bsw_oracle(make_params(), b, 4), b)


@pytest.mark.parametrize("kw", [
    dict(o_del=5, e_del=2, o_ins=7, e_ins=3, zdrop=50, end_bonus=9, w=37, mat=fill_scmat(2, 5, -2)),
    dict(zdrop=0, w=5),
    dict(w=1000, zdrop=10),
    dict(o_del=0, e_del=1, o_ins=0, e_ins=1, mat=fill_scmat(3, 1, 0)),
    dict(mat=fill_scmat(31, 32, -7)),             # the corners of the six-bit score fields
    dict(mat=fill_scmat(40, 3, -1)),              # does not fit six bits: every pair stays on the row kernels
])
def test_lane_non_default_scoring(kw, lane):
    p = make_params(**kw)
    b = adversarial_bsw(3000, 21)
    assert_same(extend_host(p, b), O.bsw_oracle(p, b, 4), b)


@pytest.mark.parametrize("env", [{}, {"GBX_BSW_PREP": "0"}, {"GBX_BSW_SKIP_ROWS": "0"}, {"GBX_BSW_PACKED_LANES": "0"}, {"GBX_BSW_SPLIT_PREP": "0"},
                                 {"GBX_COPY_STREAMS": "1", "GBX_DOWN_STREAM": "0"}])
def test_lane_pipelined_chunks_with_and_without_row_kernel_pairs(lane, monkeypatch, env):
    """The pipelined host call prepares a chunk (unpacking, classify, the lane sort) on streams of their own and leaves out
    the row-kernel classes of a chunk whose pairs all go to the lane kernels, which then read the packed bases as they were
    uploaded: chunks with none, with a few (long queries, in the third chunk only) and a ragged last one give the oracle's
    results, with each of the shortcuts switched off too."""
    monkeypatch.setenv("GBX_HOST_STAGE_MIN", "0")
    monkeypatch.setenv("GBX_BSW_HOST_CHUNK", "2048")
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    p = make_params()
    b = gen_bsw(9000, 41)
    want = O.bsw_oracle(p, b, 8)
    assert_same(extend_host(p, b), want, b)
    a = adversarial_bsw(40, 3, max_q=900, max_t=1200)          # row-kernel pairs, spliced in at 4100..4139
    ref = np.concatenate([b.ref, a.ref]); qer = np.concatenate([b.qer, a.qer])
    ins = lambda x, y: np.concatenate([x[:4100], y, x[4100:]])
    m = BswBatch(ref, qer, ins(b.idr, a.idr + b.ref.size), ins(b.idq, a.idq + b.qer.size), ins(b.len1, a.len1), ins(b.len2, a.len2), ins(b.h0, a.h0))
    assert_same(extend_host(p, m), O.bsw_oracle(p, m, 8), m)


@pytest.mark.parametrize("where", ["late_slice_of_later_chunk", "pairs_reuse_first_chunks_bases"])
def test_lane_packed_chunk_followed_by_unpacked_chunk(lane, monkeypatch, where):
    """A chunk the lane kernels take whole runs on the packed bases and expands nothing; a later chunk with row-kernel
    pairs reads the byte arenas, which must then hold everything up to its own end - including what the packed chunks
    before it brought up (the call's watermark, BswChunkPrep::unp_r).  Several validation slices (32 Ki pairs each), cuts
    that are not multiples of a slice, the only row-kernel pairs in the last slice of the last chunk; and a layout whose later
    pairs point back into the first chunk's bases.  The profile shows that both forms ran in the one call."""
    monkeypatch.setenv("GBX_HOST_STAGE_MIN", "0")
    monkeypatch.setenv("GBX_BSW_HOST_CHUNK", "40000")           # cuts at 40000, 80000 (r64: 40000 = 625 * 64), slices at 32768 k
    p = make_params()
    b = gen_bsw(3 * 32768 + 5000, 43)
    a = adversarial_bsw(24, 5, max_q=900, max_t=1200)           # row-kernel pairs
    at = 3 * 32768 + 2000                                       # slice 3 (98304..), which only chunk 2 (80000..) overlaps
    ref = np.concatenate([b.ref, a.ref]); qer = np.concatenate([b.qer, a.qer])
    ins = lambda x, y: np.concatenate([x[:at], y, x[at:]])
    idr, idq = ins(b.idr, a.idr + b.ref.size), ins(b.idq, a.idq + b.qer.size)
    if where == "pairs_reuse_first_chunks_bases":
        # the 3000 pairs after the spliced ones read the bases of pairs 0..2999 (first chunk: packed, never expanded by itself)
        k = at + len(a.idr)
        idr[k:k + 3000] = b.idr[:3000]; idq[k:k + 3000] = b.idq[:3000]
        l1, l2, h0 = ins(b.len1, a.len1), ins(b.len2, a.len2), ins(b.h0, a.h0)
        l1[k:k + 3000] = b.len1[:3000]; l2[k:k + 3000] = b.len2[:3000]; h0[k:k + 3000] = b.h0[:3000]
    else:
        l1, l2, h0 = ins(b.len1, a.len1), ins(b.len2, a.len2), ins(b.h0, a.h0)
    m = BswBatch(ref, qer, idr, idq, l1, l2, h0)
    want = O.bsw_oracle(p, m, 8)
    N.profile_begin()
    got = extend_host(p, m)
    prof = N.profile_end(256)
    assert_same(got, want, m)
    assert "bsw_unpack4" in prof, sorted(prof)                   # a chunk that expanded the arenas ...
    lane_launches = sum(c for k, (ms, c) in prof.items() if k.startswith("bsw_lane_c"))
    rows_launches = sum(c for k, (ms, c) in prof.items() if k.startswith("bsw_rows") or k == "bsw_lds")
    assert lane_launches >= 6 and 0 < rows_launches, prof        # ... and three chunks of lane launches (a chunk's empty classes are not launched), row classes for one only
    assert prof["bsw_lds"][1] == 1, prof                         # (the two packed chunks launched no row class at all)


def test_lane_generated_reads_and_ragged_last_chunk(lane):
    b = gen_bsw(30_011, 77)                        # not a multiple of 64 in any length class
    assert_same(extend_host(make_params(), b), O.bsw_oracle(make_params(), b, 8), b)


@pytest.mark.parametrize("n,max_q", [(300, 100), (700, 159), (1500, 191), (2000, 255), (5000, 120), (6000, 159), (7000, 190), (9000, 250)])
def test_direct_launch_shapes(n, max_q, monkeypatch):
    """Small plain jobs (every query 1..256, <= 16 384 pairs) are one launch of a row kernel whose shape follows the job: a wavefront
    per pair (64x3 / 64x4) below 4 096 pairs, else the narrowest sixteen-lane shape that holds the longest query (16x8 .. 16x16).
    Every shape against the oracle on adversarial pairs, and the forced shapes of the tuning switch."""
    monkeypatch.delenv("GBX_BSW_DIRECT", raising=False)
    monkeypatch.setenv("GBX_COMBINE", "0")
    p = make_params()
    b = adversarial_bsw(n, 100 + max_q, max_q=max_q, max_t=max_q + 200)
    assert int(b.len2.max()) <= 256 and int(b.len2.min()) >= 1
    want = O.bsw_oracle(p, b, 4)
    N.profile_begin()
    got = extend_host(p, b)
    prof = N.profile_end(64)
    assert_same(got, want, b)
    assert len(prof) == 1 and list(prof)[0].startswith("bsw_rows_64x" if n < 4096 else "bsw_rows_16x"), prof
    for shape in ("64x4", "16x16"):
        monkeypatch.setenv("GBX_BSW_DIRECT_SHAPE", shape)
        assert_same(extend_host(p, b), want, b)
