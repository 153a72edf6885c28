"""GPU parity tests for abea: the HIP kernel (through the C-ABI) vs the oracle - identical aligned pairs, counts and
QC verdicts (the oracle restates the CPU align(), R/benchmarks/abea/src/align.c:169-548, with its float / double mix)."""
import numpy as np
import pytest

from conftest import has_gpu
from genomicsbench_amd.abea import AbeaReadSet, DeviceAbeaReadSet, KMER, align_host, make_model
from genomicsbench_amd.datagen import gen_abea
from oracle import oracle_py as O
from test_abea_cpu import one_read

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")]


def assert_same(rs, got, want):
    (go, gn), (wo, wn) = got, want
    bad = np.nonzero(gn != wn)[0]
    assert not len(bad), "n_pairs differs for %d of %d reads, first %d: got %d want %d" % (len(bad), rs.n_reads, bad[0], gn[bad[0]], wn[bad[0]])
    for r, (g, w) in enumerate(zip(rs.split_pairs(go, gn), rs.split_pairs(wo, wn))):
        if not np.array_equal(g, w):
            k = int(np.nonzero((g["ref_pos"] != w["ref_pos"]) | (g["read_pos"] != w["read_pos"]))[0][0])
            raise AssertionError("read %d: pair %d of %d differs: got %s want %s" % (r, k, len(w), g[k], w[k]))


def test_generated_reads_host_entry():
    rs = gen_abea(24, 5001)
    assert_same(rs, align_host(rs), O.abea_oracle(rs, 8))


def test_device_entry_and_cell_count():
    import torch
    rs = gen_abea(40, 9, first=100)
    d = DeviceAbeaReadSet(rs, torch.device("cuda:0"))
    s = torch.cuda.current_stream().cuda_stream
    d.run(s)
    torch.cuda.synchronize()
    wo, wn, cells = O.abea_oracle(rs, 8, True)
    assert_same(rs, d.results(), (wo, wn))
    assert d.cells(s) == cells


def test_hand_made_reads():
    """perfect signal, stays, a rejected read (garbage signal), a read barely longer than a k-mer, one event."""
    rng = np.random.default_rng(4)
    model = make_model(rng.uniform(65, 125, 4096).astype(np.float32), rng.uniform(1.2, 3.2, 4096).astype(np.float32))
    sets = []
    for k, (n, stays, noise) in enumerate([(300, 1, 0.0), (500, 2, 0.4), (450, 1, 0.2), (KMER, 1, 0.0), (KMER + 3, 3, 0.1), (1500, 1, 0.5)]):
        rs, _ = one_read("".join(rng.choice(list("ACGT"), n)), levels_noise=noise, stays=stays, seed=k, model=model)
        sets.append(rs)
    sets[2].event_mean[:] = rng.uniform(20, 200, len(sets[2].event_mean)).astype(np.float32)          # fails QC
    # concatenate into one read set
    seq_off, ev_off, arena, ev = [], [0], [], []
    pos = 0
    for rs in sets:
        seq_off.append(pos); arena.append(rs.seq_arena); pos += len(rs.seq_arena)
        ev.append(rs.event_mean); ev_off.append(ev_off[-1] + len(rs.event_mean))
    rs = AbeaReadSet(seq_off, [s.seq_len[0] for s in sets], np.concatenate(arena), ev_off, np.concatenate(ev),
                     [1.0] * len(sets), [0.0] * len(sets), model)
    want = O.abea_oracle(rs, 4)
    assert want[1][2] == 0 and want[1][0] == 300 - KMER + 1
    assert_same(rs, align_host(rs), want)


def test_non_acgt_bases_and_skips():
    """Bases outside ACGT rank as A (align.c:10-24); deleted events force skips (FROM_L moves)."""
    rs = gen_abea(6, 31)
    arena = rs.seq_arena.copy()
    arena[rs.seq_off[1] + 50:rs.seq_off[1] + 60] = ord("N")
    keep = np.ones(len(rs.event_mean), dtype=bool)
    a = int(rs.event_off[2])
    keep[a + 400:a + 430] = False                          # drop 30 events of read 2: a gap of ~17 k-mers
    ev_counts = np.array([keep[int(rs.event_off[r]):int(rs.event_off[r + 1])].sum() for r in range(rs.n_reads)])
    rs2 = AbeaReadSet(rs.seq_off, rs.seq_len, arena, np.concatenate([[0], np.cumsum(ev_counts)]), rs.event_mean[keep], rs.scale, rs.shift, rs.model)
    assert_same(rs2, align_host(rs2), O.abea_oracle(rs2, 8))


def test_shard_equivalence():
    rs = gen_abea(30, 555)
    whole = align_host(rs)
    parts = [align_host(rs.take(0, 11)), align_host(rs.take(11, 30))]
    n = np.concatenate([p[1] for p in parts])
    assert np.array_equal(n, whole[1])
    got = [x for p, sub in zip(parts, (rs.take(0, 11), rs.take(11, 30))) for x in sub.split_pairs(*p)]
    for g, w in zip(got, rs.split_pairs(*whole)):
        assert np.array_equal(g, w)


def test_reads_outside_the_fast_division_range():
    """A read whose events / model leave [2^-40, 2^40] (or hold a zero stdv-free extreme) takes the kernel's plain
    IEEE-division path; reads next to it keep the hoisted-reciprocal path.  Both must equal the oracle bit for bit."""
    rs = gen_abea(12, 77)
    ev = rs.event_mean.copy()
    ev[int(rs.event_off[3]) + 10] = np.float32(1e-30)       # tiny but non-zero: v_div_scale territory for a / stdv
    ev[int(rs.event_off[5]) + 200] = np.float32(3e13)       # beyond 2^40
    ev[int(rs.event_off[8]) + 7] = np.float32(0.0)          # zero is inside the fast range
    scale = np.asarray(rs.scale, dtype=np.float32).copy()
    scale[10] = np.float32(1e-14)                           # scaled means of read 10 become ~1e-12 < 2^-40
    rs2 = AbeaReadSet(rs.seq_off, rs.seq_len, rs.seq_arena, rs.event_off, ev, scale, rs.shift, rs.model)
    assert_same(rs2, align_host(rs2), O.abea_oracle(rs2, 8))


def test_staged_host_entry_scatters_every_read(monkeypatch):
    """The staged branch of gbx_abea_align_host (stage_field4 gather in the upload workers, abea_pack_kernel,
    HostPipe::fetch_scatter): forced on a small job, with reads that produce no pairs (failed QC) among the others and
    enough pairs to span several download pieces, every read compared with the oracle."""
    monkeypatch.setenv("GBX_HOST_STAGE_MIN", "0")
    monkeypatch.setenv("GBX_HOST_DOWN_PIECE", str(1 << 16))       # small pieces: the scatter cursor carries across them
    rs = gen_abea(96, 77, first=300)
    rng = np.random.default_rng(5)
    for r in (3, 4, 50, 95):                                      # garbage signal: QC rejects the read, n_pairs == 0
        a, b = int(rs.event_off[r]), int(rs.event_off[r + 1])
        rs.event_mean[a:b] = rng.uniform(20, 200, b - a).astype(np.float32)
    want = O.abea_oracle(rs, 8)
    assert (want[1][[3, 4, 50, 95]] == 0).all() and (want[1] > 0).sum() > 80
    assert_same(rs, align_host(rs), want)
