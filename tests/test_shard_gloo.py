"""N>1 plumbing on CPU: contiguous cost-balanced shards of every kernel's units + scatter / gather over
torch.distributed (gloo, world 2).

The ranks stand in for GPUs; each "computes" its shard with the oracle (test infrastructure) so that the
gathered result can be compared with the unsharded one: shard-equivalence by construction.  The same
shard builders, pack/unpack and scatter/gather functions carry bench.py's multi-GPU mode over RCCL.
Units: bsw pairs (R/benchmarks/bsw/main_banded.cpp:279-291), chain calls (chain/src/host_kernel.cpp:98-107),
phmm whole batches (phmm/PairHMMUnitTest.cpp:224-247), poa windows (poa/msa_spoa_omp.cpp:230-260)."""
import os
import sys

import numpy as np
import pytest

from genomicsbench_amd import shard as S
from genomicsbench_amd.shard import split_by_cost

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_split_by_cost_balances_and_covers():
    rng = np.random.default_rng(0)
    costs = rng.integers(1, 1000, 10_000)
    for parts in (1, 2, 4, 8):
        r = split_by_cost(costs, parts)
        assert r[0][0] == 0 and r[-1][1] == len(costs) and all(a[1] == b[0] for a, b in zip(r, r[1:]))
        sums = [costs[lo:hi].sum() for lo, hi in r]
        assert max(sums) - min(sums) <= 2 * costs.max()
    assert split_by_cost([], 4) == [(0, 0)] * 4
    assert split_by_cost([5], 3)[0] == (0, 0) or split_by_cost([5], 3)[-1] == (1, 1)


def test_pack_unpack_roundtrip():
    import torch
    d = dict(a=np.arange(7, dtype=np.uint8), b=np.arange(5, dtype=np.int64) - 2, c=np.zeros(0, dtype=np.int32),
             d=np.array([2 ** 63 + 5, 1], dtype=np.uint64), e=np.arange(12, dtype=np.int32).reshape(2, 6),
             f=np.array([1.5, -2.25]))
    buf, meta = S.pack_arrays(d)
    assert buf.size % 256 == 0 and all(m[3] % 256 == 0 for m in meta)
    got = S.unpack_tensor(torch.from_numpy(buf), meta)
    for k, v in d.items():
        g = got[k].numpy()
        assert g.shape == v.shape
        assert np.array_equal(g.view(v.dtype) if v.dtype == np.uint64 else g, v)


# ---- the shard builders alone: self-contained shards whose concatenated oracle results equal the unsharded run
def _bsw_full(n=3000):
    from genomicsbench_amd.datagen import gen_bsw
    return gen_bsw(n, 21)


def test_bsw_shards_are_slices_and_equivalent():
    from genomicsbench_amd.bsw import make_params
    from oracle import oracle_py as O
    full = _bsw_full()
    want = O.bsw_oracle(make_params(), full)
    for parts in (1, 3, 8):
        sh = S.bsw_shards(full, parts)
        assert sum(s.n for s in sh) == full.n
        for s in sh:                                # arenas are cut down to the shard, offsets start at 0
            assert s.n == 0 or (s.idr[0] == 0 and s.idq[0] == 0 and s.ref.size <= full.ref.size // parts * 2 + 4096)
        got = np.concatenate([O.bsw_oracle(make_params(), s) for s in sh])
        assert np.array_equal(got, want)


def test_bsw_shards_repack_scattered_arena():
    """A non-monotone arena (pairs stored in reverse, with gaps) takes the gather path: same results."""
    from genomicsbench_amd.bsw import BswBatch, make_params
    from oracle import oracle_py as O
    full = _bsw_full(500)
    # rebuild the arenas back to front with 3 bytes of junk between sequences
    def reverse_layout(arena, off, lens):
        pos, new_off = 0, np.zeros(len(off), dtype=np.int64)
        out = np.full(arena.size + 8 * len(off) + 64, 9, dtype=np.uint8)
        for k in range(len(off) - 1, -1, -1):
            new_off[k] = pos
            out[pos:pos + lens[k]] = arena[off[k]:off[k] + lens[k]]
            pos += int(lens[k]) + 3
        return out, new_off
    ref, idr = reverse_layout(full.ref, full.idr, full.len1)
    qer, idq = reverse_layout(full.qer, full.idq, full.len2)
    scattered = BswBatch(ref, qer, idr, idq, full.len1, full.len2, full.h0)
    want = O.bsw_oracle(make_params(), full)
    sh = S.bsw_shards(scattered, 3)
    assert all(bool(np.all(np.diff(s.idr) >= 0)) for s in sh)           # re-packed front to back
    assert np.array_equal(np.concatenate([O.bsw_oracle(make_params(), s) for s in sh]), want)


def test_chain_shards_equivalent():
    from genomicsbench_amd.datagen import gen_chain
    from oracle import oracle_py as O
    case = gen_chain(40, 2001)
    want = O.chain_oracle(*case)
    for parts in (2, 5):
        sh = S.chain_shards(*case, parts)
        assert sum(len(s[0]) - 1 for s in sh) == 40 and all(s[0][0] == 0 for s in sh)
        anchors = [int(s[0][-1]) for s in sh]
        assert max(anchors) - min(anchors) <= 2 * int(np.diff(case[0]).max())
        got = [O.chain_oracle(*s) for s in sh]
        for f in range(4):
            assert np.array_equal(np.concatenate([g[f] for g in got]), want[f])


def test_phmm_shards_keep_whole_batches():
    from genomicsbench_amd.datagen import gen_phmm
    from oracle import oracle_py as O
    bs = gen_phmm(30, 3001)
    want = O.phmm_oracle(bs, 2)
    for parts in (2, 4):
        sh = S.phmm_shards(bs, parts)
        assert sum(len(s.n_reads) for s in sh) == 30 and sum(s.n_pairs for s in sh) == bs.n_pairs
        assert sum(s.cells for s in sh) == bs.cells
        for s in sh:
            assert len(s.read_off) == 0 or (s.read_off[0] == 0 and s.hap_off[0] == 0)
        got = np.concatenate([O.phmm_oracle(s, 2) for s in sh])
        assert np.array_equal(got, want)            # same pairs, same order, same arithmetic: bit-identical


def test_poa_shards_equivalent_and_cost_weighted():
    from genomicsbench_amd.datagen import gen_poa
    from genomicsbench_amd.poa import make_params
    from oracle import oracle_py as O
    ws = gen_poa(12, 4001)
    want = O.poa_oracle(make_params(), ws, 2)
    sh = S.poa_shards(ws, 3)
    assert sum(s.n_windows for s in sh) == 12 and all(s.seq_off[0] == 0 for s in sh if s.n_seqs)
    got = sum((O.poa_oracle(make_params(), s, 2) for s in sh), [])
    assert got == want
    # a deep window costs more than two shallow ones of the same total length
    cost = S.poa_cost(ws)
    n = np.diff(ws.win_first_seq)
    assert cost[np.argmax(n)] > cost[np.argmin(n)]


def test_empty_shards():
    """More ranks than units: trailing shards are empty but well-formed."""
    from genomicsbench_amd.datagen import gen_chain, gen_phmm, gen_poa
    assert [s.n for s in S.bsw_shards(_bsw_full(2), 4)].count(0) >= 2
    assert sum(len(s[0]) - 1 for s in S.chain_shards(*gen_chain(1, 2001), 3)) == 1
    assert sum(len(s.n_reads) for s in S.phmm_shards(gen_phmm(1, 3001), 3)) == 1
    assert sum(s.n_windows for s in S.poa_shards(gen_poa(1, 4001), 3)) == 1


# ---- world-2 scatter -> compute -> gather for each kernel ---------------------------------------------------
def _worker(rank, world, port, tmp, kernel):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from genomicsbench_amd import shard as S
    from oracle import oracle_py as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    np_of = lambda d: {k: v.numpy() for k, v in d.items()}
    per_rank = full = None
    if kernel == "bsw":
        from genomicsbench_amd.bsw import make_params
        from genomicsbench_amd.datagen import gen_bsw
        if rank == 0:
            full = gen_bsw(3000, 21)
            per_rank = [S.bsw_to_arrays(s) for s in S.bsw_shards(full, world)]
        mine, _ = S.scatter_arrays(per_rank)
        b = S.bsw_from_arrays(np_of(mine))
        parts = S.gather_array(torch.from_numpy(O.bsw_oracle(make_params(), b)))
        units = b.n
        if rank == 0:
            ok = np.array_equal(torch.cat(parts).numpy(), O.bsw_oracle(make_params(), full))
    elif kernel == "chain":
        from genomicsbench_amd.datagen import gen_chain
        if rank == 0:
            full = gen_chain(24, 2001)
            per_rank = [S.chain_to_arrays(s) for s in S.chain_shards(*full, world)]
        mine, _ = S.scatter_arrays(per_rank)
        case = S.chain_from_arrays(np_of(mine))
        got = O.chain_oracle(*case)
        parts = S.gather_array(torch.from_numpy(np.stack(got, axis=1)))      # [anchors, 4] fixed-stride records
        units = len(case[0]) - 1
        if rank == 0:
            ok = np.array_equal(torch.cat(parts).numpy(), np.stack(O.chain_oracle(*full), axis=1))
    elif kernel == "phmm":
        from genomicsbench_amd.datagen import gen_phmm
        if rank == 0:
            full = gen_phmm(24, 3001)
            per_rank = [S.phmm_to_arrays(s) for s in S.phmm_shards(full, world)]
        mine, _ = S.scatter_arrays(per_rank)
        bs = S.phmm_from_arrays(np_of(mine))
        parts = S.gather_array(torch.from_numpy(O.phmm_oracle(bs, 2)))
        units = len(bs.n_reads)
        if rank == 0:
            ok = np.array_equal(torch.cat(parts).numpy(), O.phmm_oracle(full, 2))
    elif kernel == "fmi":
        from genomicsbench_amd.datagen import gen_fmi_genome, gen_fmi_reads
        from genomicsbench_amd.fmi import SMEM_DTYPE, build_index
        idx = build_index(gen_fmi_genome(30000, 6001))        # every rank holds the whole index (built from the same seed)
        if rank == 0:
            full = gen_fmi_reads(gen_fmi_genome(30000, 6001), 300, 6002)
            per_rank = [S.fmi_to_arrays(s) for s in S.fmi_shards(full, world)]
        mine, _ = S.scatter_arrays(per_rank)
        rs = S.fmi_from_arrays(np_of(mine))
        out, off = O.fmi_oracle(idx, rs)
        flat = np.concatenate([np.array([rs.n_reads], dtype=np.int64).view(np.uint8), off.view(np.uint8), out.view(np.uint8)])
        parts = S.gather_array(torch.from_numpy(flat))      # variable-length records: [reads | offsets | records] as bytes
        units = rs.n_reads
        if rank == 0:
            wo, woff = O.fmi_oracle(idx, full)
            recs, first, ok_off = [], 0, True
            for p in parts:
                p = p.numpy()
                nr = int(p[:8].view(np.int64)[0])
                o = p[8:8 + 8 * (nr + 1)].view(np.int64)
                r = p[8 + 8 * (nr + 1):].view(SMEM_DTYPE).copy()
                r["rid"] += first                            # rid is shard-local, as it is batch-local in fmi.cpp:270-273
                ok_off = ok_off and np.array_equal(o + woff[first], woff[first:first + nr + 1])
                recs.append(r)
                first += nr
            got = np.concatenate(recs)
            ok = ok_off and first == full.n_reads and all(np.array_equal(got[f], wo[f]) for f in ("rid", "m", "n", "k", "l", "s"))
    elif kernel == "abea":
        from genomicsbench_amd.datagen import gen_abea
        if rank == 0:
            full = gen_abea(8, 5001)
            per_rank = [S.abea_to_arrays(s) for s in S.abea_shards(full, world)]
        mine, _ = S.scatter_arrays(per_rank)
        rs = S.abea_from_arrays(np_of(mine))
        out, n = O.abea_oracle(rs, 2)
        flat = np.concatenate([out["ref_pos"], out["read_pos"], n]).astype(np.int32)      # pairs then counts, one message
        parts = S.gather_array(torch.from_numpy(flat))
        units = rs.n_reads
        if rank == 0:
            wo, wn = O.abea_oracle(full, 2)
            want_pairs = full.split_pairs(wo, wn)
            got_pairs, k = [], 0
            for p, sh in zip(parts, S.abea_shards(full, world)):
                p = p.numpy()
                ne = 2 * max(int(sh.event_off[-1]), 1)
                o = np.zeros(ne, dtype=wo.dtype); o["ref_pos"] = p[:ne]; o["read_pos"] = p[ne:2 * ne]
                got_pairs += sh.split_pairs(o, p[2 * ne:])
            ok = len(got_pairs) == len(want_pairs) and all(np.array_equal(g, w) for g, w in zip(got_pairs, want_pairs))
    else:
        from genomicsbench_amd.datagen import gen_poa
        from genomicsbench_amd.poa import make_params
        if rank == 0:
            full = gen_poa(10, 4001)
            per_rank = [S.poa_to_arrays(s) for s in S.poa_shards(full, world)]
        mine, _ = S.scatter_arrays(per_rank)
        ws = S.poa_from_arrays(np_of(mine))
        cons = O.poa_oracle(make_params(), ws, 2)
        stride = max([len(c) for c in cons] + [1])
        rec = np.zeros((len(cons), stride + 4), dtype=np.uint8)                # length-prefixed fixed-stride rows
        for k, c in enumerate(cons):
            rec[k, :4] = np.frombuffer(np.int32(len(c)).tobytes(), dtype=np.uint8)
            rec[k, 4:4 + len(c)] = np.frombuffer(c.encode(), dtype=np.uint8)
        parts = S.gather_array(torch.from_numpy(rec))
        units = ws.n_windows
        if rank == 0:
            got = []
            for p in parts:
                for row in p.numpy():
                    ln = int(np.frombuffer(row[:4].tobytes(), dtype=np.int32)[0])
                    got.append(row[4:4 + ln].tobytes().decode())
            ok = got == O.poa_oracle(make_params(), full, 2)
    if rank == 0:
        np.save(os.path.join(tmp, "ok.npy"), np.array([int(ok), len(parts), units]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("kernel,total", [("bsw", 3000), ("chain", 24), ("phmm", 24), ("poa", 10), ("abea", 8), ("fmi", 300)])
def test_scatter_compute_gather_world2(tmp_path, kernel, total):
    import torch.multiprocessing as mp
    port = 29500 + (os.getpid() % 2000) + {"bsw": 0, "chain": 1, "phmm": 2, "poa": 3, "abea": 4, "fmi": 5}[kernel]
    mp.spawn(_worker, args=(2, port, str(tmp_path), kernel), nprocs=2, join=True)
    ok = np.load(str(tmp_path / "ok.npy"))
    assert ok[0] == 1 and ok[1] == 2 and 0 < ok[2] < total


@pytest.mark.parametrize("kernel,total", [("bsw", 3000), ("poa", 10)])
def test_scatter_gather_in_capped_pieces_world2(tmp_path, kernel, total, monkeypatch):
    """A packed shard larger than the per-message cap (default 1 GiB: fmi's and poa's 'large' shards run to 1.5 GB) travels
    as several pieces of one grouped send / receive, cut alike on both sides; here the cap is 4 KiB, so every shard and
    every gathered result is tens of pieces."""
    import torch.multiprocessing as mp
    monkeypatch.setenv("GBX_SHARD_MSG_BYTES", "4096")
    port = 29500 + (os.getpid() % 2000) + {"bsw": 10, "poa": 11}[kernel]
    mp.spawn(_worker, args=(2, port, str(tmp_path), kernel), nprocs=2, join=True)
    ok = np.load(str(tmp_path / "ok.npy"))
    assert ok[0] == 1 and ok[1] == 2 and 0 < ok[2] < total


def test_message_pieces_cover_the_buffer(monkeypatch):
    import torch
    monkeypatch.setenv("GBX_SHARD_MSG_BYTES", "1000")
    for n in (0, 1, 999, 1000, 1001, 5000):
        buf = torch.arange(n, dtype=torch.int64).to(torch.uint8)
        ps = S._pieces(buf)
        assert sum(int(p.numel()) for p in ps) == n and all(int(p.numel()) <= 1000 for p in ps)
        if n:
            assert torch.equal(torch.cat(ps), buf) and ps[0].data_ptr() == buf.data_ptr()      # views, not copies
    monkeypatch.delenv("GBX_SHARD_MSG_BYTES")
    assert S._msg_cap() == 1 << 30
