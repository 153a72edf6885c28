"""N>1 plumbing on CPU: contiguous cost-balanced shards + scatter/gather over torch.distributed (gloo, world 2).

The ranks stand in for GPUs; each "computes" its shard with the oracle (test infrastructure) so that the
gathered result can be compared with the unsharded one: shard-equivalence by construction."""
import os
import sys

import numpy as np
import pytest

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


def _worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from genomicsbench_amd.bsw import BswBatch, make_params
    from genomicsbench_amd.datagen import gen_bsw
    from genomicsbench_amd.shard import bsw_shards, gather_array, scatter_arrays
    from oracle import oracle_py as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    per_rank = None
    if rank == 0:
        full = gen_bsw(3000, 21)
        per_rank = [dict(ref=s.ref, qer=s.qer, idr=s.idr, idq=s.idq, len1=s.len1, len2=s.len2, h0=s.h0)
                    for s in bsw_shards(full, world)]
    mine, _ = scatter_arrays(per_rank)
    b = BswBatch(*(mine[k].numpy() for k in ("ref", "qer", "idr", "idq", "len1", "len2", "h0")))
    out = torch.from_numpy(O.bsw_oracle(make_params(), b))
    parts = gather_array(out)
    if rank == 0:
        got = torch.cat(parts).numpy()
        want = O.bsw_oracle(make_params(), full)
        np.save(os.path.join(tmp, "ok.npy"), np.array([int(np.array_equal(got, want)), len(parts), b.n]))
    dist.barrier()
    dist.destroy_process_group()


def test_scatter_compute_gather_world2(tmp_path):
    import torch.multiprocessing as mp
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    ok = np.load(str(tmp_path / "ok.npy"))
    assert ok[0] == 1 and ok[1] == 2 and 0 < ok[2] < 3000
