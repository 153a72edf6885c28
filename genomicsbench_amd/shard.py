"""Sharding of independent work units (pairs / calls / batches / windows) across the GPUs of one node.

All four kernels are embarrassingly parallel over their units: no data-path collective.
Units are split into contiguous ranges balanced by cost (nominal cells); the only communication is the
one-shot scatter of a rank's input arrays from rank 0 and the gather of its outputs back, over
torch.distributed point-to-point ops (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).
"""
import numpy as np


def split_by_cost(costs, parts):
    """Contiguous ranges [(lo, hi)] * parts with near-equal total cost (prefix-sum split)."""
    costs = np.asarray(costs, dtype=np.float64)
    n = len(costs)
    if parts <= 1 or n == 0:
        return [(0, n)] + [(n, n)] * (parts - 1)
    cum = np.concatenate([[0.0], np.cumsum(costs)])
    total = cum[-1]
    cuts = [0]
    for k in range(1, parts):
        cuts.append(int(np.searchsorted(cum, total * k / parts, side="left")))
    cuts.append(n)
    cuts = np.maximum.accumulate(np.minimum(cuts, n))
    return [(int(cuts[k]), int(cuts[k + 1])) for k in range(parts)]


def _dist():
    import torch.distributed as dist
    return dist


def scatter_arrays(per_rank, device="cpu", root=0):
    """root passes a list (one dict name -> numpy array per rank); every rank returns its own dict of torch tensors.

    Implemented as a size header broadcast followed by grouped send/recv of the raw arrays."""
    import torch
    dist = _dist()
    rank, world = dist.get_rank(), dist.get_world_size()
    if rank == root:
        meta = [{k: (str(v.dtype), tuple(v.shape)) for k, v in d.items()} for d in per_rank]
    else:
        meta = None
    box = [meta]
    dist.broadcast_object_list(box, src=root)
    meta = box[0]
    mine = {}
    if rank == root:
        reqs = []
        for r in range(world):
            for k, v in per_rank[r].items():
                t = torch.from_numpy(np.ascontiguousarray(_as_signed(v))).to(device)
                if r == root:
                    mine[k] = t
                else:
                    reqs.append(dist.isend(t, dst=r))
        for q in reqs:
            q.wait()
    else:
        for k, (dt, shape) in meta[rank].items():
            t = torch.empty(shape, dtype=_torch_dtype(dt), device=device)
            dist.recv(t, src=root)
            mine[k] = t
    return mine, meta[rank]


def gather_array(local, root=0):
    """Every rank passes a torch tensor (first dimension may differ); root gets the list of all of them."""
    import torch
    dist = _dist()
    rank, world = dist.get_rank(), dist.get_world_size()
    shapes = [None] * world
    dist.all_gather_object(shapes, (tuple(local.shape), str(local.dtype)))
    if rank != root:
        dist.send(local.contiguous(), dst=root)
        return None
    out = []
    for r in range(world):
        if r == root:
            out.append(local)
        else:
            shape, dt = shapes[r]
            t = torch.empty(shape, dtype=getattr(torch, dt.split(".")[-1]), device=local.device)
            dist.recv(t, src=r)
            out.append(t)
    return out


def _as_signed(a):
    """torch has no uint16/32/64 tensors on every backend: ship unsigned arrays as same-width signed views."""
    if a.dtype == np.uint64:
        return a.view(np.int64)
    if a.dtype == np.uint32:
        return a.view(np.int32)
    if a.dtype == np.uint16:
        return a.view(np.int16)
    return a


def _torch_dtype(name):
    import torch
    return {"uint8": torch.uint8, "int8": torch.int8, "int16": torch.int16, "uint16": torch.int16,
            "int32": torch.int32, "uint32": torch.int32, "int64": torch.int64, "uint64": torch.int64,
            "float32": torch.float32, "float64": torch.float64}[name]


# ---- per-kernel shard builders (host side, rank 0) ---------------------------------------------------
def bsw_shards(batch, parts):
    """Slices a BswBatch into `parts` BswBatch views balanced by nominal cells (arenas are re-packed per shard)."""
    from .bsw import BswBatch
    cost = batch.len1.astype(np.int64) * batch.len2
    out = []
    for lo, hi in split_by_cost(cost, parts):
        ts = [batch.ref[batch.idr[k]:batch.idr[k] + batch.len1[k]] for k in range(lo, hi)]
        qs = [batch.qer[batch.idq[k]:batch.idq[k] + batch.len2[k]] for k in range(lo, hi)]
        out.append(BswBatch.from_sequences(ts, qs, batch.h0[lo:hi]) if hi > lo else
                   BswBatch(np.zeros(4, np.uint8), np.zeros(4, np.uint8), [], [], [], [], []))
    return out
