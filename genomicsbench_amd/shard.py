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


_ALIGN = 256


def pack_arrays(d):
    """dict name -> numpy array  =>  (one uint8 buffer, [(name, dtype, shape, byte offset)]); 256-byte aligned pieces."""
    meta, pos = [], 0
    for k, v in d.items():
        v = np.ascontiguousarray(v)
        meta.append((k, str(v.dtype), tuple(v.shape), pos))
        pos += (v.nbytes + _ALIGN - 1) // _ALIGN * _ALIGN
    buf = np.zeros(pos + _ALIGN, dtype=np.uint8)           # trailing slack: kernels may read 16 B past an arena
    for (k, _, _, off), v in zip(meta, d.values()):
        v = np.ascontiguousarray(v)
        buf[off:off + v.nbytes] = v.reshape(-1).view(np.uint8)
    return buf, meta


def unpack_tensor(buf, meta):
    """Views of the pieces of a packed buffer (a uint8 torch tensor on any device) as typed tensors."""
    out = {}
    for k, dt, shape, off in meta:
        tdt = _torch_dtype(dt)
        nbytes = int(np.prod(shape, dtype=np.int64)) * np.dtype(dt).itemsize
        out[k] = buf[off:off + nbytes].view(tdt).reshape(shape)
    return out


def _msg_cap():
    """Largest single send / receive, in bytes (default 1 GiB; GBX_SHARD_MSG_BYTES for the tests): a packed shard of fmi's
    reads or poa's windows runs to 1.5 GB, and one message of that size is the first thing a transport's internal 32-bit
    counts or staging buffers would trip over - larger buffers travel as several pieces of one grouped launch."""
    import os
    return max(256, int(os.environ.get("GBX_SHARD_MSG_BYTES", str(1 << 30))))


def _pieces(buf):
    """A contiguous 1-D uint8 tensor as views of at most _msg_cap() bytes (sender and receiver cut alike)."""
    cap, n = _msg_cap(), int(buf.numel())
    return [buf[a:min(n, a + cap)] for a in range(0, n, cap)] or [buf]


def scatter_arrays(per_rank, device="cpu", root=0):
    """root passes a list (one dict name -> numpy array per rank); every rank returns its own dict of torch tensors.

    One message per rank: the arrays of a shard are packed into one byte buffer (offset table broadcast beforehand as
    a small object), all sends posted as one group (RCCL: one grouped ncclSend/ncclRecv launch)."""
    import torch
    dist = _dist()
    rank, world = dist.get_rank(), dist.get_world_size()
    packed = [pack_arrays(d) for d in per_rank] if rank == root else None
    box = [[(m, int(b.size)) for b, m in packed] if rank == root else None]
    dist.broadcast_object_list(box, src=root)
    meta, nbytes = box[0][rank]
    if rank == root:
        bufs = [torch.from_numpy(b).to(device) for b, _ in packed]
        ops = [dist.P2POp(dist.isend, piece, r) for r in range(world) if r != root for piece in _pieces(bufs[r])]
        mine = bufs[root]
    else:
        mine = torch.empty(nbytes, dtype=torch.uint8, device=device)
        ops = [dist.P2POp(dist.irecv, piece, root) for piece in _pieces(mine)]
    if ops:
        for q in dist.batch_isend_irecv(ops):
            q.wait()
    return unpack_tensor(mine, meta), meta


def gather_array(local, root=0):
    """Every rank passes a torch tensor (first dimension may differ); root gets the list of all of them."""
    import torch
    dist = _dist()
    rank, world = dist.get_rank(), dist.get_world_size()
    shapes = [None] * world
    dist.all_gather_object(shapes, (tuple(local.shape), str(local.dtype)))
    local = local.contiguous()
    as_bytes = lambda t: t.reshape(-1).view(torch.uint8) if t.numel() else t.reshape(-1)
    if rank != root:
        if local.numel():
            for q in dist.batch_isend_irecv([dist.P2POp(dist.isend, piece, root) for piece in _pieces(as_bytes(local))]):
                q.wait()
        return None
    out, ops = [], []
    for r in range(world):
        if r == root:
            out.append(local)
        else:
            shape, dt = shapes[r]
            t = torch.empty(shape, dtype=getattr(torch, dt.split(".")[-1]), device=local.device)
            if t.numel():
                ops += [dist.P2POp(dist.irecv, piece, r) for piece in _pieces(as_bytes(t))]
            out.append(t)
    if ops:
        for q in dist.batch_isend_irecv(ops):
            q.wait()
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
# A shard is a contiguous range of units, cut where the prefix sum of a per-unit cost crosses k/parts
# of the total.  Every builder returns `parts` objects of the kernel's own batch type with arenas cut down to the
# shard and offsets re-based, so that a shard is a self-contained input of the C-ABI; `*_to_arrays` / `*_from_arrays`
# turn one into the flat dict scatter_arrays() ships and back.  Reference units: bsw pairs
# (R/benchmarks/bsw/main_banded.cpp:279-291), chain calls (chain/src/host_kernel.cpp:98-107), phmm batches
# (phmm/PairHMMUnitTest.cpp:224-247), poa windows (poa/msa_spoa_omp.cpp:230-260).

def _cut_arena(arena, off, lens, align=1):
    """Bytes [off[k], off[k]+lens[k]) of a run of units as (sub-arena, re-based offsets).

    A monotone, non-overlapping layout (what every loader and generator here produces) is one slice; anything else
    is re-packed with a vectorised gather in bounded pieces."""
    n = len(off)
    if n == 0:
        return np.zeros(8, dtype=arena.dtype), np.zeros(0, dtype=np.int64)
    off = np.asarray(off, dtype=np.int64)
    lens64 = np.asarray(lens, dtype=np.int64)
    if n == 1 or bool(np.all(off[1:] >= off[:-1] + lens64[:-1])):
        lo, hi = int(off[0]), int(off[-1] + lens64[-1])
        sub = np.empty(hi - lo + 8, dtype=arena.dtype)
        sub[:hi - lo] = arena[lo:hi]
        sub[hi - lo:] = 0
        return sub, off - lo
    al = (lens64 + align - 1) // align * align
    new_off = np.concatenate([[0], np.cumsum(al)[:-1]]).astype(np.int64)
    sub = np.zeros(int(al.sum()) + 8, dtype=arena.dtype)
    step = 1 << 16
    for a in range(0, n, step):
        b = min(n, a + step)
        ln = lens64[a:b]
        tot = int(ln.sum())
        if tot == 0:
            continue
        within = np.arange(tot, dtype=np.int64) - np.repeat(np.cumsum(ln) - ln, ln)
        sub[np.repeat(new_off[a:b], ln) + within] = arena[np.repeat(off[a:b], ln) + within]
    return sub, new_off


def bsw_cost(batch):
    return batch.len1.astype(np.int64) * batch.len2


def bsw_shards(batch, parts, ranges=None):
    """`parts` BswBatch objects over contiguous pair ranges balanced by nominal cells (len1*len2)."""
    from .bsw import BswBatch
    out = []
    for lo, hi in (ranges or split_by_cost(bsw_cost(batch), parts)):
        ref, idr = _cut_arena(batch.ref, batch.idr[lo:hi], batch.len1[lo:hi], 4)
        qer, idq = _cut_arena(batch.qer, batch.idq[lo:hi], batch.len2[lo:hi], 4)
        out.append(BswBatch(ref, qer, idr, idq, batch.len1[lo:hi], batch.len2[lo:hi], batch.h0[lo:hi]))
    return out


def bsw_to_arrays(b):
    return dict(ref=b.ref, qer=b.qer, idr=b.idr, idq=b.idq, len1=b.len1, len2=b.len2, h0=b.h0)


def bsw_from_arrays(d):
    from .bsw import BswBatch
    return BswBatch(*(np.asarray(d[k]) for k in ("ref", "qer", "idr", "idq", "len1", "len2", "h0")))


def chain_cost(off):
    """Anchors per call: every anchor looks back over a bounded window, so a call's work is linear in its length."""
    return np.diff(np.asarray(off, dtype=np.int64))


def chain_shards(off, ax, ay, hdr, parts, ranges=None):
    """`parts` tuples (anchor_off, ax, ay, hdr) over contiguous call ranges balanced by anchors."""
    off = np.asarray(off, dtype=np.int64)
    out = []
    for lo, hi in (ranges or split_by_cost(chain_cost(off), parts)):
        a, b = int(off[lo]), int(off[hi])
        out.append((off[lo:hi + 1] - a, np.ascontiguousarray(ax[a:b]), np.ascontiguousarray(ay[a:b]),
                    np.ascontiguousarray(hdr[lo:hi])))
    return out


def chain_to_arrays(case):
    off, ax, ay, hdr = case
    return dict(off=np.ascontiguousarray(off, dtype=np.int64), ax=np.ascontiguousarray(ax, dtype=np.uint64),
                ay=np.ascontiguousarray(ay, dtype=np.uint64), hdr=np.ascontiguousarray(hdr).view(np.uint8))


def chain_from_arrays(d):
    from ._native import CHAIN_CALL_DTYPE
    return (np.asarray(d["off"]), np.asarray(d["ax"]).view(np.uint64), np.asarray(d["ay"]).view(np.uint64),
            np.ascontiguousarray(d["hdr"]).view(np.uint8).view(CHAIN_CALL_DTYPE))


def phmm_cost(bs):
    """Cells of a batch = (sum of its read lengths) x (sum of its haplotype lengths): every read meets every haplotype
    of its batch (PairHMMUnitTest.cpp:137,232-244)."""
    rb = np.concatenate([[0], np.cumsum(bs.n_reads.astype(np.int64))])
    hb = np.concatenate([[0], np.cumsum(bs.n_haps.astype(np.int64))])
    rl = np.concatenate([[0], np.cumsum(bs.read_len.astype(np.int64))])
    hl = np.concatenate([[0], np.cumsum(bs.hap_len.astype(np.int64))])
    return (rl[rb[1:]] - rl[rb[:-1]]) * (hl[hb[1:]] - hl[hb[:-1]])


def phmm_shards(bs, parts, ranges=None):
    """`parts` PhmmBatchSet objects over contiguous runs of WHOLE batches (a batch's reads and haplotypes stay
    together, so its arenas are shared by its pairs exactly as in the unsharded job), balanced by cells."""
    from .phmm import PhmmBatchSet
    rb = np.concatenate([[0], np.cumsum(bs.n_reads.astype(np.int64))])
    hb = np.concatenate([[0], np.cumsum(bs.n_haps.astype(np.int64))])
    out = []
    for lo, hi in (ranges or split_by_cost(phmm_cost(bs), parts)):
        r0, r1, h0, h1 = int(rb[lo]), int(rb[hi]), int(hb[lo]), int(hb[hi])
        roff, rlen = bs.read_off[r0:r1], bs.read_len[r0:r1]
        tracks = []
        new_roff = None
        for tr in (bs.rs, bs.q, bs.qi, bs.qd, bs.qc):
            sub, new_roff = _cut_arena(tr, roff, rlen)
            tracks.append(sub)
        hap, new_hoff = _cut_arena(bs.hap, bs.hap_off[h0:h1], bs.hap_len[h0:h1])
        out.append(PhmmBatchSet(bs.n_reads[lo:hi], bs.n_haps[lo:hi], new_roff, rlen, *tracks,
                                new_hoff, bs.hap_len[h0:h1], hap))
    return out


_PHMM_KEYS = ("n_reads", "n_haps", "read_off", "read_len", "rs", "q", "qi", "qd", "qc", "hap_off", "hap_len", "hap")


def phmm_to_arrays(bs):
    return {k: getattr(bs, k) for k in _PHMM_KEYS}


def phmm_from_arrays(d):
    from .phmm import PhmmBatchSet
    return PhmmBatchSet(*(np.asarray(d[k]) for k in _PHMM_KEYS))


def poa_cost(ws):
    """DP cells of a window ~ sum over its sequences of (graph nodes when it is aligned) x (its length); the graph
    starts as the first sequence and grows by about a tenth of every later one, so with n sequences of mean length
    L the sum is ~ n L^2 (1 + n/20): quadratic in the window's depth (poa/msa_spoa_omp.cpp:237-252)."""
    wf = ws.win_first_seq
    cl = np.concatenate([[0], np.cumsum(ws.seq_len.astype(np.int64))])
    tot = (cl[wf[1:]] - cl[wf[:-1]]).astype(np.float64)
    n = np.maximum(np.diff(wf), 1).astype(np.float64)
    return tot * (tot / n) * (1.0 + n / 20.0)


def poa_shards(ws, parts, ranges=None):
    """`parts` PoaWindowSet objects over contiguous window ranges balanced by poa_cost."""
    from .poa import PoaWindowSet
    out = []
    for lo, hi in (ranges or split_by_cost(poa_cost(ws), parts)):
        a, b = int(ws.win_first_seq[lo]), int(ws.win_first_seq[hi])
        arena, off = _cut_arena(ws.arena, ws.seq_off[a:b], ws.seq_len[a:b])
        out.append(PoaWindowSet(ws.win_first_seq[lo:hi + 1] - a, off, ws.seq_len[a:b], arena))
    return out


def poa_to_arrays(ws):
    return dict(win_first_seq=ws.win_first_seq, seq_off=ws.seq_off, seq_len=ws.seq_len, arena=ws.arena)


def poa_from_arrays(d):
    from .poa import PoaWindowSet
    return PoaWindowSet(*(np.asarray(d[k]) for k in ("win_first_seq", "seq_off", "seq_len", "arena")))


def abea_cost(rs):
    """Bands of a read = events + k-mers + 2; every band is 100 cells (R/benchmarks/abea/src/align.c:209-211)."""
    return rs.n_bands


def abea_shards(rs, parts, ranges=None):
    """`parts` AbeaReadSet objects over contiguous read ranges balanced by bands (reads are independent, f5c.c:1350-1370)."""
    from .abea import AbeaReadSet
    out = []
    for lo, hi in (ranges or split_by_cost(abea_cost(rs), parts)):
        a, b = int(rs.event_off[lo]), int(rs.event_off[hi])
        arena, off = _cut_arena(rs.seq_arena, rs.seq_off[lo:hi], rs.seq_len[lo:hi])
        out.append(AbeaReadSet(off, rs.seq_len[lo:hi], arena, rs.event_off[lo:hi + 1] - a, rs.event_mean[a:b],
                               rs.scale[lo:hi], rs.shift[lo:hi], rs.model))
    return out


def abea_to_arrays(rs):
    return dict(seq_off=rs.seq_off, seq_len=rs.seq_len, seq_arena=rs.seq_arena, event_off=rs.event_off,
                event_mean=np.concatenate([rs.event_mean, np.zeros(4, np.float32)]), scale=rs.scale, shift=rs.shift,
                model=np.ascontiguousarray(rs.model).view(np.uint8))


def abea_from_arrays(d):
    from .abea import MODEL_DTYPE, AbeaReadSet
    eo = np.asarray(d["event_off"])
    return AbeaReadSet(np.asarray(d["seq_off"]), np.asarray(d["seq_len"]), np.asarray(d["seq_arena"]), eo,
                       np.asarray(d["event_mean"])[:int(eo[-1] - eo[0])] if len(eo) else np.zeros(0, np.float32),
                       np.asarray(d["scale"]), np.asarray(d["shift"]),
                       np.ascontiguousarray(d["model"]).view(np.uint8).view(MODEL_DTYPE))


def fmi_cost(rs):
    """A read costs about five backwardExt calls per base, whatever its content (fmi.cpp:193-282 hands the reads out in
    batches of equal count)."""
    return rs.read_len.astype(np.int64) + 1


def fmi_shards(rs, parts, ranges=None):
    """`parts` FmiReadSet objects over contiguous read ranges (reads are independent; a batch of the reference is a
    contiguous rid range, fmi.cpp:193-197).  The index is not part of a shard: every rank holds all of it."""
    from .fmi import FmiReadSet
    out = []
    for lo, hi in (ranges or split_by_cost(fmi_cost(rs), parts)):
        enc, off = _cut_arena(rs.enc, rs.read_off[lo:hi], rs.read_len[lo:hi])
        out.append(FmiReadSet(enc, off, rs.read_len[lo:hi]))
    return out


def fmi_to_arrays(rs):
    return dict(enc=np.concatenate([rs.enc, np.zeros(8, np.uint8)]), read_off=rs.read_off, read_len=rs.read_len)


def fmi_from_arrays(d):
    from .fmi import FmiReadSet
    return FmiReadSet(np.asarray(d["enc"]), np.asarray(d["read_off"]), np.asarray(d["read_len"]))
