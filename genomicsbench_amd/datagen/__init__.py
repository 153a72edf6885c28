"""Seeded synthetic datasets in the reference input shapes.  Tooling only."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(os.path.dirname(_HERE), "libgbx_datagen.so")
_lib = None


def _L():
    global _lib
    if _lib is None:
        from ..build import build_datagen
        build_datagen()                      # no-op when the library is newer than datagen.c
        _lib = C.CDLL(_LIB)
        vp, i64, u64 = C.c_void_p, C.c_int64, C.c_uint64
        _lib.gbx_gen_bsw_lengths.argtypes = [u64, i64, i64, vp, vp, vp]
        _lib.gbx_gen_bsw_lengths.restype = None
        _lib.gbx_gen_bsw_fill.argtypes = [u64, i64, i64, vp, vp, vp, vp, vp, vp]
        _lib.gbx_gen_bsw_fill.restype = None
        _lib.gbx_gen_chain_count.argtypes = [u64, i64]
        _lib.gbx_gen_chain_count.restype = i64
        _lib.gbx_gen_chain_fill.argtypes = [u64, i64, i64, vp, vp]
        _lib.gbx_gen_chain_fill.restype = None
        _lib.gbx_gen_phmm_batch.argtypes = [u64, i64, C.c_int] + [vp] * 10
        _lib.gbx_gen_phmm_batch.restype = None
        _lib.gbx_gen_poa_window.argtypes = [u64, i64, C.c_int, vp, vp, vp]
        _lib.gbx_gen_poa_window.restype = None
        _lib.gbx_gen_chain_counts_many.argtypes = [u64, i64, i64, vp]
        _lib.gbx_gen_chain_fill_many.argtypes = [u64, i64, i64, vp, vp, vp]
        _lib.gbx_gen_chain_fill_real_many.argtypes = [u64, i64, i64, vp, vp, vp]
        _lib.gbx_gen_phmm_counts_many.argtypes = [u64, i64, i64, vp, vp]
        _lib.gbx_gen_phmm_lengths_many.argtypes = [u64, i64, i64, vp, vp, vp, vp]
        _lib.gbx_gen_phmm_fill_many.argtypes = [u64, i64, i64] + [vp] * 10
        _lib.gbx_gen_poa_counts_many.argtypes = [u64, i64, i64, vp]
        _lib.gbx_gen_poa_many.argtypes = [u64, i64, i64, C.c_int, vp, vp, vp, vp]
        _lib.gbx_gen_abea_model.argtypes = [u64, vp, vp]
        _lib.gbx_gen_abea_counts_many.argtypes = [u64, i64, i64, vp, vp]
        _lib.gbx_gen_abea_fill_many.argtypes = [u64, i64, i64] + [vp] * 8
        _lib.gbx_gen_fmi_genome.argtypes = [u64, i64, vp]
        _lib.gbx_gen_fmi_reads.argtypes = [u64, i64, i64, vp, i64, C.c_int32, vp]
        for f in ("abea_model", "abea_counts_many", "abea_fill_many", "fmi_genome", "fmi_reads"):
            getattr(_lib, "gbx_gen_" + f).restype = None
        for f in ("chain_counts_many", "chain_fill_many", "chain_fill_real_many", "phmm_counts_many", "phmm_lengths_many", "phmm_fill_many",
                  "poa_counts_many", "poa_many"):
            getattr(_lib, "gbx_gen_" + f).restype = None
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def gen_bsw(n_pairs, seed, first=0):
    """bsw 'small' = (100_000, seed 1001); 'large' = (2_000_000, seed 1002).  Returns a BswBatch."""
    from ..bsw import BswBatch
    L = _L()
    len1 = np.zeros(n_pairs, dtype=np.int32)
    len2 = np.zeros(n_pairs, dtype=np.int32)
    h0 = np.zeros(n_pairs, dtype=np.int32)
    L.gbx_gen_bsw_lengths(seed, first, n_pairs, _p(len1), _p(len2), _p(h0))
    a1 = (len1.astype(np.int64) + 3) & ~3
    a2 = (len2.astype(np.int64) + 3) & ~3
    idr = np.concatenate([[0], np.cumsum(a1)[:-1]]).astype(np.int64) if n_pairs else np.zeros(0, np.int64)
    idq = np.concatenate([[0], np.cumsum(a2)[:-1]]).astype(np.int64) if n_pairs else np.zeros(0, np.int64)
    ref = np.zeros(int(a1.sum()) + 4, dtype=np.uint8)
    qer = np.zeros(int(a2.sum()) + 4, dtype=np.uint8)
    L.gbx_gen_bsw_fill(seed, first, n_pairs, _p(len1), _p(len2), _p(idr), _p(idq), _p(ref), _p(qer))
    return BswBatch(ref, qer, idr, idq, len1, len2, h0)


def write_bsw_pairs_fast(path, b):
    """The reference's bsw input file for a BswBatch (same bytes as io.write_bsw_pairs, written by the C library)."""
    L = _L()
    L.gbx_write_bsw_pairs.restype = C.c_int64
    got = L.gbx_write_bsw_pairs(str(path).encode(), C.c_int64(b.n), _p(b.ref), _p(b.idr), _p(b.len1), _p(b.qer), _p(b.idq), _p(b.len2), _p(b.h0))
    if got < 0:
        raise OSError("cannot write %s" % path)
    return got


def gen_chain(n_calls, seed, first=0, n_override=None, realistic=False):
    """chain 'large' = (10_000, seed 2001).  Returns (anchor_off, ax, ay, hdr).  realistic=True: the same call sizes with
    minimap2's structure inside a call (both strands, six reference ids in the upper x word, repeat copies, isolated
    hits: datagen.c gbx_gen_chain_fill_real)."""
    from .._native import CHAIN_CALL_DTYPE
    L = _L()
    if n_override is None:
        counts = np.zeros(n_calls, dtype=np.int64)
        L.gbx_gen_chain_counts_many(seed, first, n_calls, _p(counts))
    else:
        counts = np.array([n_override[c] for c in range(n_calls)], dtype=np.int64)
    off = np.zeros(n_calls + 1, dtype=np.int64)
    np.cumsum(counts, out=off[1:])
    ax = np.zeros(int(off[-1]), dtype=np.uint64)
    ay = np.zeros(int(off[-1]), dtype=np.uint64)
    (L.gbx_gen_chain_fill_real_many if realistic else L.gbx_gen_chain_fill_many)(seed, first, n_calls, _p(off), _p(ax), _p(ay))
    hdr = np.zeros(n_calls, dtype=CHAIN_CALL_DTYPE)
    hdr["avg_qspan"] = 15.0
    hdr["max_dist_x"] = 5000
    hdr["max_dist_y"] = 5000
    hdr["bw"] = 500
    hdr["n_segs"] = 1
    return off, ax, ay, hdr


def gen_phmm(n_batches, seed, first=0):
    """phmm 'large' = (20_000 batches, seed 3001).  Returns a PhmmBatchSet (flat arenas + pair list)."""
    from ..phmm import PhmmBatchSet
    L = _L()
    nr = np.zeros(n_batches, dtype=np.int32)
    nh = np.zeros(n_batches, dtype=np.int32)
    L.gbx_gen_phmm_counts_many(seed, first, n_batches, _p(nr), _p(nh))
    roff = np.zeros(n_batches + 1, dtype=np.int64); np.cumsum(nr, out=roff[1:])
    hoff = np.zeros(n_batches + 1, dtype=np.int64); np.cumsum(nh, out=hoff[1:])
    read_len = np.zeros(int(roff[-1]), dtype=np.int32)
    hap_len = np.zeros(int(hoff[-1]), dtype=np.int32)
    L.gbx_gen_phmm_lengths_many(seed, first, n_batches, _p(roff), _p(hoff), _p(read_len), _p(hap_len))
    read_off = np.zeros(len(read_len) + 1, dtype=np.int64); np.cumsum(read_len, out=read_off[1:])
    hap_off = np.zeros(len(hap_len) + 1, dtype=np.int64); np.cumsum(hap_len, out=hap_off[1:])
    rs, q, qi, qd, qc = (np.zeros(int(read_off[-1]) + 8, dtype=np.uint8) for _ in range(5))
    hap = np.zeros(int(hap_off[-1]) + 8, dtype=np.uint8)
    L.gbx_gen_phmm_fill_many(seed, first, n_batches, _p(roff), _p(hoff), _p(read_off), _p(hap_off), _p(rs), _p(q),
                             _p(qi), _p(qd), _p(qc), _p(hap))
    return PhmmBatchSet(nr, nh, read_off[:-1].copy(), read_len, rs, q, qi, qd, qc, hap_off[:-1].copy(), hap_len, hap)


def gen_poa(n_windows, seed, first=0):
    """poa 'large' = (6_000 windows, seed 4001).  Returns a PoaWindowSet."""
    from ..poa import PoaWindowSet
    L = _L()
    nr = np.zeros(n_windows, dtype=np.int32)
    L.gbx_gen_poa_counts_many(seed, first, n_windows, _p(nr))
    wf = np.zeros(n_windows + 1, dtype=np.int64); np.cumsum(nr, out=wf[1:])
    lens = np.zeros(int(wf[-1]), dtype=np.int32)
    off = np.zeros(len(lens) + 1, dtype=np.int64)
    L.gbx_gen_poa_many(seed, first, n_windows, 1, _p(wf), _p(lens), _p(off), None)
    np.cumsum(lens, out=off[1:])
    arena = np.zeros(int(off[-1]) + 8, dtype=np.uint8)
    L.gbx_gen_poa_many(seed, first, n_windows, 2, _p(wf), _p(lens), _p(off), _p(arena))
    return PoaWindowSet(wf, off[:-1].copy(), lens, arena)


def gen_abea(n_reads, seed, first=0):
    """abea 'large' = (10000 reads, seed 5001), the read count of the reference's large input.  Returns an AbeaReadSet (synthetic pore model, reads, events, scalings)."""
    from ..abea import AbeaReadSet, make_model
    L = _L()
    lm, ls = np.zeros(4096, np.float32), np.zeros(4096, np.float32)
    L.gbx_gen_abea_model(seed, _p(lm), _p(ls))
    seq_len = np.zeros(n_reads, dtype=np.int32)
    n_ev = np.zeros(n_reads, dtype=np.int64)
    L.gbx_gen_abea_counts_many(seed, first, n_reads, _p(seq_len), _p(n_ev))
    seq_off = np.zeros(n_reads + 1, dtype=np.int64); np.cumsum(seq_len, out=seq_off[1:])
    event_off = np.zeros(n_reads + 1, dtype=np.int64); np.cumsum(n_ev, out=event_off[1:])
    seq = np.zeros(int(seq_off[-1]) + 8, dtype=np.uint8)
    ev = np.zeros(int(event_off[-1]) + 4, dtype=np.float32)
    scale, shift = np.zeros(n_reads, np.float32), np.zeros(n_reads, np.float32)
    L.gbx_gen_abea_fill_many(seed, first, n_reads, _p(lm), _p(ls), _p(seq_off), _p(event_off), _p(seq), _p(ev), _p(scale), _p(shift))
    return AbeaReadSet(seq_off[:-1].copy(), seq_len, seq, event_off, ev[:int(event_off[-1])], scale, shift, make_model(lm, ls))


def gen_fmi_genome(length, seed):
    """Synthetic genome (one strand, base codes 0..3) with repeat families and low-complexity runs."""
    L = _L()
    ref = np.zeros(int(length), dtype=np.uint8)
    L.gbx_gen_fmi_genome(seed, int(length), _p(ref))
    return ref


def gen_fmi_reads(ref, n_reads, seed, first=0, read_len=151):
    """fmi reads: 151-bp samples of the genome (either strand, 1 % errors, a few N), the reference's fixed-stride layout."""
    from ..fmi import FmiReadSet
    L = _L()
    enc = np.zeros((int(n_reads), int(read_len)), dtype=np.uint8)
    L.gbx_gen_fmi_reads(seed, first, int(n_reads), _p(ref), len(ref), int(read_len), _p(enc))
    return FmiReadSet.fixed(enc)
