"""ctypes binding of oracle/liboracle.so and (when present) oracle/_ref/*.so.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py — never by the genomicsbench_amd package.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_cache = {}


def _load(path):
    if path not in _cache:
        _cache[path] = C.CDLL(path) if os.path.exists(path) else None
    return _cache[path]


def oracle_lib():
    L = _load(os.path.join(_HERE, "liboracle.so"))
    if L is None:
        raise ImportError("oracle/liboracle.so not built (make -C oracle)")
    return L


def ref_lib(name):
    """oracle/_ref/lib<name>_ref.so or None (the reference build only exists where /root/reference did)."""
    return _load(os.path.join(_HERE, "_ref", "lib%s_ref.so" % name))


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _bsw_args(params, b, out):
    return [C.byref(params), C.c_int64(b.n), _p(b.ref), _p(b.qer), _p(b.idr), _p(b.idq), _p(b.len1), _p(b.len2),
            _p(b.h0), _p(out)]


def bsw_oracle(params, batch, nthreads=1, return_cells=False):
    out = np.zeros((batch.n, 6), dtype=np.int32)
    cells = C.c_int64(0)
    f = oracle_lib().oracle_bsw_extend
    f.restype = None
    f(*_bsw_args(params, batch, out), C.c_int(nthreads), C.byref(cells))
    return (out, cells.value) if return_cells else out


def bsw_ref_scalar(params, batch):
    L = ref_lib("bsw")
    out = np.zeros((batch.n, 6), dtype=np.int32)
    L.ref_bsw_scalar(*_bsw_args(params, batch, out))
    return out


def bsw_ref_avx2(params, batch, batch_size=512):
    L = ref_lib("bsw")
    out = np.zeros((batch.n, 6), dtype=np.int32)
    L.ref_bsw_getscores16(*_bsw_args(params, batch, out), C.c_int32(batch_size))
    return out


def _chain_args(off, ax, ay, hdr, score, parent, target, peak):
    return [C.c_int64(len(off) - 1), _p(off), _p(ax), _p(ay), _p(hdr), _p(score), _p(parent), _p(target), _p(peak)]


def chain_oracle(off, ax, ay, hdr, nthreads=1, return_pairs=False):
    n = int(off[-1])
    score, parent, target, peak = (np.zeros(n, dtype=np.int32) for _ in range(4))
    ev = C.c_int64(0)
    f = oracle_lib().oracle_chain
    f.restype = None
    f(*_chain_args(off, ax, ay, hdr, score, parent, target, peak), C.c_int(nthreads), C.byref(ev))
    r = (score, parent, target, peak)
    return r + (ev.value,) if return_pairs else r


def chain_ref(off, ax, ay, hdr, nthreads=1):
    L = ref_lib("chain")
    n = int(off[-1])
    score, parent, target, peak = (np.zeros(n, dtype=np.int32) for _ in range(4))
    L.ref_chain(*_chain_args(off, ax, ay, hdr, score, parent, target, peak), C.c_int(nthreads))
    return score, parent, target, peak


def phmm_oracle(bs, nthreads=1, return_ndouble=False):
    """oracle_phmm_forward over a PhmmBatchSet -> float64[n_pairs] (and the count of fp64 fallbacks)."""
    out = np.zeros(bs.n_pairs, dtype=np.float64)
    nd = C.c_int64(0)
    f = oracle_lib().oracle_phmm_forward
    f.restype = None
    f(C.c_int64(bs.n_pairs), _p(bs.pair_read), _p(bs.pair_hap), _p(bs.read_off), _p(bs.read_len), _p(bs.rs),
      _p(bs.q), _p(bs.qi), _p(bs.qd), _p(bs.qc), _p(bs.hap_off), _p(bs.hap_len), _p(bs.hap), _p(out),
      C.c_int(nthreads), C.byref(nd))
    return (out, nd.value) if return_ndouble else out


def phmm_pair(rs, hap, q, qi, qd, qc, f64_only=False):
    """One pair from python byte strings / uint8 arrays (already normalised qualities)."""
    L = oracle_lib()
    b = lambda a: bytes(bytearray(np.asarray(a, dtype=np.uint8).tolist())) if not isinstance(a, (bytes, str)) else (
        a.encode() if isinstance(a, str) else a)
    rs, hap, q, qi, qd, qc = b(rs), b(hap), b(q), b(qi), b(qd), b(qc)
    if f64_only:
        L.oracle_phmm_pair_f64.restype = C.c_double
        return L.oracle_phmm_pair_f64(len(rs), len(hap), rs, hap, q, qi, qd, qc)
    L.oracle_phmm_pair.restype = C.c_double
    ud = C.c_int(0)
    return L.oracle_phmm_pair(len(rs), len(hap), rs, hap, q, qi, qd, qc, C.byref(ud)), ud.value


def poa_oracle(params, ws, nthreads=1, stride=None, return_cells=False):
    """oracle_poa_consensus over a PoaWindowSet -> list of consensus strings."""
    stride = stride or ws.default_stride
    cons = np.zeros((max(ws.n_windows, 1), stride), dtype=np.uint8)
    clen = np.zeros(max(ws.n_windows, 1), dtype=np.int32)
    cells = C.c_int64(0)
    f = oracle_lib().oracle_poa_consensus
    f.restype = None
    f(C.byref(params), C.c_int64(ws.n_windows), _p(ws.win_first_seq), _p(ws.seq_off), _p(ws.seq_len), _p(ws.arena),
      _p(cons), _p(clen), C.c_int64(stride), C.c_int(nthreads), C.byref(cells))
    out = [cons[w, :min(clen[w], stride)].tobytes().decode() for w in range(ws.n_windows)]
    return (out, cells.value) if return_cells else out


def abea_oracle(rs, nthreads=1, return_cells=False):
    """oracle_abea_align over an AbeaReadSet -> (pairs flat structured array, n_pairs)."""
    from genomicsbench_amd.abea import PAIR_DTYPE
    out = np.zeros(2 * max(int(rs.event_off[-1]), 1), dtype=PAIR_DTYPE)
    n_pairs = np.zeros(max(rs.n_reads, 1), dtype=np.int32)
    cells = C.c_int64(0)
    f = oracle_lib().oracle_abea_align
    f.restype = None
    f(C.c_int64(rs.n_reads), _p(rs.seq_off), _p(rs.seq_len), _p(rs.seq_arena), _p(rs.event_off), _p(rs.event_mean), _p(rs.model),
      _p(rs.scale), _p(rs.shift), _p(out), _p(n_pairs), C.c_int(nthreads), C.byref(cells))
    r = (out, n_pairs[:rs.n_reads])
    return r + (cells.value,) if return_cells else r


def fmi_oracle(index, reads, params=None, nthreads=1, return_stats=False):
    """oracle_fmi_smem over an FmiIndex (host) + FmiReadSet -> (SMEM array, smem_off[, backwardExt calls, per-round counts])."""
    from genomicsbench_amd.fmi import SMEM_DTYPE, default_params
    params = params or default_params()
    idx = index.host()
    st = idx.struct(idx.cp_occ.ctypes.data)
    off = np.zeros(reads.n_reads + 1, dtype=np.int64)
    ext, rounds = C.c_int64(0), (C.c_int64 * 3)()
    f = oracle_lib().oracle_fmi_smem
    f.restype = C.c_int64
    args = [C.byref(st), C.byref(params), C.c_int64(reads.n_reads), _p(reads.enc), _p(reads.read_off), _p(reads.read_len)]
    cap = max(64, 24 * reads.n_reads)
    while True:
        out = np.zeros(cap, dtype=SMEM_DTYPE)
        total = f(*args, _p(out), C.c_int64(cap), _p(off), C.c_int(nthreads), C.byref(ext), rounds)
        if total <= cap:
            break
        cap = int(total)
    r = (out[:total], off)
    return r + (ext.value, [int(v) for v in rounds]) if return_stats else r


def fmi_build_index_plain(text, sa):
    """oracle_fmi_build_index: the tables from a text (codes 0..3, already reference + reverse complement) and its
    suffix array incl. the sentinel row - the plain loop the tensor builder of genomicsbench_amd.fmi is checked against."""
    from genomicsbench_amd.fmi import CP_OCC_DTYPE, FmiIndex
    text = np.ascontiguousarray(text, dtype=np.uint8)
    sa = np.ascontiguousarray(sa, dtype=np.int64)
    n = len(text)
    cp = np.zeros(((n + 1) >> 6) + 1, dtype=CP_OCC_DTYPE)
    count = np.zeros(5, dtype=np.int64)
    sent = C.c_int64(-1)
    f = oracle_lib().oracle_fmi_build_index
    f.restype = None
    f(_p(text), C.c_int64(n), _p(sa), _p(cp), _p(count), C.byref(sent))
    return FmiIndex(n + 1, count, sent.value, cp)
