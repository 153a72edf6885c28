"""Host mirror of the reference bsw interface (R/benchmarks/bsw/bandedSWA.h:114-342).

``BandedPairWiseSW`` keeps the reference constructor's argument order and meaning and
offers ``getScores16`` over numpy arrays; all arithmetic happens in libgbx.so on the GPU.
"""
import ctypes as C

import numpy as np

from . import _native as N


def fill_scmat(a=1, b=4, ambig=-1):
    """bwa_fill_scmat (R/benchmarks/bsw/main_banded.cpp:73-81) -> int8[25]."""
    mat = (C.c_int8 * 25)()
    N.lib().gbx_bsw_fill_scmat(a, b, ambig, mat)
    return np.frombuffer(mat, dtype=np.int8).copy()


def make_params(o_del=6, e_del=1, o_ins=6, e_ins=1, zdrop=100, end_bonus=5, w=100, mat=None):
    p = N.BswParams()
    N.lib().gbx_bsw_default_params(C.byref(p))
    p.o_del, p.e_del, p.o_ins, p.e_ins = o_del, e_del, o_ins, e_ins
    p.zdrop, p.end_bonus, p.w = zdrop, end_bonus, w
    if mat is not None:
        m = np.asarray(mat, dtype=np.int8).reshape(25)
        for k in range(25):
            p.mat[k] = int(m[k])
    return p


class BswBatch:
    """Flat-array view of a set of extension pairs (what the C-ABI consumes)."""

    def __init__(self, ref, qer, idr, idq, len1, len2, h0):
        self.ref = np.ascontiguousarray(ref, dtype=np.uint8)
        self.qer = np.ascontiguousarray(qer, dtype=np.uint8)
        self.idr = np.ascontiguousarray(idr, dtype=np.int64)
        self.idq = np.ascontiguousarray(idq, dtype=np.int64)
        self.len1 = np.ascontiguousarray(len1, dtype=np.int32)
        self.len2 = np.ascontiguousarray(len2, dtype=np.int32)
        self.h0 = np.ascontiguousarray(h0, dtype=np.int32)
        self.n = int(self.len1.shape[0])

    @property
    def nominal_cells(self):
        """Sum of len1*len2: the reference's own (commented) cell count, main_banded.cpp:183."""
        return int((self.len1.astype(np.int64) * self.len2.astype(np.int64)).sum())

    @property
    def algorithmic_bytes(self):
        """len1+len2 bases + 12 B of (len1,len2,h0) + 24 B of results per pair."""
        return int(self.len1.astype(np.int64).sum() + self.len2.astype(np.int64).sum() + 36 * self.n)

    def slice(self, lo, hi):
        return BswBatch(self.ref, self.qer, self.idr[lo:hi], self.idq[lo:hi], self.len1[lo:hi],
                        self.len2[lo:hi], self.h0[lo:hi])

    @staticmethod
    def from_sequences(targets, queries, h0):
        """Packs python lists of uint8 code arrays into 4-byte-aligned arenas."""
        def pack(seqs):
            lens = np.array([len(s) for s in seqs], dtype=np.int32)
            offs = np.zeros(len(seqs), dtype=np.int64)
            pos = 0
            for k, ln in enumerate(lens):
                offs[k] = pos
                pos += (int(ln) + 3) & ~3
            arena = np.zeros(max(pos, 4), dtype=np.uint8)
            for k, s in enumerate(seqs):
                arena[offs[k]:offs[k] + lens[k]] = s
            return arena, offs, lens
        ref, idr, l1 = pack(targets)
        qer, idq, l2 = pack(queries)
        return BswBatch(ref, qer, idr, idq, l1, l2, np.asarray(h0, dtype=np.int32))


def extend_host(params, batch, out=None):
    """gbx_bsw_extend_host -> int32[n,6] (score,tle,gtle,qle,gscore,max_off); `out` reuses a caller's array."""
    if out is None:
        out = np.zeros((batch.n, 6), dtype=np.int32)
    N.check(N.lib().gbx_bsw_extend_host(C.byref(params), batch.n, N.ptr(batch.ref), batch.ref.size,
                                        N.ptr(batch.qer), batch.qer.size, N.ptr(batch.idr), N.ptr(batch.idq),
                                        N.ptr(batch.len1), N.ptr(batch.len2), N.ptr(batch.h0), N.ptr(out)))
    return out


class DeviceBswBatch:
    """A BswBatch resident in HBM as torch tensors, plus the output and workspace buffers."""

    def __init__(self, batch, device):
        import torch
        t = lambda a: torch.from_numpy(a).to(device)
        pad = np.zeros(64, dtype=np.uint8)
        self._init(dict(ref=t(np.concatenate([batch.ref, pad])), qer=t(np.concatenate([batch.qer, pad])),
                        idr=t(batch.idr), idq=t(batch.idq), len1=t(batch.len1), len2=t(batch.len2), h0=t(batch.h0)),
                   device)

    @classmethod
    def from_tensors(cls, d, device):
        """Device tensors as shard.scatter_arrays delivers them (arenas followed by >= 16 readable bytes)."""
        self = cls.__new__(cls)
        self._init(d, device)
        return self

    def _init(self, d, device):
        import torch
        self.device = device
        self.ref, self.qer, self.idr, self.idq = d["ref"], d["qer"], d["idr"], d["idq"]
        self.len1, self.len2, self.h0 = d["len1"], d["len2"], d["h0"]
        self.n = int(self.len1.shape[0])
        self.out = torch.empty((max(self.n, 1), 6), dtype=torch.int32, device=device)
        self.work_bytes = N.lib().gbx_bsw_workspace_bytes(self.n)
        self.work = torch.empty(self.work_bytes, dtype=torch.uint8, device=device)

    def run(self, params, stream=None):
        """Asynchronous launch on `stream` (a raw hipStream_t handle or None)."""
        N.check(N.lib().gbx_bsw_extend_device(C.byref(params), self.n, self.ref.data_ptr(), self.qer.data_ptr(),
                                              self.idr.data_ptr(), self.idq.data_ptr(), self.len1.data_ptr(),
                                              self.len2.data_ptr(), self.h0.data_ptr(), self.out.data_ptr(),
                                              self.work.data_ptr(), self.work_bytes, stream))

    def results(self):
        return self.out[:self.n].cpu().numpy()


class BandedPairWiseSW:
    """Same constructor as the reference class (bandedSWA.cpp:51-100); numThreads is accepted and ignored."""

    def __init__(self, o_del, e_del, o_ins, e_ins, zdrop, end_bonus, mat, w_match=1, w_mismatch=4, numThreads=1):
        self.params = make_params(o_del, e_del, o_ins, e_ins, zdrop, end_bonus, 100, mat)

    def getScores16(self, pairs, seqBufRef, seqBufQer, numPairs=None, numThreads=1, w=100):
        """pairs: numpy structured array of SEQPAIR_DTYPE, updated in place (bandedSWA.cpp:1124-1148)."""
        assert pairs.dtype == N.SEQPAIR_DTYPE and pairs.flags["C_CONTIGUOUS"]
        n = len(pairs) if numPairs is None else int(numPairs)
        self.params.w = w
        ref = np.ascontiguousarray(seqBufRef, dtype=np.uint8)
        qer = np.ascontiguousarray(seqBufQer, dtype=np.uint8)
        N.check(N.lib().gbx_bsw_extend_seqpairs(C.byref(self.params), N.ptr(pairs), n, N.ptr(ref), ref.size,
                                                N.ptr(qer), qer.size))
        return pairs
