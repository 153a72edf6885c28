"""Host mirror of the reference poa interface (spoa calls at R/benchmarks/poa/msa_spoa_omp.cpp:184-252).

A window ("Batch" in the driver, :42-47) is a list of sequences; the product computes one consensus
string per window on the GPU through libgbx.so.
"""
import ctypes as C

import numpy as np

from . import _native as N


class PoaParams(C.Structure):
    _fields_ = [("m", C.c_int8), ("n", C.c_int8), ("g", C.c_int8), ("e", C.c_int8), ("q", C.c_int8), ("c", C.c_int8),
                ("pad_", C.c_int8 * 2)]


def make_params(m=2, x=4, o1=4, e1=2, o2=24, e2=1):
    """Driver CLI semantics (-m, -x, -o a,b, -e a,b): g = -(o1+e1), e = -e1, q = -(o2+e2), c = -e2."""
    return PoaParams(m, -x, -(o1 + e1), -e1, -(o2 + e2), -e2)


class PoaWindowSet:
    def __init__(self, win_first_seq, seq_off, seq_len, arena):
        self.win_first_seq = np.ascontiguousarray(win_first_seq, dtype=np.int64)
        self.seq_off = np.ascontiguousarray(seq_off, dtype=np.int64)
        self.seq_len = np.ascontiguousarray(seq_len, dtype=np.int32)
        self.arena = np.ascontiguousarray(arena, dtype=np.uint8)
        self.n_windows = len(self.win_first_seq) - 1
        self.n_seqs = len(self.seq_len)

    @staticmethod
    def from_lists(windows):
        """windows: list of lists of python strings."""
        lens = [len(s) for w in windows for s in w]
        wf = np.concatenate([[0], np.cumsum([len(w) for w in windows])]).astype(np.int64)
        off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        arena = np.frombuffer(("".join(s for w in windows for s in w) + "\\0" * 8).encode(), dtype=np.uint8).copy()
        return PoaWindowSet(wf, off[:-1], lens, arena)

    def window(self, w):
        a, b = int(self.win_first_seq[w]), int(self.win_first_seq[w + 1])
        return [self.arena[self.seq_off[s]:self.seq_off[s] + self.seq_len[s]].tobytes().decode() for s in range(a, b)]

    def take(self, lo, hi):
        a, b = int(self.win_first_seq[lo]), int(self.win_first_seq[hi])
        return PoaWindowSet(self.win_first_seq[lo:hi + 1] - a, self.seq_off[a:b], self.seq_len[a:b], self.arena)

    @property
    def default_stride(self):
        return int(2 * (self.seq_len.max() if self.n_seqs else 0) + 64)


def consensus_host(params, ws, stride=None):
    """gbx_poa_consensus_host -> list of consensus strings (one per window)."""
    stride = stride or ws.default_stride
    cons = np.zeros((max(ws.n_windows, 1), stride), dtype=np.uint8)
    clen = np.zeros(max(ws.n_windows, 1), dtype=np.int32)
    N.check(N.lib().gbx_poa_consensus_host(C.byref(params), ws.n_windows, N.ptr(ws.win_first_seq), ws.n_seqs,
                                           N.ptr(ws.seq_off), N.ptr(ws.seq_len), N.ptr(ws.arena), ws.arena.size,
                                           N.ptr(cons), N.ptr(clen), stride))
    return [cons[w, :clen[w]].tobytes().decode() for w in range(ws.n_windows)]


class PoaPlan(C.Structure):
    _fields_ = [("max_seq_len", C.c_int32), ("max_seqs_per_window", C.c_int32), ("node_cap", C.c_int32),
                ("n_slots", C.c_int32), ("n_long_windows", C.c_int32), ("long_slots", C.c_int32), ("n_windows", C.c_int64)]


class DevicePoaWindowSet:
    """A PoaWindowSet resident in HBM (torch tensors) + outputs, per-window status and the workspace."""

    def __init__(self, ws, device, stride=None):
        import torch
        t = lambda a: torch.from_numpy(a).to(device)
        self._init(dict(win_first_seq=t(ws.win_first_seq), seq_off=t(ws.seq_off), seq_len=t(ws.seq_len),
                        arena=t(np.concatenate([ws.arena, np.zeros(16, np.uint8)]))),
                   ws.win_first_seq, ws.seq_len, stride or ws.default_stride, device)
        self.ws = ws

    @classmethod
    def from_tensors(cls, d, device, stride=None):
        """Device tensors as shard.scatter_arrays delivers them; the plan is made from the (small) host copies of
        the window table and the sequence lengths."""
        wf = d["win_first_seq"].cpu().numpy()
        sl = d["seq_len"].cpu().numpy()
        self = cls.__new__(cls)
        self._init(d, wf, sl, stride or int(2 * (sl.max() if len(sl) else 0) + 64), device)
        self.ws = None
        return self

    def _init(self, d, wf_host, seq_len_host, stride, device):
        import torch
        wf_host = np.ascontiguousarray(wf_host, dtype=np.int64)
        seq_len_host = np.ascontiguousarray(seq_len_host, dtype=np.int32)
        self.n_windows = len(wf_host) - 1
        self.stride = stride
        self.plan = PoaPlan()
        N.check(N.lib().gbx_poa_plan_host(self.n_windows, N.ptr(wf_host), N.ptr(seq_len_host), C.byref(self.plan)))
        self.win_first_seq, self.seq_off, self.seq_len, self.arena = d["win_first_seq"], d["seq_off"], d["seq_len"], d["arena"]
        n = max(self.n_windows, 1)
        self.cons = torch.zeros((n, self.stride), dtype=torch.uint8, device=device)
        self.cons_len = torch.zeros(n, dtype=torch.int32, device=device)
        self.status = torch.zeros(n, dtype=torch.int32, device=device)
        self.work_bytes = N.lib().gbx_poa_workspace_bytes(C.byref(self.plan))
        self.work = torch.empty(self.work_bytes, dtype=torch.uint8, device=device)

    def run(self, params, stream=None):
        N.check(N.lib().gbx_poa_consensus_device(C.byref(params), C.byref(self.plan), self.n_windows,
                                                 self.win_first_seq.data_ptr(), self.seq_off.data_ptr(),
                                                 self.seq_len.data_ptr(), self.arena.data_ptr(), self.cons.data_ptr(),
                                                 self.cons_len.data_ptr(), self.status.data_ptr(), self.stride,
                                                 self.work.data_ptr(), self.work_bytes, stream))

    def results(self):
        cons = self.cons.cpu().numpy()
        clen = self.cons_len.cpu().numpy()
        st = self.status.cpu().numpy()
        if st[:self.n_windows].any():
            raise N.GbxError(N.GBX_ERR_UNSUPPORTED, "window %d overflowed a device capacity" % int(np.nonzero(st)[0][0]))
        return [cons[w, :clen[w]].tobytes().decode() for w in range(self.n_windows)]

    def cells(self, stream=None):
        """DP cells of the last run() (device-side counter)."""
        v = C.c_int64(0)
        N.check(N.lib().gbx_poa_cells(C.byref(self.plan), self.work.data_ptr(), C.byref(v), stream))
        return v.value
