"""Host mirror of the reference chain interface (R/benchmarks/chain/src/host_kernel.h:6).

``host_chain_kernel(calls) -> rets`` over flat numpy arrays; all arithmetic in libgbx.so on the GPU.
"""
import ctypes as C

import numpy as np

from . import _native as N


def chain_host(off, ax, ay, hdr, want_target=True, want_peak=True, out=None):
    """gbx_chain_host -> (score, parent, target, peak) int32 arrays over the concatenated anchors.
    out: four int32 arrays of that length to write into (what a C driver hands over: memory that exists), instead of new ones."""
    off = np.ascontiguousarray(off, dtype=np.int64)
    ax = np.ascontiguousarray(ax, dtype=np.uint64)
    ay = np.ascontiguousarray(ay, dtype=np.uint64)
    hdr = np.ascontiguousarray(hdr, dtype=N.CHAIN_CALL_DTYPE)
    n = int(off[-1]) if len(off) else 0
    if out is not None:
        score, parent, target, peak = out
        for a in out:
            assert a.dtype == np.int32 and a.flags.c_contiguous and len(a) == n
    else:
        score = np.zeros(n, dtype=np.int32)
        parent = np.zeros(n, dtype=np.int32)
        target = np.zeros(n, dtype=np.int32) if want_target else None
        peak = np.zeros(n, dtype=np.int32) if want_peak else None
    N.check(N.lib().gbx_chain_host(len(off) - 1, N.ptr(off), N.ptr(ax), N.ptr(ay), N.ptr(hdr), N.ptr(score),
                                   N.ptr(parent), N.ptr(target), N.ptr(peak)))
    return score, parent, target, peak


host_chain_kernel = chain_host


class DeviceChainBatch:
    """Chaining calls resident in HBM (torch tensors) plus outputs and workspace."""

    def __init__(self, off, ax, ay, hdr, device):
        import torch
        t = lambda a: torch.from_numpy(a).to(device)
        self._init(dict(off=t(np.ascontiguousarray(off, dtype=np.int64)),
                        ax=t(np.ascontiguousarray(ax, dtype=np.uint64).view(np.int64)),
                        ay=t(np.ascontiguousarray(ay, dtype=np.uint64).view(np.int64)),
                        hdr=t(np.ascontiguousarray(hdr, dtype=N.CHAIN_CALL_DTYPE).view(np.uint8))),
                   int(off[-1]) if len(off) else 0, device)

    @classmethod
    def from_tensors(cls, d, device):
        """Device tensors as shard.scatter_arrays delivers them (off int64, ax/ay int64 views, hdr bytes)."""
        self = cls.__new__(cls)
        self._init(d, int(d["off"][-1].item()) if d["off"].numel() else 0, device)
        return self

    def _init(self, d, n_anchors, device):
        import torch
        self.off, self.ax, self.ay, self.hdr = d["off"], d["ax"], d["ay"], d["hdr"]
        self.n_calls = max(int(self.off.numel()) - 1, 0)
        self.n_anchors = n_anchors
        n = max(self.n_anchors, 1)
        self.score, self.parent, self.target, self.peak = (torch.empty(n, dtype=torch.int32, device=device)
                                                           for _ in range(4))
        self.work_bytes = N.lib().gbx_chain_workspace_bytes(self.n_calls, self.n_anchors)
        self.work = torch.empty(self.work_bytes, dtype=torch.uint8, device=device)

    def run(self, stream=None):
        N.check(N.lib().gbx_chain_device(self.n_calls, self.n_anchors, self.off.data_ptr(), self.ax.data_ptr(),
                                         self.ay.data_ptr(), self.hdr.data_ptr(), self.score.data_ptr(),
                                         self.parent.data_ptr(), self.target.data_ptr(), self.peak.data_ptr(),
                                         self.work.data_ptr(), self.work_bytes, stream))

    def results(self):
        k = self.n_anchors
        return tuple(a[:k].cpu().numpy() for a in (self.score, self.parent, self.target, self.peak))

    def job_stats(self, stream=None):
        """(jobs, anchors of the longest job) of the last run(): calls are cut into independent pieces on the device."""
        j, m = C.c_int64(0), C.c_int64(0)
        N.check(N.lib().gbx_chain_job_stats(self.work.data_ptr(), self.n_calls, self.n_anchors, C.byref(j), C.byref(m), stream))
        return j.value, m.value

    def evaluated_pairs(self, stream=None):
        """Predecessor pairs visited by the last run() (device-side counter)."""
        v = C.c_int64(0)
        N.check(N.lib().gbx_chain_evaluated_pairs(self.work.data_ptr(), C.byref(v), stream))
        return v.value
