"""Host mirror of the reference abea interface (R/benchmarks/abea/src/align.c:169-171, f5c.c:1344-1349).

``align`` for a set of reads: per read the aligned (k-mer index, event index) pairs of the adaptive banded event
alignment, or none when a QC rule failed.  All arithmetic happens in libgbx.so on the GPU.
"""
import ctypes as C

import numpy as np

from . import _native as N

MODEL_DTYPE = np.dtype([("level_mean", "<f4"), ("level_stdv", "<f4"), ("level_log_stdv", "<f4")])          # model_t, f5c.h:122-136
EVENT_DTYPE = np.dtype({"names": ["start", "length", "mean", "stdv"], "formats": ["<u8", "<f4", "<f4", "<f4"],
                        "offsets": [0, 8, 12, 16], "itemsize": 24})                                           # event_t, f5c.h:104-111
PAIR_DTYPE = np.dtype([("ref_pos", "<i4"), ("read_pos", "<i4")])                                             # AlignedPair, f5c.h:163-166
KMER = 6


def make_model(level_mean, level_stdv):
    """model_t table with the cached log (model.c:53: level_log_stdv = log(level_stdv), a double log stored as float)."""
    m = np.zeros(4096, dtype=MODEL_DTYPE)
    m["level_mean"] = level_mean
    m["level_stdv"] = level_stdv
    m["level_log_stdv"] = np.log(m["level_stdv"].astype(np.float64)).astype(np.float32)
    return m


class AbeaReadSet:
    """Reads (bases), their events (means are all align() reads), scalings and the pore model."""

    def __init__(self, seq_off, seq_len, seq_arena, event_off, event_mean, scale, shift, model):
        self.seq_off = np.ascontiguousarray(seq_off, dtype=np.int64)
        self.seq_len = np.ascontiguousarray(seq_len, dtype=np.int32)
        self.seq_arena = np.ascontiguousarray(seq_arena, dtype=np.uint8)
        self.event_off = np.ascontiguousarray(event_off, dtype=np.int64)          # n_reads + 1
        self.event_mean = np.ascontiguousarray(event_mean, dtype=np.float32)
        self.scale = np.ascontiguousarray(scale, dtype=np.float32)
        self.shift = np.ascontiguousarray(shift, dtype=np.float32)
        self.model = np.ascontiguousarray(model, dtype=MODEL_DTYPE)
        self.n_reads = len(self.seq_len)

    @property
    def n_events(self):
        return np.diff(self.event_off)

    @property
    def n_bands(self):
        """bands per read: (n_events + 1) + (n_kmers + 1), align.c:209-211."""
        return self.n_events + 1 + (self.seq_len.astype(np.int64) - KMER + 1) + 1

    @property
    def band_cells(self):
        """upper bound of the DP cells: bands x bandwidth (the filled ones are counted on the device)."""
        return int(self.n_bands.sum()) * 100

    def events_struct(self):
        """the 24-byte event_t array a reference caller holds (only `mean` is meaningful here)."""
        ev = np.zeros(len(self.event_mean), dtype=EVENT_DTYPE)
        ev["mean"] = self.event_mean
        return ev

    def take(self, lo, hi):
        a, b = int(self.event_off[lo]), int(self.event_off[hi])
        return AbeaReadSet(self.seq_off[lo:hi], self.seq_len[lo:hi], self.seq_arena, self.event_off[lo:hi + 1] - a,
                           self.event_mean[a:b], self.scale[lo:hi], self.shift[lo:hi], self.model)

    def split_pairs(self, out, n_pairs):
        """flat output arrays -> list of (n_pairs[r], 2) arrays"""
        return [out[2 * int(self.event_off[r]):2 * int(self.event_off[r]) + int(n_pairs[r])].copy() for r in range(self.n_reads)]


def align_host(rs):
    """gbx_abea_align_host -> (pairs structured array of 2*total_events entries, n_pairs int32[n_reads])."""
    out = np.zeros(2 * max(int(rs.event_off[-1]), 1), dtype=PAIR_DTYPE)
    n_pairs = np.zeros(max(rs.n_reads, 1), dtype=np.int32)
    ev = rs.events_struct()
    N.check(N.lib().gbx_abea_align_host(rs.n_reads, N.ptr(rs.seq_off), N.ptr(rs.seq_len), N.ptr(rs.seq_arena), rs.seq_arena.size,
                                        N.ptr(rs.event_off), N.ptr(ev), N.ptr(rs.model), N.ptr(rs.scale), N.ptr(rs.shift),
                                        N.ptr(out), N.ptr(n_pairs)))
    return out, n_pairs[:rs.n_reads]


class DeviceAbeaReadSet:
    """An AbeaReadSet resident in HBM (torch tensors) + outputs, the band plan and the workspace."""

    def __init__(self, rs, device):
        import torch
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
        self._init(dict(seq_off=t(rs.seq_off), seq_len=t(rs.seq_len), seq_arena=t(np.concatenate([rs.seq_arena, np.zeros(16, np.uint8)])),
                        event_off=t(rs.event_off), event_mean=t(np.concatenate([rs.event_mean, np.zeros(4, np.float32)])),
                        scale=t(rs.scale), shift=t(rs.shift), model=t(rs.model.view(np.uint8))),
                   rs.seq_len, rs.event_off, device)

    @classmethod
    def from_tensors(cls, d, device):
        """Device tensors as shard.scatter_arrays delivers them; the plan is made from host copies of the two small tables."""
        self = cls.__new__(cls)
        self._init(d, d["seq_len"].cpu().numpy(), d["event_off"].cpu().numpy(), device)
        return self

    def _init(self, d, seq_len_host, event_off_host, device):
        import torch
        self.d = d
        seq_len_host = np.ascontiguousarray(seq_len_host, dtype=np.int32)
        event_off_host = np.ascontiguousarray(event_off_host, dtype=np.int64)
        self.n_reads = len(seq_len_host)
        self.n_events_total = int(event_off_host[-1]) if len(event_off_host) else 0
        band_off = np.zeros(self.n_reads + 1, dtype=np.int64)
        order = np.zeros(max(self.n_reads, 1), dtype=np.int32)
        lp = np.zeros((max(self.n_reads, 1), 2), dtype=np.float64)
        N.check(N.lib().gbx_abea_plan_host(self.n_reads, N.ptr(seq_len_host), N.ptr(event_off_host), N.ptr(band_off), N.ptr(order), N.ptr(lp)))
        self.band_off = torch.from_numpy(band_off).to(device)
        self.order = torch.from_numpy(order).to(device)
        self.lp = torch.from_numpy(lp).to(device)
        self.n_kmers_total = int((seq_len_host.astype(np.int64) - KMER + 1).sum())
        self.n_bands_total = int(band_off[-1])
        self.work_bytes = N.lib().gbx_abea_workspace_bytes(self.n_reads, self.n_kmers_total, self.n_bands_total)
        self.work = torch.empty(self.work_bytes, dtype=torch.uint8, device=device)
        self.out = torch.zeros((2 * max(self.n_events_total, 1), 2), dtype=torch.int32, device=device)
        self.n_pairs = torch.zeros(max(self.n_reads, 1), dtype=torch.int32, device=device)

    def run(self, stream=None):
        d = self.d
        N.check(N.lib().gbx_abea_align_device(self.n_reads, d["seq_off"].data_ptr(), d["seq_len"].data_ptr(), d["seq_arena"].data_ptr(),
                                              d["event_off"].data_ptr(), d["event_mean"].data_ptr(), d["model"].data_ptr(),
                                              d["scale"].data_ptr(), d["shift"].data_ptr(), self.band_off.data_ptr(),
                                              self.order.data_ptr(), self.lp.data_ptr(), self.n_kmers_total, self.n_bands_total,
                                              self.out.data_ptr(), self.n_pairs.data_ptr(),
                                              self.work.data_ptr(), self.work_bytes, stream))

    def results(self):
        return self.out.cpu().numpy().view(PAIR_DTYPE).reshape(-1), self.n_pairs[:self.n_reads].cpu().numpy()

    def cells(self, stream=None):
        v = C.c_int64(0)
        N.check(N.lib().gbx_abea_cells(self.work.data_ptr(), C.byref(v), stream))
        return v.value
