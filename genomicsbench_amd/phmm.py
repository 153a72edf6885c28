"""Host mirror of the reference phmm interface (R/benchmarks/phmm/PairHMMUnitTest.cpp:84-86, 224-247).

``computelikelihoodsboth`` over flat arenas: reads (bases + four quality tracks sharing offsets),
haplotypes, and a pair list built read-major / hap-minor per batch exactly as the driver builds its
``testcase`` array (PairHMMUnitTest.cpp:232-244).  All arithmetic happens in libgbx.so on the GPU.
"""
import ctypes as C

import numpy as np

from . import _native as N


class PhmmBatchSet:
    def __init__(self, n_reads, n_haps, read_off, read_len, rs, q, qi, qd, qc, hap_off, hap_len, hap):
        u8 = lambda a: np.ascontiguousarray(a, dtype=np.uint8)
        self.n_reads = np.ascontiguousarray(n_reads, dtype=np.int32)      # per batch
        self.n_haps = np.ascontiguousarray(n_haps, dtype=np.int32)
        self.read_off = np.ascontiguousarray(read_off, dtype=np.int64)
        self.read_len = np.ascontiguousarray(read_len, dtype=np.int32)
        self.rs, self.q, self.qi, self.qd, self.qc = u8(rs), u8(q), u8(qi), u8(qd), u8(qc)
        self.hap_off = np.ascontiguousarray(hap_off, dtype=np.int64)
        self.hap_len = np.ascontiguousarray(hap_len, dtype=np.int32)
        self.hap = u8(hap)
        # pair list: batch-major, read-major, hap-minor
        nb = len(self.n_reads)
        hbase = np.concatenate([[0], np.cumsum(self.n_haps)])[:-1] if nb else np.zeros(0, np.int64)
        per_read = np.repeat(self.n_haps.astype(np.int64), self.n_reads)          # pairs of each read
        start = np.concatenate([[0], np.cumsum(per_read)])
        total = int(start[-1])
        self.pair_read = np.ascontiguousarray(np.repeat(np.arange(len(per_read), dtype=np.int32), per_read))
        first_hap = np.repeat(hbase.astype(np.int64), self.n_reads)                # first haplotype of each read's batch
        self.pair_hap = np.ascontiguousarray(np.arange(total, dtype=np.int64) - np.repeat(start[:-1], per_read)
                                             + np.repeat(first_hap, per_read), dtype=np.int32)
        self.n_pairs = len(self.pair_read)
        self.batch_pair_off = np.concatenate([[0], np.cumsum(self.n_reads.astype(np.int64) * self.n_haps)])

    @property
    def cells(self):
        """sum over pairs of rslen*haplen."""
        return int((self.read_len[self.pair_read].astype(np.int64) * self.hap_len[self.pair_hap]).sum())

    @property
    def algorithmic_bytes(self):
        """5*rslen per read + haplen per hap (each read once, shared by the batch's pairs) + 8 B per result."""
        return int(5 * self.read_len.astype(np.int64).sum() + self.hap_len.astype(np.int64).sum() + 8 * self.n_pairs)

    def take_batches(self, lo, hi):
        """Sub-set of whole batches [lo,hi) (arenas shared, offsets kept)."""
        rb = np.concatenate([[0], np.cumsum(self.n_reads)])
        hb = np.concatenate([[0], np.cumsum(self.n_haps)])
        return PhmmBatchSet(self.n_reads[lo:hi], self.n_haps[lo:hi], self.read_off[rb[lo]:rb[hi]],
                            self.read_len[rb[lo]:rb[hi]], self.rs, self.q, self.qi, self.qd, self.qc,
                            self.hap_off[hb[lo]:hb[hi]], self.hap_len[hb[lo]:hb[hi]], self.hap)


def forward_host(bs):
    """gbx_phmm_forward_host -> float64[n_pairs] log10 likelihoods."""
    out = np.zeros(bs.n_pairs, dtype=np.float64)
    N.check(N.lib().gbx_phmm_forward_host(
        bs.n_pairs, N.ptr(bs.pair_read), N.ptr(bs.pair_hap),
        len(bs.read_len), N.ptr(bs.read_off), N.ptr(bs.read_len), bs.rs.size,
        N.ptr(bs.rs), N.ptr(bs.q), N.ptr(bs.qi), N.ptr(bs.qd), N.ptr(bs.qc),
        len(bs.hap_len), N.ptr(bs.hap_off), N.ptr(bs.hap_len), bs.hap.size, N.ptr(bs.hap), N.ptr(out)))
    return out


computelikelihoodsboth = forward_host


class DevicePhmmBatchSet:
    """A PhmmBatchSet resident in HBM (torch tensors) + output and workspace."""

    def __init__(self, bs, device):
        import torch
        t = lambda a: torch.from_numpy(np.concatenate([a, np.zeros(16, a.dtype)])).to(device)
        d = {k: t(getattr(bs, k)) for k in ("read_off", "read_len", "rs", "q", "qi", "qd", "qc", "hap_off", "hap_len", "hap")}
        self._init(d, t(bs.pair_read), t(bs.pair_hap), bs.n_pairs, len(bs.read_len), len(bs.hap_len),
                   int(bs.hap_len.max()) if len(bs.hap_len) else 1, device)

    @classmethod
    def from_tensors(cls, d, device):
        """Device tensors as shard.scatter_arrays delivers them: the batch table (n_reads, n_haps) and the arenas;
        the pair list (read-major, hap-minor per batch, PairHMMUnitTest.cpp:232-244) is rebuilt on the device."""
        import torch
        nr, nh = d["n_reads"].long(), d["n_haps"].long()
        per_read = torch.repeat_interleave(nh, nr)                              # pairs of each read
        start = torch.cumsum(per_read, 0) - per_read
        n_pairs = int(per_read.sum().item()) if per_read.numel() else 0
        hbase = torch.cumsum(nh, 0) - nh
        first_hap = torch.repeat_interleave(hbase, nr)
        pair_read = torch.repeat_interleave(torch.arange(per_read.numel(), device=device), per_read)
        pair_hap = (torch.arange(n_pairs, device=device) - torch.repeat_interleave(start, per_read)
                    + torch.repeat_interleave(first_hap, per_read))
        pad = torch.zeros(16, dtype=torch.int32, device=device)
        self = cls.__new__(cls)
        self._init(d, torch.cat([pair_read.int(), pad]), torch.cat([pair_hap.int(), pad]), n_pairs,
                   int(d["read_len"].numel()), int(d["hap_len"].numel()),
                   int(d["hap_len"].max().item()) if d["hap_len"].numel() else 1, device)
        return self

    def _init(self, d, pair_read, pair_hap, n_pairs, n_reads, n_haps, max_hap_len, device):
        import torch
        self.n_pairs, self.n_reads, self.n_haps = n_pairs, n_reads, n_haps
        self.pair_read, self.pair_hap = pair_read, pair_hap
        self.read_off, self.read_len = d["read_off"], d["read_len"]
        self.rs, self.q, self.qi, self.qd, self.qc = d["rs"], d["q"], d["qi"], d["qd"], d["qc"]
        self.hap_off, self.hap_len, self.hap = d["hap_off"], d["hap_len"], d["hap"]
        self.out = torch.empty(max(self.n_pairs, 1), dtype=torch.float64, device=device)
        self.max_hap_len = max_hap_len
        self.work_bytes = N.lib().gbx_phmm_workspace_bytes(self.n_pairs, self.n_reads, self.max_hap_len)
        self.work = torch.empty(self.work_bytes, dtype=torch.uint8, device=device)

    def run(self, stream=None):
        N.check(N.lib().gbx_phmm_forward_device(
            self.n_pairs, self.pair_read.data_ptr(), self.pair_hap.data_ptr(), self.n_reads, self.read_off.data_ptr(),
            self.read_len.data_ptr(), self.rs.data_ptr(), self.q.data_ptr(), self.qi.data_ptr(), self.qd.data_ptr(),
            self.qc.data_ptr(), self.hap_off.data_ptr(), self.hap_len.data_ptr(), self.hap.data_ptr(),
            self.max_hap_len, self.out.data_ptr(), self.work.data_ptr(), self.work_bytes, stream))

    def results(self):
        return self.out[:self.n_pairs].cpu().numpy()
