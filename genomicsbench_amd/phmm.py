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
        rbase = np.concatenate([[0], np.cumsum(self.n_reads)])[:-1] if nb else np.zeros(0, np.int64)
        hbase = np.concatenate([[0], np.cumsum(self.n_haps)])[:-1] if nb else np.zeros(0, np.int64)
        pr, ph = [], []
        for b in range(nb):
            r = np.arange(self.n_reads[b], dtype=np.int32) + np.int32(rbase[b])
            h = np.arange(self.n_haps[b], dtype=np.int32) + np.int32(hbase[b])
            pr.append(np.repeat(r, self.n_haps[b]))
            ph.append(np.tile(h, self.n_reads[b]))
        self.pair_read = np.ascontiguousarray(np.concatenate(pr) if pr else np.zeros(0), dtype=np.int32)
        self.pair_hap = np.ascontiguousarray(np.concatenate(ph) if ph else np.zeros(0), dtype=np.int32)
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
        self.n_pairs, self.n_reads, self.n_haps = bs.n_pairs, len(bs.read_len), len(bs.hap_len)
        self.pair_read, self.pair_hap = t(bs.pair_read), t(bs.pair_hap)
        self.read_off, self.read_len = t(bs.read_off), t(bs.read_len)
        self.rs, self.q, self.qi, self.qd, self.qc = t(bs.rs), t(bs.q), t(bs.qi), t(bs.qd), t(bs.qc)
        self.hap_off, self.hap_len, self.hap = t(bs.hap_off), t(bs.hap_len), t(bs.hap)
        self.out = torch.empty(max(self.n_pairs, 1), dtype=torch.float64, device=device)
        self.max_hap_len = int(bs.hap_len.max()) if len(bs.hap_len) else 1
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
