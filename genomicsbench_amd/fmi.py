"""Host mirror of the reference fmi interface (R/benchmarks/fmi/fmi.cpp:79-80,180-286: FMI_search::load_index and the
three seeding rounds per batch of reads).

``build_index`` makes the tables FMI_search::load_index reads (count[], the CP_OCC checkpoints of 64 BWT symbols, the
sentinel row) for reference + reverse complement - the job of `bwa-mem2 index`, which is outside the benchmark's
timed region; tensor arithmetic only (suffix array by prefix doubling over torch.sort), so it runs on the GPU for the
bench-sized genome and on the CPU in the tests.  ``smem_host`` / ``DeviceFmi`` call libgbx.so; all seeding
arithmetic happens there on the GPU.
"""
import ctypes as C

import numpy as np

from . import _native as N

CP_OCC_DTYPE = np.dtype([("cp_count", "<i8", (4,)), ("one_hot_bwt_str", "<u8", (4,))])          # CP_OCC, FMI_search.h (64 B)
SMEM_DTYPE = np.dtype([("rid", "<u4"), ("m", "<u4"), ("n", "<u4"), ("pad_", "<u4"), ("k", "<i8"), ("l", "<i8"), ("s", "<i8")])   # SMEM (40 B)


class FmiIndexStruct(C.Structure):       # gbx_fmi_index
    _fields_ = [("ref_seq_len", C.c_int64), ("count", C.c_int64 * 5), ("sentinel_index", C.c_int64), ("cp_occ", C.c_void_p)]


class FmiParams(C.Structure):            # gbx_fmi_params
    _fields_ = [("min_seed_len", C.c_int32), ("split_width", C.c_int32), ("split_len", C.c_int32), ("max_mem_intv", C.c_int32)]


def default_params(min_seed_len=19):
    """fmi.cpp:135-140,178: splitWidth 10, maxMemIntv 20, split_len = (int)(minSeedLen * 1.5 + .499)."""
    return FmiParams(min_seed_len, 10, int(min_seed_len * 1.5 + .499), 20)


def suffix_array(text, device=None):
    """Suffix array of text + sentinel (row 0 = the sentinel suffix), text = base codes 0..3, by prefix doubling:
    ranks of the first 16 symbols from a base-5 number, then rank pairs (rank[i], rank[i + k]) sorted with torch.sort
    until every rank is unique.  O(n log n) sorts; int64 keys, so (n + 2)^2 must stay below 2^63."""
    import torch
    t = text if isinstance(text, torch.Tensor) else torch.as_tensor(np.ascontiguousarray(text, dtype=np.uint8))
    dev = torch.device(device) if device is not None else t.device
    t = t.to(dev)
    n1 = t.numel() + 1
    sym = torch.zeros(n1 + 16, dtype=torch.int64, device=dev)
    sym[:n1 - 1] = t.to(torch.int64) + 1                  # sentinel and everything behind it: 0, the smallest
    key = torch.zeros(n1, dtype=torch.int64, device=dev)
    for d in range(16):
        key.mul_(5).add_(sym[d:d + n1])
    del sym

    def dense_ranks(key):
        skey, sa = torch.sort(key)
        new = torch.ones(n1, dtype=torch.int64, device=dev)
        new[0] = 0
        new[1:] = (skey[1:] != skey[:-1]).to(torch.int64)
        new = torch.cumsum(new, 0)
        rank = torch.empty(n1, dtype=torch.int64, device=dev)
        rank[sa] = new
        return rank, sa, int(new[-1].item())

    rank, sa, top = dense_ranks(key)
    k = 16
    while top < n1 - 1:
        nxt = torch.zeros(n1, dtype=torch.int64, device=dev)
        if k < n1:
            nxt[:n1 - k] = rank[k:] + 1                   # 0 = "past the end", below every real rank + 1
        key = rank * (n1 + 1) + nxt
        del nxt
        rank, sa, top = dense_ranks(key)
        k *= 2
    return sa


class FmiIndex:
    """count[5], sentinel_index, cp_occ[(ref_seq_len >> 6) + 1] of reference + reverse complement."""

    def __init__(self, ref_seq_len, count, sentinel_index, cp_occ):
        self.ref_seq_len, self.count, self.sentinel_index = int(ref_seq_len), [int(c) for c in count], int(sentinel_index)
        self.cp_occ = cp_occ              # numpy CP_OCC_DTYPE array (host) or a torch uint8 tensor of the same bytes (device)

    @property
    def n_cp(self):
        return (self.ref_seq_len >> 6) + 1

    def struct(self, cp_ptr):
        return FmiIndexStruct(self.ref_seq_len, (C.c_int64 * 5)(*self.count), self.sentinel_index, cp_ptr)

    def host(self):
        if isinstance(self.cp_occ, np.ndarray):
            return self
        return FmiIndex(self.ref_seq_len, self.count, self.sentinel_index,
                        self.cp_occ.cpu().numpy().view(CP_OCC_DTYPE).reshape(-1))

    def to(self, device):
        import torch
        if not isinstance(self.cp_occ, np.ndarray):
            return FmiIndex(self.ref_seq_len, self.count, self.sentinel_index, self.cp_occ.to(device))
        return FmiIndex(self.ref_seq_len, self.count, self.sentinel_index,
                        torch.from_numpy(self.cp_occ.view(np.uint8).reshape(-1)).to(device))


def build_index(ref, device=None):
    """ref: base codes 0..3 of the genome (one strand).  The index is over ref + reverse complement + sentinel, as
    bwa-mem2 builds it.  Returns an FmiIndex whose cp_occ lives on `device` (a torch uint8 tensor), or as a numpy
    array when device is None / cpu."""
    import torch
    dev = torch.device(device) if device is not None else torch.device("cpu")
    f = (ref if isinstance(ref, torch.Tensor) else torch.as_tensor(np.ascontiguousarray(ref, dtype=np.uint8))).to(dev)
    text = torch.cat([f, 3 - torch.flip(f, [0])])
    n = text.numel()
    n1 = n + 1
    sa = suffix_array(text, dev)
    ncp = (n1 >> 6) + 1
    bwt = torch.full((ncp * 64,), 4, dtype=torch.uint8, device=dev)
    prev = sa - 1
    sent = int(torch.nonzero(sa == 0)[0].item())
    prev[sent] = 0
    bwt[:n1] = text[prev]
    bwt[sent] = 4
    del prev, sa
    cp = torch.zeros((ncp, 8), dtype=torch.int64, device=dev)
    blk = bwt.view(ncp, 64)
    weights = (torch.ones(64, dtype=torch.int64, device=dev) << torch.arange(63, -1, -1, device=dev))   # bit 63 - j for symbol j
    count = [1]
    for b in range(4):
        hit = blk == b
        per = hit.sum(1, dtype=torch.int64)
        cp[:, b] = torch.cumsum(per, 0) - per
        cp[:, 4 + b] = (hit.to(torch.int64) * weights).sum(1)          # int64 wrap-around is the uint64 bit pattern
        count.append(count[-1] + int(per.sum().item()))
    raw = cp.view(torch.uint8).reshape(-1)
    idx = FmiIndex(n1, count, sent, raw)
    return idx.host() if dev.type == "cpu" else idx


def save_index(index, path):
    """The tables FMI_search::load_index fills, in the file format of the fmi driver (csrc/drivers/fmi_main.cpp):
    "GBXFMI01", int64 ref_seq_len, int64 count[5], int64 sentinel_index, then the CP_OCC records (bwa-mem2's layout)."""
    idx = index.host()
    with open(path, "wb") as f:
        f.write(b"GBXFMI01")
        f.write(np.array([idx.ref_seq_len] + idx.count + [idx.sentinel_index], dtype="<i8").tobytes())
        f.write(idx.cp_occ.view(np.uint8).tobytes())


def load_index(path):
    with open(path, "rb") as f:
        assert f.read(8) == b"GBXFMI01", "not an fmi index file"
        head = np.frombuffer(f.read(56), dtype="<i8")
        cp = np.frombuffer(f.read(), dtype=CP_OCC_DTYPE).copy()
    assert len(cp) == (int(head[0]) >> 6) + 1
    return FmiIndex(int(head[0]), head[1:6], int(head[6]), cp)


# ---- bwa-mem2's own index file, <prefix>.bwt.2bit.64, as FMI_search::load_index reads it (the call the reference driver
# makes, fmi.cpp:79-80).  Layout as published in bwa-mem2's src/FMI_search.cpp (tools/bwa-mem2 is an empty submodule here:
# UNPINNED, nothing of it can be compiled or run in this image):
#     int64   reference_seq_len                      2 x genome length + 1
#     int64   count[5]                               symbols smaller than base c; load_index adds 1 to each (the sentinel row)
#     CP_OCC  cp_occ[(reference_seq_len >> 6) + 1]   64 bytes each: int64 cp_count[4], uint64 one_hot_bwt_str[4]
#     int8    sa_ms_byte[n_sa]; uint32 sa_ls_word[n_sa]   suffix-array samples: n_sa = (reference_seq_len >> 3) + 1 with
#                                                    SA_COMPRESSION (SA_COMPX 3, v2.1 on), = reference_seq_len before
#     int64   sentinel_index
# The SMEM search reads neither array of SA samples; a reader takes their size from the file length.
def save_bwa_mem2_index(index, prefix, sa=None, sa_compx=3):
    """Writes <prefix>.bwt.2bit.64.  sa: the full suffix array (int64[ref_seq_len]) to sample from, or None - the samples are
    then written as zeros: such a file serves the seeding benchmark (and this repo's drivers), not `bwa-mem2 mem`."""
    idx = index.host()
    n = int(idx.ref_seq_len)
    n_sa = (n >> sa_compx) + 1 if sa_compx else n
    path = "%s.bwt.2bit.64" % prefix
    with open(path, "wb") as f:
        f.write(np.array([n] + [int(c) - 1 for c in idx.count], dtype="<i8").tobytes())
        f.write(idx.cp_occ.view(np.uint8).tobytes())
        if sa is None:
            f.write(np.zeros(n_sa, dtype=np.int8).tobytes())
            f.write(np.zeros(n_sa, dtype="<u4").tobytes())
        else:
            smp = np.asarray(sa, dtype=np.int64)[::(1 << sa_compx) if sa_compx else 1][:n_sa]
            smp = np.concatenate([smp, np.zeros(n_sa - len(smp), dtype=np.int64)])
            f.write((smp >> 32).astype(np.int8).tobytes())
            f.write((smp & 0xffffffff).astype("<u4").tobytes())
        f.write(np.array([idx.sentinel_index], dtype="<i8").tobytes())
    return path


def load_bwa_mem2_index(prefix):
    """FMI_search::load_index as far as the SMEM search needs it: reads <prefix>.bwt.2bit.64 (or the file itself when
    `prefix` already names one).  The SA samples are skipped, whichever of the two published sizes they have."""
    import os
    path = prefix if os.path.exists(prefix) and not os.path.exists("%s.bwt.2bit.64" % prefix) else "%s.bwt.2bit.64" % prefix
    size = os.path.getsize(path)
    with open(path, "rb") as f:
        head = np.frombuffer(f.read(48), dtype="<i8")
        n = int(head[0])
        ncp = (n >> 6) + 1
        cp = np.frombuffer(f.read(ncp * 64), dtype=CP_OCC_DTYPE).copy()
        rest = size - 48 - ncp * 64 - 8
        if len(cp) != ncp or rest < 0 or rest % 5 or rest // 5 not in (n, (n >> 3) + 1):
            raise ValueError("%s: not a bwa-mem2 .bwt.2bit.64 file (reference_seq_len %d, %d bytes)" % (path, n, size))
        f.seek(size - 8)
        sentinel = int(np.frombuffer(f.read(8), dtype="<i8")[0])
    return FmiIndex(n, [int(c) + 1 for c in head[1:6]], sentinel, cp)


def write_reads(path, reads, fastq=True, wrap=0):
    """FASTQ (four lines per read) or FASTA (sequence lines wrapped at `wrap` bases when > 0) of an FmiReadSet."""
    letters = np.frombuffer(b"ACGTN", dtype=np.uint8)
    with open(path, "wb") as f:
        for r in range(reads.n_reads):
            a = int(reads.read_off[r])
            seq = letters[np.minimum(reads.enc[a:a + int(reads.read_len[r])], 4)].tobytes()
            if fastq:
                f.write(b"@r%d\n" % r + seq + b"\n+\n" + b"I" * len(seq) + b"\n")
            else:
                f.write(b">r%d\n" % r)
                step = wrap if wrap > 0 else max(len(seq), 1)
                for k in range(0, max(len(seq), 1), step):
                    f.write(seq[k:k + step] + b"\n")


def smems_text(smem):
    """What the reference prints under PRINT_OUTPUT (fmi.cpp:312-343): "rid:" for every read up to the last one that has an
    SMEM, "[m,n+1]" per SMEM."""
    out, prev = [], -1
    for s in smem:
        rid = int(s["rid"])
        if rid != prev:
            out += ["%d:" % j for j in range(prev + 1, rid + 1)]
        prev = rid
        out.append("[%d,%d]" % (int(s["m"]), int(s["n"]) + 1))
    return out


class FmiReadSet:
    """Reads as base codes 0..3, 4 = ambiguous (fmi.cpp:113-124): read r = enc[read_off[r] ..+ read_len[r])."""

    def __init__(self, enc, read_off, read_len):
        self.enc = np.ascontiguousarray(enc, dtype=np.uint8)
        self.read_off = np.ascontiguousarray(read_off, dtype=np.int64)
        self.read_len = np.ascontiguousarray(read_len, dtype=np.int32)
        self.n_reads = len(self.read_len)

    @classmethod
    def fixed(cls, enc2d):
        """the reference's layout: every read padded to max_readlength (query_cum_len_ar[r] = r * max_readlength)."""
        enc2d = np.ascontiguousarray(enc2d, dtype=np.uint8)
        n, L = enc2d.shape
        return cls(enc2d.reshape(-1), np.arange(n, dtype=np.int64) * L, np.full(n, L, dtype=np.int32))

    @property
    def max_len(self):
        return int(self.read_len.max()) if self.n_reads else 0

    def take(self, lo, hi):
        a = int(self.read_off[lo]) if hi > lo else 0
        b = int(self.read_off[hi - 1] + self.read_len[hi - 1]) if hi > lo else 0
        return FmiReadSet(self.enc[a:b], self.read_off[lo:hi] - a, self.read_len[lo:hi])


def smem_host(index, reads, params=None, out_cap=None):
    """gbx_fmi_smem_host -> (SMEM_DTYPE array, smem_off int64[n_reads + 1])."""
    params = params or default_params()
    idx = index.host()
    cap = int(out_cap if out_cap is not None else max(64, 20 * reads.n_reads))     # the reference's quota: 20 per read (fmi.cpp:183)
    off = np.zeros(reads.n_reads + 1, dtype=np.int64)
    n_out = C.c_int64(0)
    st = idx.struct(idx.cp_occ.ctypes.data)
    while True:
        out = np.empty(cap, dtype=SMEM_DTYPE)
        rc = N.lib().gbx_fmi_smem_host(C.byref(st), C.byref(params), reads.n_reads, N.ptr(reads.enc), reads.enc.size,
                                       N.ptr(reads.read_off), N.ptr(reads.read_len), N.ptr(out), cap, N.ptr(off), C.byref(n_out))
        if rc == -1 and out_cap is None and n_out.value > cap:       # like the reference's "realloc" (fmi.cpp:207-216)
            cap = int(n_out.value)
            continue
        N.check(rc)
        return out[:n_out.value], off


class DeviceFmi:
    """Device-resident index (re-laid for the device once) + reads; run() = one gbx_fmi_smem_device call."""

    def __init__(self, index, reads, device, params=None, out_cap=None):
        import torch
        self.params = params or default_params()
        self.index = index.to(device)
        self.n_reads, self.max_len = reads.n_reads, reads.max_len
        L = N.lib()
        self.st = self.index.struct(self.index.cp_occ.data_ptr())
        self.dindex = torch.empty(L.gbx_fmi_index_bytes(self.index.ref_seq_len), dtype=torch.uint8, device=device)
        N.check(L.gbx_fmi_index_build(C.byref(self.st), self.dindex.data_ptr(), self.dindex.numel(), None))
        torch.cuda.synchronize()
        self.set_reads(reads, device, out_cap)

    def set_reads(self, reads, device, out_cap=None):
        import torch
        self.n_reads, self.max_len = reads.n_reads, reads.max_len
        self.enc = torch.from_numpy(reads.enc).to(device)
        self.read_off = torch.from_numpy(reads.read_off).to(device)
        self.read_len = torch.from_numpy(reads.read_len).to(device)
        self.out_cap = int(out_cap if out_cap is not None else max(64, 24 * reads.n_reads))
        self.out = torch.zeros(self.out_cap * SMEM_DTYPE.itemsize, dtype=torch.uint8, device=device)
        self.smem_off = torch.zeros(reads.n_reads + 1, dtype=torch.int64, device=device)
        self.n_out = torch.zeros(1, dtype=torch.int64, device=device)
        self.work_bytes = N.lib().gbx_fmi_workspace_bytes(reads.n_reads, self.max_len, self.params.min_seed_len)
        self.work = torch.empty(self.work_bytes, dtype=torch.uint8, device=device)

    @classmethod
    def from_tensors(cls, index, tensors, device, params=None):
        """tensors: (enc uint8, read_off int64, read_len int32) already on the device (a received shard)."""
        import torch
        self = cls.__new__(cls)
        self.params = params or default_params()
        self.index = index
        L = N.lib()
        self.st = index.struct(index.cp_occ.data_ptr())
        self.dindex = torch.empty(L.gbx_fmi_index_bytes(index.ref_seq_len), dtype=torch.uint8, device=device)
        N.check(L.gbx_fmi_index_build(C.byref(self.st), self.dindex.data_ptr(), self.dindex.numel(), None))
        torch.cuda.synchronize()
        self.enc, self.read_off, self.read_len = tensors
        self.n_reads = self.read_len.numel()
        self.max_len = int(self.read_len.max().item()) if self.n_reads else 0
        self.out_cap = max(64, 24 * self.n_reads)
        self.out = torch.zeros(self.out_cap * SMEM_DTYPE.itemsize, dtype=torch.uint8, device=device)
        self.smem_off = torch.zeros(self.n_reads + 1, dtype=torch.int64, device=device)
        self.n_out = torch.zeros(1, dtype=torch.int64, device=device)
        self.work_bytes = L.gbx_fmi_workspace_bytes(self.n_reads, self.max_len, self.params.min_seed_len)
        self.work = torch.empty(self.work_bytes, dtype=torch.uint8, device=device)
        return self

    def run(self, stream=None):
        N.check(N.lib().gbx_fmi_smem_device(C.byref(self.st), self.dindex.data_ptr(), C.byref(self.params), self.n_reads,
                                            self.max_len, self.enc.data_ptr(), self.read_off.data_ptr(), self.read_len.data_ptr(),
                                            self.out.data_ptr(), self.out_cap, self.smem_off.data_ptr(), self.n_out.data_ptr(),
                                            self.work.data_ptr(), self.work_bytes, stream))

    def overflow(self, stream=None):
        """0, or a lower bound of the SMEM count of a read that asked for more than its slot holds (gbx_fmi_overflow)."""
        v = C.c_int64(0)
        N.check(N.lib().gbx_fmi_overflow(self.work.data_ptr(), C.byref(v), stream))
        return v.value

    def results(self):
        n = int(self.n_out.item())
        if n > self.out_cap:
            raise RuntimeError("fmi: %d SMEMs do not fit the output capacity %d" % (n, self.out_cap))
        if self.overflow():
            raise RuntimeError("fmi: a read produced more SMEMs than its slot holds (gbx_fmi_overflow): use the host entry, which resizes")
        raw = self.out[:n * SMEM_DTYPE.itemsize].cpu().numpy()
        return raw.view(SMEM_DTYPE).copy(), self.smem_off.cpu().numpy()

    def extensions(self, stream=None):
        v = C.c_int64(0)
        N.check(N.lib().gbx_fmi_extensions(self.work.data_ptr(), C.byref(v), stream))
        return v.value
