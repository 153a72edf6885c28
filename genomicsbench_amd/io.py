"""Readers / writers for the reference benchmarks' on-disk formats.

bsw   : 3 text lines per pair (h0, target digits 0-4, query digits 0-4)   R/benchmarks/bsw/main_banded.cpp:131-185
phmm  : per batch `num_reads num_haps`, reads `bases q i d c` (Phred+33), haplotypes        R/benchmarks/phmm/PairHMMUnitTest.cpp:95-140
poa   : FASTA-like, 2 lines per record; a header whose 2nd character is '0' opens a new window        R/benchmarks/poa/msa_spoa_omp.cpp:82-116
chain : `n avg_qspan max_dist_x max_dist_y bw n_segs`, n lines `x y`, `EOR`  R/benchmarks/chain/src/host_data_io.cpp:13-60
"""
import gzip
import io as _io

import numpy as np

from . import _native as N


def _open(path, mode):
    if str(path).endswith(".gz"):
        return gzip.open(path, mode + "t")
    return open(path, mode)


# ----------------------------------------------------------------------------- bsw
def write_bsw_pairs(path, batch):
    with _open(path, "w") as f:
        for k in range(batch.n):
            t = batch.ref[batch.idr[k]:batch.idr[k] + batch.len1[k]]
            q = batch.qer[batch.idq[k]:batch.idq[k] + batch.len2[k]]
            f.write("%d\n%s\n%s\n" % (batch.h0[k], (t + 48).tobytes().decode(), (q + 48).tobytes().decode()))


def read_bsw_pairs(path):
    """Same acceptance as loadPairs: 3 lines per pair, digits minus 48, both lengths > 0."""
    from .bsw import BswBatch
    with _open(path, "r") as f:
        lines = f.read().split("\n")
    n = len(lines) // 3 if lines and lines[-1] != "" else (len(lines) - 1) // 3
    h0 = np.zeros(n, dtype=np.int32)
    ts, qs = [], []
    for k in range(n):
        h0[k] = int(lines[3 * k].strip() or 0)
        ts.append(np.frombuffer(lines[3 * k + 1].encode(), dtype=np.uint8) - 48)
        qs.append(np.frombuffer(lines[3 * k + 2].encode(), dtype=np.uint8) - 48)
    return BswBatch.from_sequences(ts, qs, h0)


# --------------------------------------------------------------------------- chain
def write_chain_calls(path, off, ax, ay, hdr):
    with _open(path, "w") as f:
        for c in range(len(off) - 1):
            h = hdr[c]
            f.write("%d %.6g %d %d %d %d\n" % (off[c + 1] - off[c], float(h["avg_qspan"]), h["max_dist_x"],
                                               h["max_dist_y"], h["bw"], h["n_segs"]))
            for i in range(int(off[c]), int(off[c + 1])):
                f.write("%d %d\n" % (int(ax[i]), int(ay[i])))
            f.write("EOR\n")


def read_chain_calls(path):
    with _open(path, "r") as f:
        toks = f.read().split()
    counts, hdrs, xs, ys = [], [], [], []
    p = 0
    while p + 6 <= len(toks):
        try:
            n = int(toks[p])
        except ValueError:
            break
        hdrs.append((float(np.float32(toks[p + 1])), int(toks[p + 2]), int(toks[p + 3]), int(toks[p + 4]),
                     int(toks[p + 5])))
        p += 6
        a = np.array(toks[p:p + 2 * n], dtype=np.uint64)
        xs.append(a[0::2])
        ys.append(a[1::2])
        p += 2 * n
        assert toks[p] == "EOR", "missing EOR after call %d" % len(counts)
        p += 1
        counts.append(n)
    off = np.zeros(len(counts) + 1, dtype=np.int64)
    np.cumsum(np.array(counts, dtype=np.int64), out=off[1:])
    hdr = np.array(hdrs, dtype=N.CHAIN_CALL_DTYPE) if hdrs else np.zeros(0, dtype=N.CHAIN_CALL_DTYPE)
    ax = np.concatenate(xs) if xs else np.zeros(0, np.uint64)
    ay = np.concatenate(ys) if ys else np.zeros(0, np.uint64)
    return off, np.ascontiguousarray(ax), np.ascontiguousarray(ay), hdr


def write_chain_returns(path_or_file, off, score, parent):
    """print_return: `n`, n lines `score\\tparent`, `EOR`  (host_data_io.cpp:53-60)."""
    f = path_or_file if hasattr(path_or_file, "write") else _open(path_or_file, "w")
    for c in range(len(off) - 1):
        f.write("%d\n" % (off[c + 1] - off[c]))
        for i in range(int(off[c]), int(off[c + 1])):
            f.write("%d\t%d\n" % (score[i], parent[i]))
        f.write("EOR\n")
    if f is not path_or_file:
        f.close()


# ---------------------------------------------------------------------------- phmm
def write_phmm_batches(path, bs):
    """Quality tracks are stored normalised (minus 33, q >= 6); the file carries Phred+33 ASCII."""
    asc = lambda a: (a + 33).astype(np.uint8).tobytes().decode()
    with _open(path, "w") as f:
        r = h = 0
        for b in range(len(bs.n_reads)):
            f.write("%d %d\n" % (bs.n_reads[b], bs.n_haps[b]))
            for _ in range(bs.n_reads[b]):
                o, n = int(bs.read_off[r]), int(bs.read_len[r])
                f.write("%s %s %s %s %s\n" % (bs.rs[o:o + n].tobytes().decode(), asc(bs.q[o:o + n]),
                                              asc(bs.qi[o:o + n]), asc(bs.qd[o:o + n]), asc(bs.qc[o:o + n])))
                r += 1
            for _ in range(bs.n_haps[b]):
                o, n = int(bs.hap_off[h]), int(bs.hap_len[h])
                f.write("%s\n" % bs.hap[o:o + n].tobytes().decode())
                h += 1


def read_phmm_batches(path):
    """read_testfile / read_batch: whitespace-separated tokens; normalize(): max(min, ch-33), q min 6."""
    from .phmm import PhmmBatchSet
    with _open(path, "r") as f:
        toks = f.read().split()
    p = 0
    nr, nh, rl, hl = [], [], [], []
    rs, q, qi, qd, qc, hp = [], [], [], [], [], []
    norm = lambda s, lo: np.maximum(np.frombuffer(s.encode(), dtype=np.uint8).astype(np.int32) - 33, lo).astype(np.uint8)
    while p + 2 <= len(toks):
        a, b = int(toks[p]), int(toks[p + 1])
        p += 2
        nr.append(a)
        nh.append(b)
        for _ in range(a):
            rs.append(np.frombuffer(toks[p].encode(), dtype=np.uint8))
            q.append(norm(toks[p + 1], 6)); qi.append(norm(toks[p + 2], 0))
            qd.append(norm(toks[p + 3], 0)); qc.append(norm(toks[p + 4], 0))
            rl.append(len(toks[p]))
            p += 5
        for _ in range(b):
            hp.append(np.frombuffer(toks[p].encode(), dtype=np.uint8))
            hl.append(len(toks[p]))
            p += 1
    cat = lambda xs: np.concatenate(xs + [np.zeros(8, np.uint8)]) if xs else np.zeros(8, np.uint8)
    roff = np.concatenate([[0], np.cumsum(rl)])[:-1] if rl else np.zeros(0, np.int64)
    hoff = np.concatenate([[0], np.cumsum(hl)])[:-1] if hl else np.zeros(0, np.int64)
    return PhmmBatchSet(nr, nh, roff, rl, cat(rs), cat(q), cat(qi), cat(qd), cat(qc), hoff, hl, cat(hp))


# ----------------------------------------------------------------------------- poa
def write_poa_windows(path, ws):
    with _open(path, "w") as f:
        for w in range(ws.n_windows):
            for k, s in enumerate(ws.window(w)):
                f.write(">%d\n%s\n" % (k, s))


def read_poa_windows(path):
    """readFile(): strictly 2 lines per record; header[1] == '0' starts a new window."""
    from .poa import PoaWindowSet
    with _open(path, "r") as f:
        lines = f.read().split("\n")
    if lines and lines[-1] == "":
        lines.pop()
    wins = []
    for k in range(0, len(lines) - 1, 2):
        if len(lines[k]) > 1 and lines[k][1] == "0":
            wins.append([])
        if wins:
            wins[-1].append(lines[k + 1])
    return PoaWindowSet.from_lists(wins)
