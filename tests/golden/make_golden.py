"""Generates the committed golden fixtures from the COMPILED REFERENCE (oracle/_ref/*.so).

Run in the build container only (needs /root/reference):  python tests/golden/make_golden.py
Fixtures are data: seeded synthetic inputs in the reference's own input formats and the
outputs the reference's code produced for them.

bsw_<name>.pairs.txt.gz   input, reference format (main_banded.cpp:131-141)
bsw_<name>.golden.txt.gz  per pair: 6 outputs of scalarBandedSWA, then 6 of getScores16 (AVX2, -b 512)
chain_<name>.in.gz        input, reference format (host_data_io.cpp:13-51)
chain_<name>.golden.txt.gz  per anchor: score parent target peak (return_t vectors of host_chain_kernel)
fmi_small.genome.txt.gz     one line of ACGT: the genome the index is built from (both strands, as bwa-mem2 does)
fmi_small.reads.fastq.gz    the reads (FASTQ, as the reference driver takes them, fmi.cpp:62-70)
fmi_small.smems.txt.gz      the SMEMs in the reference's PRINT_OUTPUT format (fmi.cpp:312-343), k l s appended to every record.
                            NOT from the reference (tools/bwa-mem2 is an empty submodule): written by oracle/fmi_oracle.c and frozen
                            here as a regression fixture; tests/test_fmi_cpu.py checks the same records against brute force
"""
import gzip
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from cases import adversarial_bsw, edge_bsw, chain_cases  # noqa: E402
from genomicsbench_amd import io as gio  # noqa: E402
from genomicsbench_amd.bsw import BswBatch, make_params  # noqa: E402
from genomicsbench_amd.datagen import gen_bsw, gen_chain  # noqa: E402
from oracle import oracle_py as O  # noqa: E402


def bsw_fixture(name, batch):
    keep = np.nonzero((batch.len1 > 0) & (batch.len2 > 0))[0]      # the text format cannot hold empty lines
    b = BswBatch(batch.ref, batch.qer, batch.idr[keep], batch.idq[keep], batch.len1[keep], batch.len2[keep],
                 batch.h0[keep])
    gio.write_bsw_pairs(os.path.join(HERE, "bsw_%s.pairs.txt.gz" % name), b)
    b = gio.read_bsw_pairs(os.path.join(HERE, "bsw_%s.pairs.txt.gz" % name))
    p = make_params()
    sc = O.bsw_ref_scalar(p, b)
    av = O.bsw_ref_avx2(p, b, 512)
    with gzip.open(os.path.join(HERE, "bsw_%s.golden.txt.gz" % name), "wt") as f:
        for k in range(b.n):
            f.write(" ".join(map(str, list(sc[k]) + list(av[k]))) + "\n")
    print(name, b.n, "pairs; scalar/AVX2 row mismatches:", int((sc != av).any(1).sum()))


def chain_fixture(name, off, ax, ay, hdr):
    gio.write_chain_calls(os.path.join(HERE, "chain_%s.in.gz" % name), off, ax, ay, hdr)
    off, ax, ay, hdr = gio.read_chain_calls(os.path.join(HERE, "chain_%s.in.gz" % name))
    s, p, t, k = O.chain_ref(off, ax, ay, hdr, 1)
    with gzip.open(os.path.join(HERE, "chain_%s.golden.txt.gz" % name), "wt") as f:
        for i in range(len(s)):
            f.write("%d %d %d %d\n" % (s[i], p[i], t[i], k[i]))
    print(name, len(off) - 1, "calls", len(s), "anchors")


def fmi_fixture():
    from genomicsbench_amd import fmi as FM
    from genomicsbench_amd.datagen import gen_fmi_genome, gen_fmi_reads
    g = gen_fmi_genome(4000, 123)
    g[1000:1300] = g[200:500]                                   # a repeat with two diverged copies: SMEMs with several hits
    g[2500:2800] = 3 - g[200:500][::-1]
    g[1100] = (g[1100] + 1) % 4
    rs = gen_fmi_reads(g, 60, 124, read_len=101)
    rs = FM.FmiReadSet(rs.enc, rs.read_off, np.maximum(20, rs.read_len - (np.arange(60) % 4).astype(np.int32) * 13))
    with gzip.open(os.path.join(HERE, "fmi_small.genome.txt.gz"), "wt") as f:
        f.write("".join("ACGT"[b] for b in g) + "\n")
    FM.write_reads(os.path.join(HERE, "fmi_small.reads.fastq"), rs, fastq=True)
    with open(os.path.join(HERE, "fmi_small.reads.fastq"), "rb") as fi, gzip.open(os.path.join(HERE, "fmi_small.reads.fastq.gz"), "wb") as fo:
        fo.write(fi.read())
    os.remove(os.path.join(HERE, "fmi_small.reads.fastq"))
    out, off = O.fmi_oracle(FM.build_index(g), rs, FM.default_params(19))
    with gzip.open(os.path.join(HERE, "fmi_small.smems.txt.gz"), "wt") as f:
        prev = -1
        for s_ in out:
            rid = int(s_["rid"])
            if rid != prev:
                for j in range(prev + 1, rid + 1):
                    f.write("%d:\n" % j)
            prev = rid
            f.write("[%d,%d] %d %d %d\n" % (s_["m"], s_["n"] + 1, s_["k"], s_["l"], s_["s"]))
    print("fmi_small", rs.n_reads, "reads", len(out), "SMEMs")


if __name__ == "__main__":
    assert O.ref_lib("bsw") is not None and O.ref_lib("chain") is not None, "run oracle/build_ref.sh first"
    only = sys.argv[1:]                       # e.g. `make_golden.py chain_cuts chain_realistic`: just these fixtures
    for name, make in (("realistic", lambda: gen_bsw(3000, 1001)), ("adversarial", lambda: adversarial_bsw(3000, 4242)),
                       ("edge", edge_bsw)):
        if not only or "bsw_" + name in only:
            bsw_fixture(name, make())
    for name, case in chain_cases().items():
        if not only or "chain_" + name in only:
            chain_fixture(name, *case)
    if not only or "fmi_small" in only:
        fmi_fixture()
