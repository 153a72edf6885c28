import gzip
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def load_bsw_golden(name):
    from genomicsbench_amd import io as gio
    b = gio.read_bsw_pairs(os.path.join(GOLDEN, "bsw_%s.pairs.txt.gz" % name))
    g = np.loadtxt(gzip.open(os.path.join(GOLDEN, "bsw_%s.golden.txt.gz" % name), "rt"), dtype=np.int32, ndmin=2)
    return b, g[:, :6], g[:, 6:]


def load_chain_golden(name):
    from genomicsbench_amd import io as gio
    case = gio.read_chain_calls(os.path.join(GOLDEN, "chain_%s.in.gz" % name))
    g = np.loadtxt(gzip.open(os.path.join(GOLDEN, "chain_%s.golden.txt.gz" % name), "rt"), dtype=np.int32, ndmin=2)
    return case, g


def load_fmi_golden():
    """tests/golden/fmi_small.*: (genome codes, FmiReadSet, records as (rid, m, n, k, l, s) rows).  The records were written by
    the oracle (tools/bwa-mem2 is an empty submodule: nothing of the reference can be run) and frozen as a regression fixture."""
    from genomicsbench_amd.fmi import FmiReadSet
    code = {"A": 0, "C": 1, "G": 2, "T": 3}
    g = np.array([code[c] for c in gzip.open(os.path.join(GOLDEN, "fmi_small.genome.txt.gz"), "rt").read().strip()], dtype=np.uint8)
    lines = gzip.open(os.path.join(GOLDEN, "fmi_small.reads.fastq.gz"), "rt").read().splitlines()
    seqs = [np.array([code.get(c, 4) for c in lines[k]], dtype=np.uint8) for k in range(1, len(lines), 4)]
    off = np.concatenate([[0], np.cumsum([len(s) for s in seqs])[:-1]]).astype(np.int64)
    rs = FmiReadSet(np.concatenate(seqs), off, np.array([len(s) for s in seqs], dtype=np.int32))
    rows, rid = [], -1
    for ln in gzip.open(os.path.join(GOLDEN, "fmi_small.smems.txt.gz"), "rt").read().splitlines():
        if ln.endswith(":"):
            rid = int(ln[:-1])
        else:
            iv, k, l, s = ln.split()
            m, n1 = iv[1:-1].split(",")
            rows.append((rid, int(m), int(n1) - 1, int(k), int(l), int(s)))
    return g, rs, np.array(rows, dtype=np.int64)


def header_symbols():
    """Every function name declared in include/gbx.h."""
    txt = open(os.path.join(ROOT, "include", "gbx.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(gbx_[a-z0-9_]+)\s*\(", txt)))
