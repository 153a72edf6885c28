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


def header_symbols():
    """Every function name declared in include/gbx.h."""
    txt = open(os.path.join(ROOT, "include", "gbx.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(gbx_[a-z0-9_]+)\s*\(", txt)))
