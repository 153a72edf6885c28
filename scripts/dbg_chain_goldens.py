import sys, os
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
from util import load_chain_golden
from genomicsbench_amd.chain import chain_host
for name in ("mixed", "dense_maxiter", "multiseg"):
    case, g = load_chain_golden(name)
    got = chain_host(*case)
    off = case[0]
    for f, nm in enumerate(("score", "parent", "target", "peak")):
        bad = np.nonzero(got[f] != g[:, f])[0]
        if len(bad):
            calls = np.searchsorted(off, bad, side="right") - 1
            print(name, nm, len(bad), "first", bad[0], "in call", calls[0], "call len", off[calls[0]+1]-off[calls[0]], "local", bad[0]-off[calls[0]], "got", got[f][bad[0]], "want", g[bad[0], f], "hdr", case[3][calls[0]], "bad calls", sorted(set(calls.tolist()))[:10])
    print(name, "done")
