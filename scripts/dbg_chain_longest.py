"""Development aid: the longest call of chain 'large' alone, the whole job and the job without that call, for the
library named by GBX_LIB (e.g. one built with -DGBX_CHAIN_RING_LIVE=2048).
usage: GBX_LIB=... python scripts/dbg_chain_longest.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from genomicsbench_amd.chain import DeviceChainBatch
from genomicsbench_amd.datagen import gen_chain

off, ax, ay, hdr = gen_chain(10000, 2001)
n = np.diff(off)
k = int(np.argmax(n))
def sub(idx):
    o = np.concatenate([[0], np.cumsum(n[idx])]).astype(np.int64)
    sel = np.concatenate([np.arange(off[i], off[i + 1]) for i in idx])
    return o, ax[sel], ay[sel], hdr[idx]
def run(c, reps=3):
    d = DeviceChainBatch(*c, torch.device("cuda:0"))
    d.run(None); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t = time.perf_counter(); d.run(None); torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t) * 1e3)
    return best
print(os.path.basename(os.environ.get("GBX_LIB", "default")), "longest call %d anchors alone: %.2f ms; whole job: %.2f ms; job without it: %.2f ms" % (
    n[k], run(sub([k])), run((off, ax, ay, hdr)), run(sub([i for i in range(len(n)) if i != k]))))
