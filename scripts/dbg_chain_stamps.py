"""Where an anchor's cycles go in chain_kernel (longest call of 'large'): needs a libgbx built with
`make EXTRA=-DGBX_CHAIN_STAMPS`.  Usage (GPU box): python3 scripts/dbg_chain_stamps.py [n_calls]"""
import ctypes as C, sys
sys.path.insert(0, ".")
import numpy as np, torch
from genomicsbench_amd import _native as N
from genomicsbench_amd.chain import DeviceChainBatch
from genomicsbench_amd.datagen import gen_chain
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
case = gen_chain(n, 2001)
d = DeviceChainBatch(*case, torch.device("cuda:0"))
t = N.StreamTimer()
for _ in range(1):
    t.start(None); d.run(None); t.stop(None)
    print("chain ms", t.elapsed_ms())
out = (C.c_ulonglong * 16)()
N.lib().gbx_debug_chain_stamps(out)
v = list(out)
names = ["st advance", "ring/global loads (wait)", "phase1 math", "phase2 marks", "phase3 scans", "ballots/readlane", "phase4 stores", "loop overhead/exit", "outputs+ring update"]
tot = sum(v[:9]); na = v[13]; nch = v[12]
print("chunks past the ring (global loads)", v[14])
print("of the chunks: quiet (nobody improves, nobody is marked) %.1f %%, with a prefix-max scan (somebody improves) %.1f %%, ending in the max_skip break %.1f %% (%.2f breaks per anchor)"
      % (100.0 * v[9] / max(v[12], 1), 100.0 * v[10] / max(v[12], 1), 100.0 * v[11] / max(v[12], 1), v[11] / max(v[13], 1)))
print("anchors", na, "chunks", nch, "chunks/anchor %.2f" % (nch / max(na, 1)), "cycles/anchor %.0f" % (tot / max(na, 1)))
for k, nm in enumerate(names):
    per = v[k] / max(nch if 1 <= k <= 6 else na, 1)
    print("%-28s %6.1f %%  %8.0f cycles per %s" % (nm, 100.0 * v[k] / max(tot, 1), per, "chunk" if 1 <= k <= 6 else "anchor"))
