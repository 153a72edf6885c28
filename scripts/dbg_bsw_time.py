"""Development aid: device-resident time of the bsw 'large' job (2 M pairs), median of 15 runs; for A/B runs of tuning
environment variables (GBX_BSW_LANE_CLASSES, GBX_LIB, ...) on one box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from genomicsbench_amd.bsw import DeviceBswBatch, make_params
from genomicsbench_amd.datagen import gen_bsw
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
b = gen_bsw(n, 1002)
d = DeviceBswBatch(b, torch.device("cuda:0"))
p = make_params()
for _ in range(3):
    d.run(p)
torch.cuda.synchronize()
ts = []
for _ in range(15):
    t = time.perf_counter(); d.run(p); torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
cells = float((b.len1.astype(np.int64) * b.len2).sum())
ms = float(np.median(ts))
print("%.3f ms median (min %.3f)  %.0f GCUPS   checksum %d" % (ms, min(ts), cells / ms / 1e6, int(d.results().astype(np.int64).sum())))
