"""Development aid: what ONE poa window costs when it has a SIMD to itself (the state of every window of BASELINE config 4's
8-GPU leg: 750 windows on 1024 SIMDs).  Runs the first n windows of 'large' for n in a sweep and prints the step time; with
a -DGBX_POA_PHASE_STATS library (GBX_LIB=build_tmp/libgbx_ps.so) also where the wave clocks go.
usage: python scripts/dbg_poa_lone.py [n ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np, torch
from genomicsbench_amd import _native as N
from genomicsbench_amd.poa import DevicePoaWindowSet, make_params
from genomicsbench_amd.datagen import gen_poa

sizes = [int(a) for a in sys.argv[1:]] or [64, 256, 750, 1024, 1500, 3072, 6000]
full = gen_poa(6000, 4001)
p = make_params()
mx = np.maximum.reduceat(full.seq_len, full.win_first_seq[:-1])
print("windows with a sequence over 512 bases: %d of %d; longest sequence %d" % (int((mx > 512).sum()), full.n_windows, int(full.seq_len.max())))
for n in sizes:
    ws = full.take(0, n)
    d = DevicePoaWindowSet(ws, torch.device("cuda:0"))
    d.run(p); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t = time.perf_counter(); d.run(p); torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
    cells = d.cells()
    line = "%5d windows: %.1f ms (best of 3), %.1f GCUPS, long windows %d" % (n, min(ts), cells / min(ts) / 1e6, d.plan.n_long_windows)
    if hasattr(N.lib(), "gbx_debug_poa_counter_offset"):
        N.lib().gbx_debug_poa_counter_offset.restype = C.c_size_t
        off = N.lib().gbx_debug_poa_counter_offset(C.byref(d.plan))
        c = d.work[off:off + 256].cpu().numpy().view(np.uint64)
        dp, tb, add, cons = [int(x) for x in c[1:5]]
        tot = dp + tb + add + cons
        if tot:
            line += " | wave clocks: DP %.1f%% traceback %.1f%% add %.1f%% consensus %.1f%%; %.0f clocks per DP row, %.0f per traceback step" % (
                100 * dp / tot, 100 * tb / tot, 100 * add / tot, 100 * cons / tot, dp / max(int(c[12]), 1), tb / max(int(c[13]), 1))
            if int(c[28]) + int(c[29]):       # traceback steps served from the register block / by the general path, block refills
                line += "; traceback steps: %.1f%% from the block, %.1f%% general, a refill per %.1f block steps" % (
                    100 * int(c[28]) / (int(c[28]) + int(c[29])), 100 * int(c[29]) / (int(c[28]) + int(c[29])), int(c[28]) / max(int(c[30]), 1))
    print(line, flush=True)
    del d
    torch.cuda.empty_cache()
