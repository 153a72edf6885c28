"""Development aid: how full the lock-step rows of bsw_lane_kernel are on the 'large' job - per launch, the cells the
lanes computed against 64 x the widest window of every row (what the wavefront paid for), and the rows the lanes ran
against 64 x the rows the wavefront ran.  Needs a library built with -DGBX_BSW_LANE_STATS:
  hipcc ... -DGBX_BSW_LANE_STATS -c bsw_kernels.hip -o /tmp/bsw_ls.o; link with the other objects into genomicsbench_amd/libgbx_ls.so
usage: GBX_LIB=$PWD/genomicsbench_amd/libgbx_ls.so python scripts/dbg_bsw_lanes.py [n_pairs]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from genomicsbench_amd import _native as N
from genomicsbench_amd.bsw import DeviceBswBatch, make_params
from genomicsbench_amd.datagen import gen_bsw

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
b = gen_bsw(n, 1002)
d = DeviceBswBatch(b, torch.device("cuda:0"))
p = make_params()
d.run(p); torch.cuda.synchronize()
out = (C.c_ulonglong * 64)()
N.lib().gbx_debug_bsw_lane_stats(out, 1)
d.run(p); torch.cuda.synchronize()
N.lib().gbx_debug_bsw_lane_stats(out, 0)
s = np.array(list(out), dtype=np.float64).reshape(16, 4)
names = ["c47", "c79", "c99", "c135", "c159", "w39", "w79", "w103", "w127", "w159"]
tc = tw = 0
for k, nm in enumerate(names):
    c, w, r, wr = s[k]
    if wr:
        print("%-5s lane cells %.3e  paid %.3e  (%.1f %% full)   lane rows %.3e paid %.3e (%.1f %%)   mean window %.1f, paid window %.1f"
              % (nm, c, w, 100 * c / w, r, wr, 100 * r / wr, c / r, w / wr))
        tc += c; tw += w
print("all: %.1f %% of the paid cells are computed cells; nominal cells %.3e, computed %.3e" % (100 * tc / tw, float((b.len1.astype(np.int64) * b.len2).sum()), tc))
