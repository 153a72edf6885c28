"""Development aid: where poa_kernel's wavefronts spend their clocks (DP / traceback / add_alignment incl. the
topological sort / consensus), summed over wavefronts.  Needs a library built with -DGBX_POA_PHASE_STATS:
  hipcc ... -DGBX_POA_PHASE_STATS -c poa_kernels.hip -o /tmp/poa_ps.o; link into genomicsbench_amd/libgbx_ps.so
usage: GBX_LIB=$PWD/genomicsbench_amd/libgbx_ps.so python scripts/dbg_poa_phases.py [n_windows]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from genomicsbench_amd.poa import DevicePoaWindowSet, make_params
from genomicsbench_amd.datagen import gen_poa

n = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
ws = gen_poa(n, 4001)
d = DevicePoaWindowSet(ws, torch.device("cuda:0"))
p = make_params()
d.run(p); torch.cuda.synchronize()
t = time.perf_counter(); d.run(p); torch.cuda.synchronize(); ms = (time.perf_counter() - t) * 1e3
import ctypes as C
from genomicsbench_amd import _native as N
N.lib().gbx_debug_poa_counter_offset.restype = C.c_size_t
off = N.lib().gbx_debug_poa_counter_offset(C.byref(d.plan))
c = d.work[off:off + 256].cpu().numpy().view(np.uint64)
print("plan: slots %d, long windows %d on %d slots, workspace %.2f GB" % (d.plan.n_slots, d.plan.n_long_windows, d.plan.long_slots, d.work_bytes / 1e9))
dp, tb, add, cons, topo, topo_n, vis, blk, dfs = [int(x) for x in c[1:10]]
# cells 1-4, 12, 13 are the last run's; the sort's and add_alignment's counters are device globals that keep counting: two runs
RUNS = 2
topo, topo_n, vis, blk, dfs = (v // RUNS for v in (topo, topo_n, vis, blk, dfs))
tot = dp + tb + add + cons
print("%d windows %.1f ms; wave clocks: DP %.1f%%  traceback %.1f%%  add_alignment %.1f%% (the topological sort alone: %.1f%% of all wave clocks, its DFS part %.1f%%)  consensus %.1f%%" % (
    n, ms, 100 * dp / tot, 100 * tb / tot, 100 * add / tot, 100 * topo / tot, 100 * dfs / tot, 100 * cons / tot))
roots, triv = int(c[10]) // RUNS, int(c[11]) // RUNS
print("sort roots per node %.3f, single-visit roots per node %.3f" % (roots / max(topo_n, 1), triv / max(topo_n, 1)))
print("sort: %.0f clocks per node, %.2f visits per node, %.3f block loads per node" % (topo / max(topo_n, 1), vis / max(topo_n, 1), blk / max(topo_n, 1)))
if int(c[21]):         # a -DGBX_POA_TOPO_CHECK build: every incremental sort was followed by a full one in global memory
    print("incremental sorts %d, orders that differ from the full sort: %d; blocks walked %.1f of %.1f per sort" % (
        int(c[21]), int(c[20]), int(c[22]) / int(c[21]), int(c[23]) / int(c[21])))
rows, steps = int(c[12]), int(c[13])
if rows:
    print("DP: %d rows, %.0f wave clocks per row; traceback: %d steps, %.0f clocks per step; add_alignment without the sort: %.0f clocks per path element"
          % (rows, dp / rows, steps, tb / max(steps, 1), (add - topo) / max(steps, 1)))
ser, uns, par, hd = (int(c[k]) // RUNS for k in (24, 25, 26, 27))
if uns:
    print("add_alignment: letter codes + fresh chains %.1f%% of all wave clocks, nodes of the elements %.1f%%, their edges %.1f%% (%d elements with a base, %.0f clocks each for both)"
          % (100 * hd / tot, 100 * par / tot, 100 * ser / tot, uns, (par + ser) / uns))
fast, slow, fill = (int(c[k]) // RUNS for k in (28, 29, 30))
if fast + slow:
    print("traceback: %.1f%% of the steps served from the 8x8 block, %.1f%% through the general path; %.2f block fills per step" % (100 * fast / (fast + slow), 100 * slow / (fast + slow), fill / (fast + slow)))
