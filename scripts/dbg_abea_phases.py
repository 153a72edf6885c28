"""Development aid: where the abea kernel's wavefronts spend their time (s_memtime ticks summed over wavefronts:
per-k-mer prologue / band loop / traceback), for the first N reads of the synthetic 'large' set.
usage: python scripts/dbg_abea_phases.py [n_reads ...]"""
import ctypes as C
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from genomicsbench_amd import _native as N
from genomicsbench_amd.abea import DeviceAbeaReadSet
from genomicsbench_amd.datagen import gen_abea

sizes = [int(a) for a in sys.argv[1:]] or [1, 64, 4096]
rs = gen_abea(max(sizes), 7)
lib = N.lib()
for n in sizes:
    d = DeviceAbeaReadSet(rs.take(0, n), "cuda:0")
    d.run(); torch.cuda.synchronize()
    t = time.perf_counter(); d.run(); torch.cuda.synchronize(); ms = (time.perf_counter() - t) * 1e3
    out = (C.c_ulonglong * 5)()
    lib.gbx_debug_abea_ticks(C.c_void_p(d.work.data_ptr()), out)
    tot = sum(out[:4]) or 1
    print("reads %5d bands %9d steps %9d  %.2f ms  ticks: prologue %.0f%%  bands %.0f%% (%.1f/band)  walk %.0f%% (%.1f/step)  pass2 %.0f%% (%.1f/step)" % (
        n, d.n_bands_total, out[4], ms, 100 * out[0] / tot, 100 * out[1] / tot, out[1] / d.n_bands_total,
        100 * out[2] / tot, out[2] / max(out[4], 1), 100 * out[3] / tot, out[3] / max(out[4], 1)))
