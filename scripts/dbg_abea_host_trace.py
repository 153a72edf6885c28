"""Timeline of gbx_abea_align_host on abea 'large' (GBX_HOST_TRACE): python3 scripts/dbg_abea_host_trace.py"""
import os, sys, time
sys.path.insert(0, ".")
os.environ["GBX_HOST_TRACE"] = "1"
import numpy as np
from genomicsbench_amd import _native as N
from genomicsbench_amd.abea import PAIR_DTYPE
from genomicsbench_amd.datagen import gen_abea
rs = gen_abea(10000, 5001)
N.check(N.lib().gbx_host_prepare())
ev = rs.events_struct()
out = np.zeros(2 * int(rs.event_off[-1]), dtype=PAIR_DTYPE); out["ref_pos"][:] = 1          # pages touched
n_pairs = np.zeros(rs.n_reads, dtype=np.int32)
for k in range(3):
    t = time.perf_counter()
    N.check(N.lib().gbx_abea_align_host(rs.n_reads, N.ptr(rs.seq_off), N.ptr(rs.seq_len), N.ptr(rs.seq_arena), rs.seq_arena.size,
                                        N.ptr(rs.event_off), N.ptr(ev), N.ptr(rs.model), N.ptr(rs.scale), N.ptr(rs.shift), N.ptr(out), N.ptr(n_pairs)))
    print("call %d: %.1f ms (in %.2f GB of 24-byte event records, out %.2f GB)" % (k, (time.perf_counter() - t) * 1e3, ev.nbytes / 1e9, out.nbytes / 1e9), flush=True)
