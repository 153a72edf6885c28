"""PCIe-inclusive time of the host-buffer entries on the 'large' sets (pageable numpy arrays in and out, Python wrapper's
own allocations included): python3 scripts/dbg_host_entries.py [chain] [phmm] [poa]"""
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from genomicsbench_amd import _native as N
N.check(N.lib().gbx_host_prepare())
which = sys.argv[1:] or ["chain", "phmm", "poa"]
def timed(tag, fn):
    ms = []
    for _ in range(3):
        t = time.perf_counter(); fn(); ms.append((time.perf_counter() - t) * 1e3)
    print("%s: first %.1f ms, best %.1f ms" % (tag, ms[0], min(ms)), flush=True)
if "chain" in which:
    from genomicsbench_amd.chain import chain_host
    from genomicsbench_amd.datagen import gen_chain
    c = gen_chain(10000, 2001)
    timed("chain large (10 000 calls, %d anchors) gbx_chain_host" % len(c[1]), lambda: chain_host(*c))
if "phmm" in which:
    from genomicsbench_amd.phmm import forward_host
    from genomicsbench_amd.datagen import gen_phmm
    b = gen_phmm(20000, 3001)
    timed("phmm large (20 000 batches, %d pairs) gbx_phmm_forward_host" % b.n_pairs, lambda: forward_host(b))
if "poa" in which:
    from genomicsbench_amd.poa import consensus_host, make_params
    from genomicsbench_amd.datagen import gen_poa
    w = gen_poa(6000, 4001); p = make_params()
    timed("poa large (6000 windows) gbx_poa_consensus_host", lambda: consensus_host(p, w))
if "abea" in which:
    from genomicsbench_amd.abea import PAIR_DTYPE
    from genomicsbench_amd.datagen import gen_abea
    rs = gen_abea(int(os.environ.get("ABEA_READS", "4000")), 5001)
    ev = rs.events_struct()
    out = np.zeros(2 * max(int(rs.event_off[-1]), 1), dtype=PAIR_DTYPE)
    out["ref_pos"][:] = -1
    npairs = np.zeros(max(rs.n_reads, 1), dtype=np.int32)
    timed("abea (%d reads, %d events) gbx_abea_align_host" % (rs.n_reads, int(rs.event_off[-1])), lambda: N.check(N.lib().gbx_abea_align_host(
        rs.n_reads, N.ptr(rs.seq_off), N.ptr(rs.seq_len), N.ptr(rs.seq_arena), rs.seq_arena.size, N.ptr(rs.event_off), N.ptr(ev), N.ptr(rs.model),
        N.ptr(rs.scale), N.ptr(rs.shift), N.ptr(out), N.ptr(npairs))))
