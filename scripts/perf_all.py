"""Quick per-kernel timing on the GPU (device-resident inputs, HIP-event stage timers).  Debug aid."""
import sys, time, json
sys.path.insert(0, ".")
import numpy as np
import torch
from genomicsbench_amd import _native as N
from genomicsbench_amd.datagen import gen_bsw, gen_chain, gen_phmm, gen_poa
from genomicsbench_amd import bsw as B, chain as CH, phmm as PH, poa as PO

dev = torch.device("cuda:0")
s = torch.cuda.current_stream().cuda_stream
which = sys.argv[1:] or ["chain", "phmm", "poa"]

def timeit(run, reps=3):
    run(); torch.cuda.synchronize()
    N.profile_begin(); t = time.perf_counter()
    for _ in range(reps): run()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / reps
    st = N.profile_end()
    return dt, {k: round(v[0] / v[1], 3) for k, v in st.items()}

if "chain" in which:
    n = int(sys.argv[sys.argv.index("--chain-calls") + 1]) if "--chain-calls" in sys.argv else 2000
    case = gen_chain(n, 2001)
    d = CH.DeviceChainBatch(*case, dev)
    dt, st = timeit(lambda: d.run(s))
    print("chain calls", n, "anchors", d.n_anchors, "ms", round(dt * 1e3, 2), st, "Manchors/s", round(d.n_anchors / dt / 1e6, 1), flush=True)
if "phmm" in which:
    bs = gen_phmm(2000, 3001)
    d = PH.DevicePhmmBatchSet(bs, dev)
    dt, st = timeit(lambda: d.run(s))
    print("phmm pairs", bs.n_pairs, "cells", bs.cells, "ms", round(dt * 1e3, 2), st, "GCUPS", round(bs.cells / dt / 1e9, 1), flush=True)
if "poa" in which:
    ws = gen_poa(1024, 4001)
    p = PO.make_params()
    d = PO.DevicePoaWindowSet(ws, dev)
    print("poa plan", d.plan.max_seq_len, d.plan.max_seqs_per_window, d.plan.node_cap, d.plan.n_slots, "work GB", d.work_bytes / 1e9, flush=True)
    dt, st = timeit(lambda: d.run(p, s), reps=1)
    print("poa windows", ws.n_windows, "ms", round(dt * 1e3, 2), st, flush=True)
if "bsw" in which:
    b = gen_bsw(2_000_000, 1002)
    d = B.DeviceBswBatch(b, dev)
    p = B.make_params()
    dt, st = timeit(lambda: d.run(p, s))
    print("bsw pairs", b.n, "ms", round(dt * 1e3, 2), st, "GCUPS", round(b.nominal_cells / dt / 1e9, 1), flush=True)
