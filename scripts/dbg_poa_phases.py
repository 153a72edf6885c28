import sys, time, ctypes as C
sys.path.insert(0, ".")
import numpy as np, torch
from genomicsbench_amd import _native as N
from genomicsbench_amd.datagen import gen_poa
from genomicsbench_amd import poa as PO
dev = torch.device("cuda:0"); s = torch.cuda.current_stream().cuda_stream
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
ws = gen_poa(n, 4001); p = PO.make_params(); d = PO.DevicePoaWindowSet(ws, dev)
d.run(p, s); torch.cuda.synchronize()
t = time.perf_counter(); d.run(p, s); torch.cuda.synchronize(); dt = time.perf_counter() - t
tail = d.work[d.work_bytes - 128:].cpu().numpy().view(np.uint64)
print("windows", n, "ms", round(dt * 1e3, 1), "cells", tail[0], "cycles dp/tb/add/cons", tail[1:5], "shares", np.round(tail[1:5] / max(1, tail[1:5].sum()), 3), flush=True)
print("topo (cumulative over both runs; block 0 snapshot): cycles", tail[5], "nodes", tail[6], "cyc/node", tail[5] / max(1, tail[6]),
      "visits", tail[7], "visits/node", tail[7] / max(1, tail[6]), "block loads", tail[8], "dfs-loop cycles", tail[9],
      "dfs cyc/visit", tail[9] / max(1, tail[7]), flush=True)
