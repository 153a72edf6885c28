#!/usr/bin/env python3
"""Time gbx_phmm_forward_device on jobs of a few batches, grouped stream path vs the small-job path."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from genomicsbench_amd.datagen import gen_phmm  # noqa: E402
from genomicsbench_amd.phmm import DevicePhmmBatchSet, forward_host  # noqa: E402

dev = torch.device("cuda:0")
for nb in (1, 2, 4, 8, 16, 32, 64):
    bs = gen_phmm(nb, 3001)
    line = "batches %3d pairs %6d:" % (nb, bs.n_pairs)
    for mode in ("0", "1"):
        os.environ["GBX_PHMM_SMALL"] = mode
        d = DevicePhmmBatchSet(bs, dev)
        s = torch.cuda.current_stream().cuda_stream
        for _ in range(3):
            d.run(s)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            d.run(s)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20
        t0 = time.perf_counter()
        for _ in range(5):
            forward_host(bs)
        dh = (time.perf_counter() - t0) / 5
        line += "  mode %s device %.3f ms host %.3f ms" % (mode, dt * 1e3, dh * 1e3)
    print(line, flush=True)
