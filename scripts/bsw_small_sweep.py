#!/usr/bin/env python3
"""Time gbx_bsw_extend_device / gbx_bsw_extend_host / the SeqPair drop-in on small batches of pairs."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from genomicsbench_amd import _native as N  # noqa: E402
from genomicsbench_amd.bsw import BandedPairWiseSW, DeviceBswBatch, extend_host, fill_scmat, make_params  # noqa: E402
from genomicsbench_amd.datagen import gen_bsw  # noqa: E402

dev = torch.device("cuda:0")
p = make_params()
N.check(N.lib().gbx_host_prepare())
for n in (64, 512, 4096, 32768, 100000):
    b = gen_bsw(n, 1002)
    d = DeviceBswBatch(b, dev)
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        d.run(p, s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        d.run(p, s)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    out = np.zeros((b.n, 6), dtype=np.int32)
    for _ in range(3):
        extend_host(p, b, out)
    t0 = time.perf_counter()
    for _ in range(20):
        extend_host(p, b, out)
    dh = (time.perf_counter() - t0) / 20
    pairs = np.zeros(b.n, dtype=N.SEQPAIR_DTYPE)
    pairs["idr"], pairs["idq"], pairs["id"] = b.idr, b.idq, np.arange(b.n)
    pairs["len1"], pairs["len2"], pairs["h0"] = b.len1, b.len2, b.h0
    sw = BandedPairWiseSW(6, 1, 6, 1, 100, 5, fill_scmat(1, 4, -1), 1, 4, 1)
    for _ in range(3):
        sw.getScores16(pairs, b.ref, b.qer, b.n, 1, 100)
    t0 = time.perf_counter()
    for _ in range(20):
        sw.getScores16(pairs, b.ref, b.qer, b.n, 1, 100)
    ds = (time.perf_counter() - t0) / 20
    print("pairs %6d: device entry %.3f ms, host entry %.3f ms, SeqPair drop-in %.3f ms" % (n, dt * 1e3, dh * 1e3, ds * 1e3), flush=True)
