"""Timeline of gbx_bsw_extend_host on the 'large' shard (GBX_HOST_TRACE): python3 scripts/dbg_bsw_host_trace.py"""
import os, sys, time
sys.path.insert(0, ".")
os.environ["GBX_HOST_TRACE"] = "1"
import numpy as np
from genomicsbench_amd import _native as N
from genomicsbench_amd.bsw import extend_host, make_params
from genomicsbench_amd.datagen import gen_bsw
b = gen_bsw(2_000_000, 1002); p = make_params()
N.check(N.lib().gbx_host_prepare())
out = np.full((b.n, 6), -1, dtype=np.int32)
for k in range(3):
    t = time.perf_counter(); extend_host(p, b, out); print("call", k, "ms", (time.perf_counter() - t) * 1e3, flush=True)
