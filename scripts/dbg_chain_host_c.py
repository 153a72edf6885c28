"""gbx_chain_host on chain 'large' called straight through the C-ABI with output arrays that exist and are touched (what a C
driver hands over), a few calls, GBX_HOST_TRACE on the last: python3 scripts/dbg_chain_host_c.py [calls]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from genomicsbench_amd import _native as N
from genomicsbench_amd.datagen import gen_chain
off, ax, ay, hdr = gen_chain(10000, 2001)
off = np.ascontiguousarray(off, dtype=np.int64); ax = np.ascontiguousarray(ax, dtype=np.uint64); ay = np.ascontiguousarray(ay, dtype=np.uint64)
hdr = np.ascontiguousarray(hdr, dtype=N.CHAIN_CALL_DTYPE)
n = int(off[-1])
outs = [np.full(n, -7, dtype=np.int32) for _ in range(4)]
N.check(N.lib().gbx_host_prepare())
def call():
    t = time.perf_counter()
    N.check(N.lib().gbx_chain_host(len(off) - 1, N.ptr(off), N.ptr(ax), N.ptr(ay), N.ptr(hdr), *[N.ptr(o) for o in outs]))
    return (time.perf_counter() - t) * 1e3
ms = [call() for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5)]
print("calls ms:", " ".join("%.1f" % m for m in ms), "| env", {k: v for k, v in os.environ.items() if k.startswith("GBX_")}, flush=True)
os.environ["GBX_HOST_TRACE"] = "1"
call()
