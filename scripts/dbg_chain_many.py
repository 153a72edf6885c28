import sys, time
sys.path.insert(0, ".")
import numpy as np, torch, faulthandler
faulthandler.dump_traceback_later(100, exit=True)
from genomicsbench_amd.datagen import gen_chain
from genomicsbench_amd.chain import DeviceChainBatch, chain_host
from oracle import oracle_py as O
n = int(sys.argv[1]) if len(sys.argv) > 1 else 9000
case = gen_chain(n, 5, n_override=[60 + (k % 37) for k in range(n)])
t = time.time(); got = chain_host(*case); print("gpu", time.time() - t, flush=True)
want = O.chain_oracle(*case, nthreads=8)
print("equal", all(np.array_equal(a, b) for a, b in zip(got, want)), flush=True)
