import sys, time
sys.path.insert(0, ".")
from genomicsbench_amd import _native as N
from genomicsbench_amd.datagen import gen_poa
from genomicsbench_amd.poa import consensus_host, make_params
ws = gen_poa(6000, 4001); p = make_params()
N.check(N.lib().gbx_host_prepare())
for i in range(4):
    t = time.perf_counter(); r = consensus_host(p, ws); print("call %d: %.3f s" % (i, time.perf_counter() - t), flush=True)
