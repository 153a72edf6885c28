"""Step-by-step phmm bring-up on the GPU with flushed prints (debug aid)."""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import faulthandler
faulthandler.dump_traceback_later(50, exit=True)
from genomicsbench_amd import _native as N
from genomicsbench_amd.phmm import forward_host
from oracle import oracle_py as O
import test_phmm_gpu as T

def say(*a):
    print(*a, flush=True)

say("devices", N.device_count())
say("init", N.lib().gbx_phmm_init())
for reads, haps in ([["A"], ["A"]], [["ACGTA"], ["ACGTTA", "AC"]], [["A" * 64], ["A" * 70]], [["ACGT" * 20], ["ACGT" * 25]],
                    [["ACGT" * 38], ["ACGT" * 50]], [["ACGT" * 60], ["ACGT" * 70]], [["ACGT" * 90], ["ACGT" * 70]],
                    [["ACGT" * 120], ["ACGT" * 70]], [["ACGT" * 200], ["ACGT" * 70]]):
    bs = T.make_set(reads, haps, seed=1)
    t = time.time()
    got = forward_host(bs)
    want = O.phmm_oracle(bs)
    say(len(reads[0]), len(haps[0]), "got", got, "want", want, "dt", round(time.time() - t, 3))
