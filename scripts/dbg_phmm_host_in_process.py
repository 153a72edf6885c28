"""Why is gbx_phmm_forward_host 30 ms slower inside bench.py than alone?  The same call alone, after torch has initialised the
device, and with a device-resident copy of the job alive beside it (what bench.py holds): dbg_phmm_host_in_process.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from genomicsbench_amd import _native as N
from genomicsbench_amd.datagen import gen_phmm
from genomicsbench_amd.phmm import forward_host, DevicePhmmBatchSet
N.check(N.lib().gbx_host_prepare())
b = gen_phmm(20000, 3001)


def timed(tag):
    ms = []
    for _ in range(4):
        t = time.perf_counter(); forward_host(b); ms.append((time.perf_counter() - t) * 1e3)
    print("%-60s first %.1f ms, best %.1f ms" % (tag, ms[0], min(ms)), flush=True)


timed("alone")
import torch
torch.zeros(1, device="cuda:0"); torch.cuda.synchronize()
timed("after torch initialised the device")
d = DevicePhmmBatchSet(b, torch.device("cuda:0"))
s = torch.cuda.current_stream().cuda_stream
d.run(s); torch.cuda.synchronize()
timed("with the device-resident job and its workspace alive")
d.run(s); d.run(s); torch.cuda.synchronize()
timed("... after two more device-resident steps")
time.sleep(2.0)
timed("... after two seconds of rest")
os.environ["GBX_HOST_TRACE"] = "1"
forward_host(b)
del os.environ["GBX_HOST_TRACE"]
for k in range(3):
    d.run(s)
torch.cuda.synchronize()
os.environ["GBX_HOST_TRACE"] = "1"
forward_host(b)
