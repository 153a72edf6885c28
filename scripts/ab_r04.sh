#!/bin/bash
# A/B runs of round 4's tuning aids on the GPU box -> gpurun_out/<tag>_ab.txt
tag=${1:-r04}; out=gpurun_out/${tag}_ab.txt; : > $out
line() { python3 -c '
import json,sys
for ln in sys.stdin:
    if ln.startswith("{"):
        d=json.loads(ln); print(json.dumps({"ms_per_step": round(d["ms_per_step"],3), "value": round(d["value"],1), "kernels_ms": {k: round(v,3) for k,v in d["kernels_ms"].items() if v > 0.3}}))
'; }
for rep in 1 2; do
for prio in "0,0,0" "-1,-1,0" "-1,0,1" "-1,-1,-1"; do
  echo "== bsw GBX_SIDE_PRIO=$prio" | tee -a $out
  GBX_SIDE_PRIO=$prio timeout 300 python3 bench.py --kernel bsw --steps 20 --warmup 5 --no-cpu 2>/dev/null | line | tee -a $out
done
done
for occ in 3 2 3 2; do
  echo "== poa GBX_POA_OCC=$occ" | tee -a $out
  GBX_POA_OCC=$occ timeout 300 python3 bench.py --kernel poa --steps 3 --warmup 1 --no-cpu 2>/dev/null | line | tee -a $out
done
echo "== poa GBX_POA_OCC=2 GBX_POA_WAVES_PER_CU=8" | tee -a $out
GBX_POA_OCC=2 GBX_POA_WAVES_PER_CU=8 timeout 300 python3 bench.py --kernel poa --steps 3 --warmup 1 --no-cpu 2>/dev/null | line | tee -a $out
