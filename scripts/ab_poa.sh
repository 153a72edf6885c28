#!/bin/bash
# poa 'large': the in-tree build against another build of the library (GBX_LIB), alternating, on one box -> gpurun_out/<tag>_ab_poa.txt
tag=${1:-r04}; other=${2:-build_tmp/libgbx_prev.so}; out=gpurun_out/${tag}_ab_poa.txt; : > $out
line() { python3 -c '
import json,sys
for ln in sys.stdin:
    if ln.startswith("{"):
        d=json.loads(ln); print(json.dumps({"ms_per_step": round(d["ms_per_step"],3), "value": round(d["value"],1), "verified": d.get("gather_verified"), "kernels_ms": {k: round(v,3) for k,v in d["kernels_ms"].items() if v > 0.3}}))
'; }
for rep in 1 2 3; do
  echo "== poa in-tree" | tee -a $out
  timeout 300 python3 bench.py --kernel poa --steps 3 --warmup 1 --no-cpu 2>/dev/null | line | tee -a $out
  echo "== poa $other" | tee -a $out
  GBX_LIB=$PWD/$other timeout 300 python3 bench.py --kernel poa --steps 3 --warmup 1 --no-cpu 2>/dev/null | line | tee -a $out
done
