#!/bin/bash
# One bench kernel under several builds of the library (in-tree + GBX_LIB variants), interleaved, on one box.
#   scripts/ab_libs.sh <tag> <kernel> <reps> <lib or "intree"> ...   -> gpurun_out/<tag>_ab_libs.txt
tag=$1; k=$2; reps=$3; shift 3
out=gpurun_out/${tag}_ab_libs.txt; : > $out
line() { python3 -c '
import json,sys
for ln in sys.stdin:
    if ln.startswith("{"):
        d=json.loads(ln); print(json.dumps({"ms_per_step": round(d["ms_per_step"],3), "value": round(d["value"],1), "verified": (d.get("gather_verified") or "")[-9:], "kernels_ms": {k: round(v,3) for k,v in d["kernels_ms"].items() if v > 0.3}}))
'; }
for rep in $(seq $reps); do
  for lib in "$@"; do
    echo "== $k $lib" | tee -a $out
    if [ "$lib" = intree ]; then timeout 300 python3 bench.py --kernel $k --steps 3 --warmup 1 --no-cpu 2>/dev/null | line | tee -a $out
    else GBX_LIB=$PWD/$lib timeout 300 python3 bench.py --kernel $k --steps 3 --warmup 1 --no-cpu 2>/dev/null | line | tee -a $out; fi
  done
done
