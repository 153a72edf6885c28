#!/bin/bash
# One bench kernel with the in-tree library under several environment settings, alternating, on one box.
# usage: ab_env.sh <tag> <kernel> <reps> "<env assignments A>" "<env assignments B>" ...   ("-" = no assignment)
tag=$1; k=$2; reps=$3; shift 3
out=gpurun_out/${tag}_ab_env.txt; : > $out
for rep in $(seq $reps); do for v in "$@"; do
  echo "== $k [$v]" | tee -a $out
  if [ "$v" = "-" ]; then v="GBX_NOOP=1"; fi
  env $v timeout 300 python3 bench.py --kernel $k --steps 3 --warmup 1 --no-cpu 2>/dev/null | python3 -c '
import json,sys
for ln in sys.stdin:
    if ln.startswith("{"):
        d=json.loads(ln); print(json.dumps({"ms_per_step": round(d["ms_per_step"],3), "value": round(d["value"],1), "verified": (d.get("gather_verified") or "")[-9:], "kernels_ms": {k: round(v,3) for k,v in d["kernels_ms"].items() if v > 0.3}}))
' | tee -a $out
done; done
