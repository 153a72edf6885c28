#!/bin/bash
# bsw 'large' with the in-tree library under several environment settings, alternating, on one box.
# usage: ab_bsw_env.sh <tag> <reps> "<env assignments A>" "<env assignments B>" ...
tag=$1; reps=$2; shift 2
for rep in $(seq $reps); do for v in "$@"; do
  echo "== bsw [$v]"
  env $v timeout 300 python3 bench.py --kernel bsw --steps 20 --warmup 5 --no-cpu 2>/dev/null | python3 -c '
import json,sys
for ln in sys.stdin:
    if ln.startswith("{"):
        d=json.loads(ln); print(round(d["ms_per_step"],3), round(d["value"],1), {k: round(v,3) for k,v in d["kernels_ms"].items() if v > 0.3})
'
done; done 2>&1 | tee gpurun_out/${tag}_ab_bsw_env.txt
