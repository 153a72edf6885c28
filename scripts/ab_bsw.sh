#!/bin/bash
# bsw 'large': the in-tree build against another build of the library (GBX_LIB), alternating, on one box
tag=${1:-r04}; other=${2:-build_tmp/libgbx_prev.so}; reps=${3:-3}
for rep in $(seq $reps); do for lib in intree other; do
  echo "== bsw $lib"
  if [ $lib = intree ]; then timeout 300 python3 bench.py --kernel bsw --steps 20 --warmup 5 --no-cpu 2>/dev/null; else GBX_LIB=$PWD/$other timeout 300 python3 bench.py --kernel bsw --steps 20 --warmup 5 --no-cpu 2>/dev/null; fi | python3 -c '
import json,sys
for ln in sys.stdin:
    if ln.startswith("{"):
        d=json.loads(ln); print(round(d["ms_per_step"],3), round(d["value"],1), {k: round(v,3) for k,v in d["kernels_ms"].items() if v > 0.3})
'
done; done 2>&1 | tee gpurun_out/${tag}_ab_bsw.txt
