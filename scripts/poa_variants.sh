#!/bin/bash
# poa 'large' under the forms and tuning aids of poa_kernels.hip: ms per step and the stage times of each (on the GPU box).
#   scripts/poa_variants.sh <tag>
tag=${1:-r04}
out=gpurun_out/${tag}_poa_variants.txt; : > $out
run() {
    echo "== $*" | tee -a $out
    env "$@" timeout 600 python3 bench.py --kernel poa --steps 3 --warmup 1 --no-cpu 2>/dev/null | python3 -c '
import json,sys
for ln in sys.stdin:
    if ln.startswith("{"):
        d=json.loads(ln); print(json.dumps({"ms_per_step": round(d["ms_per_step"],2), "gcups": round(d["value"],1), "verified": d.get("gather_verified"), "kernels_ms": {k: round(v,3) for k,v in d["kernels_ms"].items()}}))
' | tee -a $out
}
run GBX_POA_LOCKSTEP=0
run GBX_POA_LOCKSTEP=1 GBX_POA_DP_OCC=5 GBX_POA_TB_SERIAL=0
run GBX_POA_LOCKSTEP=1 GBX_POA_DP_OCC=6 GBX_POA_TB_SERIAL=0
run GBX_POA_LOCKSTEP=1 GBX_POA_DP_OCC=5 GBX_POA_TB_SERIAL=1
run GBX_POA_LOCKSTEP=1 GBX_POA_DP_OCC=6 GBX_POA_TB_SERIAL=1
