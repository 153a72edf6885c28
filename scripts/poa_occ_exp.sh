#!/bin/bash
# poa 'large': the default window kernel against the 16-windows-per-CU instance (128 VGPRs, four ring rows, node factor 4)
out=gpurun_out/${1:-r04}_poa_occ.txt; : > $out
line() { python3 -c '
import json,sys
for ln in sys.stdin:
    if ln.startswith("{"):
        d=json.loads(ln); print(json.dumps({"ms_per_step": round(d["ms_per_step"],2), "gcups": round(d["value"],1), "verified": (d.get("gather_verified") or "")[-10:], "slots": d["config"].get("slots"), "kernels_ms": {k: round(v,2) for k,v in d["kernels_ms"].items()}}))
'; }
run() { echo "== $*" | tee -a $out; env "$@" timeout 400 python3 bench.py --kernel poa --steps 3 --warmup 1 --no-cpu 2>/dev/null | line | tee -a $out; }
for rep in 1 2; do
run GBX_POA_OCC=3
run GBX_POA_NODE_FACTOR=4
run GBX_POA_OCC=4 GBX_POA_NODE_FACTOR=4 GBX_POA_MAX_WAVES=16
run GBX_POA_OCC=4 GBX_POA_NODE_FACTOR=4 GBX_POA_MAX_WAVES=14
done
