#!/usr/bin/env bash
# gbx_bsw_extend_host on 'large' under several chunkings of the pipelined call (GBX_BSW_HOST_CUTS = fractions of the pairs where a
# chunk ends; default: thirds), interleaved, 40 calls each: scripts/sweep_bsw_host_cuts.sh > gpurun_out/<tag>_bsw_cuts.txt
for round in 1 2; do
  for cuts in "" "0.3" "0.4" "0.5" "0.3,0.62" "0.4,0.72" "0.28,0.64"; do
    if [ -z "$cuts" ]; then python3 scripts/dbg_bsw_host_many.py 40; else GBX_BSW_HOST_CUTS=$cuts python3 scripts/dbg_bsw_host_many.py 40; fi
  done
done
