#!/usr/bin/env bash
# Where the grouped stream path overtakes one pair per wavefront (GBX_PHMM_SMALL: 0 stream / 1 tiled; default: tiled under 12 000 pairs):
# one host call of 4 ... 32 of the reference driver's batches, pairs per unit 1: scripts/sweep_phmm_small.sh > gpurun_out/<tag>_phmm_small.txt
for units in 4 8 12 16 20 24 32; do
  for small in 0 1; do
    echo -n "batches $units small $small: "; GBX_PHMM_SMALL=$small GBX_PHMM_SEG=1 python scripts/dbg_combined_call.py phmm $units 15 2>/dev/null | grep -E "median|pairs" | tr '\n' ' '; echo
  done
done
