#!/bin/bash
# Time each query-length class of the bsw large workload under alternative (lanes x columns) shapes
# (class kernels serialised so that their times are comparable).  Usage (GPU box): bash scripts/tune_bsw_shapes.sh
for v in "2x8,2x16,2x24,4x16,4x20,4x24,8x14,8x16,16x10,16x12,16x16,64x16" \
         "4x4,4x8,4x12,4x16,8x10,8x12,8x14,8x16,16x10,16x12,16x16,64x16" \
         "16x4,16x4,16x4,16x4,16x6,16x6,16x8,16x8,16x10,16x12,16x16,64x16"; do
  echo "== $v"
  GBX_BSW_SERIAL=1 GBX_BSW_SHAPES=$v timeout 300 python scripts/perf_all.py bsw 2>&1 | tail -1
done
