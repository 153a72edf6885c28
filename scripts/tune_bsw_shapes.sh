#!/bin/bash
# Time each query-length class of the bsw large workload under alternative (lanes x columns) shapes.
# Usage (GPU box): bash scripts/tune_bsw_shapes.sh
for v in "4x4,4x8,4x12,4x16,8x10,8x12,8x14,8x16,16x10,16x12,16x16,64x16" \
         "8x2,8x4,8x6,8x8,8x10,8x12,8x14,8x16,16x10,16x12,16x16,64x16" \
         "16x1,16x2,16x3,16x4,16x5,16x6,16x7,16x8,16x10,16x12,16x16,64x16"; do
  echo "== $v"
  GBX_BSW_SHAPES=$v timeout 300 python scripts/perf_all.py bsw 2>&1 | tail -1
done
