#!/bin/bash
# VGPR count per kernel of one .hip file (cross-compiles for gfx950, no GPU needed): scripts/isa_stats.sh bsw_kernels
set -e
ROOT=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)
f=${1:-bsw_kernels}
d=/tmp/isa_$f; mkdir -p $d; cd $d
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off $EXTRA -c "$ROOT"/genomicsbench_amd/csrc/$f.hip -save-temps -o $f.o 2>/dev/null
S=$d/$f-hip-amdgcn-amd-amdhsa-gfx950.s
grep -E "^\s+\.vgpr_count|^\s+\.name:|\.private_segment_fixed_size|\.vgpr_spill_count" $S | paste - - - - | awk '{print $2, "scratch", $4, "vgpr", $6, "spill", $8}' | sed 's/_ZN3gbx12_GLOBAL__N_1[0-9]*//; s/EEvNS0.*i / /'
