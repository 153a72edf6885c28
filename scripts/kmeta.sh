#!/bin/bash
# register / spill / scratch / LDS metadata of every kernel in one .hip file (cross-compiles for gfx950, no GPU needed):
#   scripts/kmeta.sh poa_kernels [extra hipcc flags]
set -e
ROOT=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)
f=${1:-poa_kernels}; shift || true
d=/tmp/kmeta_$f; mkdir -p $d; cd $d
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off "$@" -c "$ROOT"/genomicsbench_amd/csrc/$f.hip -save-temps -o $f.o 2>/dev/null
S=$d/$f-hip-amdgcn-amd-amdhsa-gfx950.s
awk '/^  - \.agpr_count|^    \.name:|\.sgpr_count|\.sgpr_spill_count|\.vgpr_count|\.vgpr_spill_count|\.private_segment_fixed_size|\.group_segment_fixed_size/{gsub(/^ +(- )?/,""); printf "%s ", $0} /\.wavefront_size/{print ""}' $S \
  | sed 's/_ZN3gbx12_GLOBAL__N_1[0-9]*//g; s/\.agpr_count: [0-9]* //; s/\.private_segment_fixed_size/scratch/; s/\.group_segment_fixed_size/lds/; s/\.sgpr_spill_count/sgpr_spill/; s/\.vgpr_spill_count/vgpr_spill/; s/\.sgpr_count/sgpr/; s/\.vgpr_count/vgpr/'
