#!/bin/bash
# A tuning build of libgbx.so with extra -D flags on ONE kernel file: scripts/build_variant.sh <name> <file stem> <flags...>
#   -> build_tmp/libgbx_<name>.so (the other objects are the in-tree build's); load it with GBX_LIB=<path> (Python mirror)
set -e
ROOT=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)
name=$1; stem=$2; shift 2
mkdir -p $ROOT/build_tmp
cd $ROOT/genomicsbench_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=off "$@" -c $stem.hip -o $ROOT/build_tmp/${stem}_$name.o
objs=$(ls *.o | grep -v "^$stem.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs $ROOT/build_tmp/${stem}_$name.o -o $ROOT/build_tmp/libgbx_$name.so
echo built build_tmp/libgbx_$name.so
