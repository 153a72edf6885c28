#!/bin/bash
# The diagnostic build of libgbx.so with every data-dependent device loop bounded (-DGBX_LOOP_GUARD, csrc/gbx_internal.h):
# DIAGNOSTIC ONLY: every launch function of this build ends in hipDeviceSynchronize() (GBX_GUARD_CHECK), which serialises the whole
# device - other callers' streams and the pipelined bsw chunks included - so nothing timed under it is representative.
#   scripts/build_guard.sh  ->  build_tmp/libgbx_guard.so   (load it with GBX_LIB=<path>)
set -e
ROOT=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)
mkdir -p $ROOT/build_tmp/guard
cd $ROOT/genomicsbench_amd/csrc
objs=""
for f in gbx_core capi_bsw capi_chain capi_phmm capi_poa capi_abea capi_fmi bsw_kernels chain_kernels phmm_kernels poa_kernels abea_kernels fmi_kernels; do
  if [ $f.hip -nt $ROOT/build_tmp/guard/$f.o ] || [ gbx_internal.h -nt $ROOT/build_tmp/guard/$f.o ] || [ poa_graph.h -nt $ROOT/build_tmp/guard/$f.o ]; then
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=off -DGBX_LOOP_GUARD -c $f.hip -o $ROOT/build_tmp/guard/$f.o &
  fi
  objs="$objs $ROOT/build_tmp/guard/$f.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o $ROOT/build_tmp/libgbx_guard.so
echo built build_tmp/libgbx_guard.so
