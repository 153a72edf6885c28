#!/usr/bin/env bash
# build_ref.sh — compile the REFERENCE's own kernels, from the sources where
# they lie under /root/reference, into oracle/_ref/ (git-ignored, but shipped
# to the GPU box by gpurun).  Test infrastructure only.  No reference source
# is copied into this repo and no stand-in header/library is written.
#
#   oracle/_ref/libbsw_ref.so    bandedSWA.cpp (scalar + AVX2 getScores16) + our shim
#   oracle/_ref/libchain_ref.so  host_kernel.cpp (chain_dp)                + our shim
#
# phmm and poa are NOT buildable here: their arithmetic lives in the GKL and
# spoa submodules, which are empty in this checkout.
#
# Drop-in check (not an oracle): the reference's own, unmodified DRIVERS linked against our GPU library
# instead of the reference kernels, to prove the boundary of include/gbx.h is the one the drivers bind:
#   oracle/_ref/bsw_refdriver_gbx    main_banded.cpp + csrc/shims/bsw_class_shim.cpp        (no bandedSWA.cpp)
#   oracle/_ref/chain_refdriver_gbx  main.cpp host_data_io.cpp common.cpp + chain_hostkernel_shim.cpp (no host_kernel.cpp)
#   oracle/_ref/phmm_refdriver_gbx   PairHMMUnitTest.cpp + our libgkl_pairhmm_c.so (csrc/shims/gkl_pairhmm_shim.cpp)
#   oracle/_ref/poa_refdriver_gbx    msa_spoa_omp.cpp + the product's spoa facade include/spoa/*.hpp (our code over the C-ABI)
#   oracle/_ref/bsw_members_gbx      ref_harness/bsw_members_harness.cpp (every public member of BandedPairWiseSW) + the shim
#   oracle/_ref/bsw_members_ref      the same harness + the reference's bandedSWA.cpp (CPU)
# They need genomicsbench_amd/libgbx.so (make -C genomicsbench_amd/csrc first) and a GPU at run time.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
REF="${GBX_REFERENCE_DIR:-/root/reference}"
OUT="$HERE/_ref"
if [ ! -d "$REF/benchmarks/bsw" ]; then
    echo "build_ref.sh: $REF not present (GPU box?) - keeping prebuilt oracle/_ref" >&2
    exit 0
fi
mkdir -p "$OUT"
CXX="${CXX:-g++}"

# ---- bsw: flags of R/benchmarks/bsw/Makefile:59 with arch=avx2 (:39-44)
BSW="$REF/benchmarks/bsw"
$CXX -shared -fPIC -O3 -std=c++11 -fopenmp -mavx2 -w \
    -DSORT_PAIRS -DENABLE_PREFETCH -DBWA_OTHER_ELE=0 \
    -I"$BSW" "$BSW/bandedSWA.cpp" "$HERE/ref_harness/bsw_ref_shim.cpp" \
    -o "$OUT/libbsw_ref.so"

# ---- bsw: the reference's own CPU driver, unmodified (R/benchmarks/bsw/Makefile:59-75 with arch=avx2): BASELINE
# config 0 runs it with -t 1 -b 512 exactly as R/scripts/run-cpu.sh:61 (scripts/run-cpu-bsw-small.sh)
$CXX -O3 -std=c++11 -fopenmp -mavx2 -w -DSORT_PAIRS -DENABLE_PREFETCH -DBWA_OTHER_ELE=0 -I"$BSW" \
    "$BSW/main_banded.cpp" "$BSW/bandedSWA.cpp" -o "$OUT/bsw_refdriver_cpu"

# ---- chain: R/benchmarks/chain/src/host_kernel.cpp names three minimap2
# headers (minimap.h, mmpriv.h, kalloc.h :8-10) from the un-vendored
# tools/minimap2 submodule and uses nothing from them.  We do not write
# stand-ins for them: the translation unit is fed to the compiler through a
# pipe with exactly those three #include lines dropped; every other byte is
# the reference's.  If a future checkout starts using those headers this
# build fails and the chain oracle must be re-labelled "parity unpinned".
CH="$REF/benchmarks/chain/src"
grep -v -E '^#include "(minimap|mmpriv|kalloc)\.h"' "$CH/host_kernel.cpp" |
    $CXX -c -x c++ -fPIC -O3 -std=c++11 -fopenmp -w -I"$CH" - -o "$OUT/chain_host_kernel.o"
$CXX -shared -fPIC -O3 -std=c++11 -fopenmp -w -I"$CH" \
    "$OUT/chain_host_kernel.o" "$HERE/ref_harness/chain_ref_shim.cpp" -o "$OUT/libchain_ref.so"
rm -f "$OUT/chain_host_kernel.o"

# ---- the reference drivers on our library (skipped until libgbx.so exists)
PKG="$HERE/../genomicsbench_amd"
SH="$PKG/csrc/shims"
INC="$HERE/../include"
if [ -f "$PKG/libgbx.so" ] && [ -f "$PKG/libgkl_pairhmm_c.so" ]; then
    RP='-Wl,-rpath,$ORIGIN/../../genomicsbench_amd'
    $CXX -O2 -std=c++11 -fopenmp -mavx2 -w -DSORT_PAIRS -DENABLE_PREFETCH -DBWA_OTHER_ELE=0 -I"$BSW" -I"$INC" \
        "$BSW/main_banded.cpp" "$SH/bsw_class_shim.cpp" -L"$PKG" -lgbx "$RP" -o "$OUT/bsw_refdriver_gbx"
    $CXX -O2 -std=c++11 -fopenmp -w -DPRINT_OUTPUT=1 -I"$CH" -I"$INC" \
        "$CH/main.cpp" "$CH/host_data_io.cpp" "$CH/common.cpp" "$SH/chain_hostkernel_shim.cpp" \
        -L"$PKG" -lgbx "$RP" -o "$OUT/chain_refdriver_gbx"
    # every public member of the class on the shim, and the same harness on the reference's own kernel file
    $CXX -O2 -std=c++11 -fopenmp -mavx2 -w -DSORT_PAIRS -DENABLE_PREFETCH -DBWA_OTHER_ELE=0 -I"$BSW" -I"$INC" \
        "$HERE/ref_harness/bsw_members_harness.cpp" "$SH/bsw_class_shim.cpp" -L"$PKG" -lgbx "$RP" -o "$OUT/bsw_members_gbx"
    $CXX -O2 -std=c++11 -fopenmp -mavx2 -w -DSORT_PAIRS -DENABLE_PREFETCH -DBWA_OTHER_ELE=0 -I"$BSW" \
        "$HERE/ref_harness/bsw_members_harness.cpp" "$BSW/bandedSWA.cpp" -o "$OUT/bsw_members_ref"
    PH="$REF/benchmarks/phmm"
    $CXX -O2 -std=c++11 -fopenmp -msse4.1 -w -DPRINT_OUTPUT -I"$PH" "$PH/PairHMMUnitTest.cpp" \
        -L"$PKG" -lgkl_pairhmm_c -lgbx "$RP" -o "$OUT/phmm_refdriver_gbx"
    PO="$REF/benchmarks/poa"
    $CXX -O2 -std=c++11 -fopenmp -w -DPRINT_OUTPUT -I"$INC" "$PO/msa_spoa_omp.cpp" \
        -L"$PKG" -lgbx "$RP" -o "$OUT/poa_refdriver_gbx"
fi
echo "built: $(ls "$OUT")"
