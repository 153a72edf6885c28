// chain_split.h - chain_launch with calls left out (chain_kernels.hip), for the host entry's two launches (capi_chain.hip).
// Kept out of gbx_internal.h: the counter tables under profiles/ are stamped with a hash of each kernel file and gbx_internal.h
// (genomicsbench_amd/srchash.py), and this touches chain only.
#pragma once
#include "gbx_internal.h"

namespace gbx {

// chain_launch, except that a call c with d_skip[c] != 0 gets no job: its stretches of the result arrays are left as they are.
// d_skip == nullptr: chain_launch.
int chain_launch_skip(int64_t n_calls, int64_t n_anchors, const int64_t *d_off,
                      const uint64_t *d_ax, const uint64_t *d_ay, const gbx_chain_call *d_hdr,
                      int32_t *d_score, int32_t *d_parent, int32_t *d_target, int32_t *d_peak,
                      void *d_work, size_t work_bytes, hipStream_t s, const uint8_t *d_skip);

}  // namespace gbx
