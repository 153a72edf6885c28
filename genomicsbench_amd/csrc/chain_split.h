// chain_split.h - the host entry's two-launch form of chain_launch (chain_kernels.hip, capi_chain.hip).
// Kept out of gbx_internal.h: the counter tables under profiles/ are stamped with a hash of each kernel file and gbx_internal.h
// (genomicsbench_amd/srchash.py), and this touches chain only.
#pragma once
#include "gbx_internal.h"

namespace gbx {

// The host entry's large calls: the `top` longest jobs of the launch run in a launch of their own on a side stream, everything else
// on `s` - so that the caller can bring the results of everything else home while the longest jobs, each a lone wavefront for tens
// of milliseconds after the rest of the chip has finished, are still at work (capi_chain.hip).  chain_launch_split fills h_tab before
// it returns: h_tab[0] = m (jobs in the top launch), then per job its first anchor, its anchors, and where its results start in
// d_packed (in ints): there a gather kernel lays, job after job, the job's ranges of the n_arrays result arrays named in src[].
struct ChainSplit {
    int top;                              // at most CHAIN_SPLIT_MAX
    hipStream_t side;
    hipEvent_t ev_fork, ev_rest, ev_top;  // the job table is on the host / everything else is done (on s) / the top jobs and their gather are done (on side)
    int64_t *d_tab, *h_tab;               // 1 + 3 * top words each
    int32_t *d_packed;
    int n_arrays;
    const int32_t *src[4];
};
constexpr int CHAIN_SPLIT_MAX = 64;
int chain_launch_split(int64_t n_calls, int64_t n_anchors, const int64_t *d_off,
                       const uint64_t *d_ax, const uint64_t *d_ay, const gbx_chain_call *d_hdr,
                       int32_t *d_score, int32_t *d_parent, int32_t *d_target, int32_t *d_peak,
                       void *d_work, size_t work_bytes, hipStream_t s, ChainSplit *split);

}  // namespace gbx
