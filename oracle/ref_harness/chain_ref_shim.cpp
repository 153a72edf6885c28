// chain_ref_shim.cpp — our own flat-array shim around the *reference's*
// host_chain_kernel (R/benchmarks/chain/src/host_kernel.cpp:96-108), so tests
// can call the real reference arithmetic.  Built by oracle/build_ref.sh into
// oracle/_ref/libchain_ref.so.  TEST INFRASTRUCTURE ONLY; build container only.
#include <cstdint>
#include <vector>
#include "host_data.h"      // from -I/root/reference/benchmarks/chain/src
#include "host_kernel.h"
#include "../../include/gbx.h"

extern "C" int ref_chain(int64_t n_calls, const int64_t *anchor_off,
                         const uint64_t *ax, const uint64_t *ay, const gbx_chain_call *hdr,
                         int32_t *score, int32_t *parent, int32_t *target, int32_t *peak,
                         int nthreads)
{
    std::vector<call_t> calls((size_t)n_calls);
    std::vector<return_t> rets((size_t)n_calls);
    for (int64_t c = 0; c < n_calls; ++c) {
        const int64_t o = anchor_off[c], n = anchor_off[c + 1] - o;
        call_t &a = calls[(size_t)c];
        a.n = n; a.avg_qspan = hdr[c].avg_qspan;
        a.max_dist_x = hdr[c].max_dist_x; a.max_dist_y = hdr[c].max_dist_y;
        a.bw = hdr[c].bw; a.n_segs = hdr[c].n_segs;
        a.anchors.resize((size_t)n);
        for (int64_t i = 0; i < n; ++i) { a.anchors[(size_t)i].x = ax[o + i]; a.anchors[(size_t)i].y = ay[o + i]; }
    }
    host_chain_kernel(calls, rets, nthreads < 1 ? 1 : nthreads);
    for (int64_t c = 0; c < n_calls; ++c) {
        const int64_t o = anchor_off[c], n = anchor_off[c + 1] - o;
        const return_t &r = rets[(size_t)c];
        for (int64_t i = 0; i < n; ++i) {
            score[o + i] = r.scores[(size_t)i];
            parent[o + i] = r.parents[(size_t)i];
            if (target) target[o + i] = r.targets[(size_t)i];
            if (peak) peak[o + i] = r.peak_scores[(size_t)i];
        }
    }
    return 0;
}
