// chain_hostkernel_shim.cpp — host_chain_kernel (R/benchmarks/chain/src/host_kernel.h:6, defined in
// host_kernel.cpp:96-108) on the C-ABI of libgbx.so.  Compiled only together with the reference's own
// main.cpp / host_data_io.cpp / host_data.h from where they lie (oracle/build_ref.sh), in place of
// host_kernel.cpp: the unmodified driver then chains on the GPU.  No reference code is copied here.
#include <cstdio>
#include <cstdlib>
#include "host_kernel.h"                   // the reference's header, -I R/benchmarks/chain/src
#include "gbx.h"

void host_chain_kernel(std::vector<call_t> &args, std::vector<return_t> &rets, int numThreads)
{
    (void)numThreads;
    const size_t nc = args.size();
    std::vector<int64_t> off(nc + 1, 0);
    for (size_t c = 0; c < nc; ++c) off[c + 1] = off[c] + (int64_t)args[c].n;
    const size_t na = (size_t)off[nc];
    std::vector<uint64_t> ax(na), ay(na);
    std::vector<gbx_chain_call> hdr(nc);
    for (size_t c = 0; c < nc; ++c) {
        const call_t &a = args[c];
        for (int64_t k = 0; k < (int64_t)a.n; ++k) { ax[(size_t)(off[c] + k)] = a.anchors[(size_t)k].x; ay[(size_t)(off[c] + k)] = a.anchors[(size_t)k].y; }
        hdr[c].avg_qspan = a.avg_qspan; hdr[c].max_dist_x = a.max_dist_x; hdr[c].max_dist_y = a.max_dist_y;
        hdr[c].bw = a.bw; hdr[c].n_segs = a.n_segs;
    }
    std::vector<int32_t> score(na), parent(na), target(na), peak(na);
    const int rc = gbx_chain_host((int64_t)nc, off.data(), ax.data(), ay.data(), hdr.data(),
                                  score.data(), parent.data(), target.data(), peak.data());
    if (rc) { fprintf(stderr, "host_chain_kernel: %s\n", gbx_last_error()); exit(EXIT_FAILURE); }
    for (size_t c = 0; c < nc; ++c) {
        return_t &r = rets[c];
        const size_t b = (size_t)off[c], e = (size_t)off[c + 1];
        r.n = args[c].n;
        r.scores.assign(score.begin() + b, score.begin() + e);
        r.parents.assign(parent.begin() + b, parent.begin() + e);
        r.targets.assign(target.begin() + b, target.begin() + e);
        r.peak_scores.assign(peak.begin() + b, peak.begin() + e);
    }
}
