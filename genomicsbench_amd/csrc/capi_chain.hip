// capi_chain.hip — chain entries of the C-ABI (include/gbx.h).
#include "capi_common.h"
#include "chain_split.h"

using namespace gbx;

extern "C" {

/* ------------------------------------------------------------------- chain */
size_t gbx_chain_workspace_bytes(int64_t n_calls, int64_t n_anchors) { return chain_workspace_bytes(n_calls, n_anchors); }

int gbx_chain_device(int64_t n_calls, int64_t n_anchors, const int64_t *d_anchor_off,
                     const uint64_t *d_ax, const uint64_t *d_ay, const gbx_chain_call *d_hdr,
                     int32_t *d_score, int32_t *d_parent, int32_t *d_target, int32_t *d_peak,
                     void *d_work, size_t work_bytes, void *stream)
{
    if (n_calls < 0 || n_anchors < 0) { set_error("gbx_chain_device: bad argument"); return GBX_ERR_ARG; }
    if (n_calls == 0) return GBX_OK;
    if (!d_anchor_off || !d_ax || !d_ay || !d_hdr || !d_score || !d_parent || !d_work) {
        set_error("gbx_chain_device: null pointer");
        return GBX_ERR_ARG;
    }
    int rc = require_device();
    if (rc) return rc;
    return chain_launch(n_calls, n_anchors, d_anchor_off, d_ax, d_ay, d_hdr, d_score, d_parent, d_target, d_peak,
                        d_work, work_bytes, (hipStream_t)stream);
}

// One device (the calling thread's current one).  `base` = index of call 0 in the caller's job (error texts only).
static int chain_host_one(int64_t n_calls, const int64_t *anchor_off, const uint64_t *ax, const uint64_t *ay,
                          const gbx_chain_call *hdr, int32_t *score, int32_t *parent, int32_t *target, int32_t *peak, int64_t base = 0)
{
    RoctxRange range_("gbx_chain_host");
    const bool trace = getenv("GBX_HOST_TRACE") != nullptr;
    const double t_begin = wall_s();
    auto mark = [&](const char *what) { if (trace) fprintf(stderr, "[gbx chain host] %9.3f ms %s\n", (wall_s() - t_begin) * 1e3, what); };
    if (n_calls < 0) { set_error("gbx_chain_host: bad argument"); return GBX_ERR_ARG; }
    if (n_calls == 0) return GBX_OK;
    if (!anchor_off || !hdr || !score || !parent) { set_error("gbx_chain_host: null pointer"); return GBX_ERR_ARG; }
    if (anchor_off[0] != 0) { set_error("gbx_chain_host: anchor_off[0] must be 0"); return GBX_ERR_ARG; }
    for (int64_t c = 0; c < n_calls; ++c) {
        const int64_t n = anchor_off[c + 1] - anchor_off[c];
        if (n < 0) { set_error("gbx_chain_host: anchor_off not monotone at call %lld", (long long)(base + c)); return GBX_ERR_ARG; }
        if (n > 0x7fffffffLL) { set_error("gbx_chain_host: call %lld has more than 2^31 anchors", (long long)(base + c)); return GBX_ERR_UNSUPPORTED; }
    }
    const int64_t na = anchor_off[n_calls];
    if (na > 0 && (!ax || !ay)) { set_error("gbx_chain_host: null anchors"); return GBX_ERR_ARG; }
    int rc = require_device();
    if (rc) return rc;
    HostLane lane;
    if ((rc = lane.acquire())) return rc;
    Lane *L = lane.l;
    DevBuf doff(L), dx(L), dy(L), dh(L), ds(L), dp(L), dt(L), dk(L), dw(L);
    const size_t wb = chain_workspace_bytes(n_calls, na);
    if ((rc = doff.alloc((n_calls + 1) * 8)) || (rc = dx.alloc(na * 8)) || (rc = dy.alloc(na * 8)) ||
        (rc = dh.alloc(n_calls * sizeof(gbx_chain_call))) || (rc = ds.alloc(na * 4)) || (rc = dp.alloc(na * 4)) ||
        (rc = dt.alloc(na * 4)) || (rc = dk.alloc(na * 4)) || (rc = dw.alloc(wb)))
        return rc;
    // one pipeline chunk (host_pipeline.h): staged uploads, the kernels on the lane's compute stream, staged
    // downloads.  The calls of a job share one load-balanced launch, so there is nothing to gain from chunks.
    HostPipe pipe(lane.l, (size_t)na * 16 + (size_t)n_calls * (8 + sizeof(gbx_chain_call)), false);
    // A large staged call: its few longest jobs - each a lone wavefront that works for tens of milliseconds after the rest of the
    // chip has finished (on 'large': 66 ms of the call's 68) - run in a launch of their own (ChainSplit), and everything else comes
    // home while they are still at work: the results of the rest are downloaded behind the main launch (the longest jobs' stretches
    // of the arrays come along unfinished), theirs follow, gathered into one buffer on the device, behind their own.
    // GBX_CHAIN_SPLIT_TOP=<jobs> (0: off; default 32), from GBX_CHAIN_SPLIT_MIN anchors on (default 4 Mi).
    int split_top = 0;
    {
        const char *e = getenv("GBX_CHAIN_SPLIT_TOP"), *em = getenv("GBX_CHAIN_SPLIT_MIN");     /* read per call: the tests vary them */
        const int want = e ? atoi(e) : 32;
        const int64_t min_anchors = em ? atoll(em) : (int64_t)4 << 20;
        if (pipe.staged && want > 0 && na >= min_anchors && n_calls > want) split_top = want < CHAIN_SPLIT_MAX ? want : CHAIN_SPLIT_MAX;
    }
    const int n_arrays = 2 + (target ? 1 : 0) + (peak ? 1 : 0);
    int64_t max_call = 0;
    if (split_top) for (int64_t c = 0; c < n_calls; ++c) max_call = anchor_off[c + 1] - anchor_off[c] > max_call ? anchor_off[c + 1] - anchor_off[c] : max_call;
    DevBuf dtab(L), dpacked(L);
    std::vector<int64_t> h_tab((size_t)1 + 3 * CHAIN_SPLIT_MAX, 0);
    if (split_top && ((rc = dtab.alloc(h_tab.size() * 8)) || (rc = dpacked.alloc((size_t)split_top * (size_t)max_call * 4 * (size_t)n_arrays)))) return rc;
    if ((rc = pipe.prepare(split_top ? 2 : 1))) return rc;
    pipe.stage(0, doff.p, anchor_off, (n_calls + 1) * 8);
    pipe.stage(0, dh.p, hdr, n_calls * sizeof(gbx_chain_call));
    pipe.stage(0, dx.p, ax, na * 8);
    pipe.stage(0, dy.p, ay, na * 8);
    mark("device buffers ready");
    pipe.start();
    if ((rc = pipe.wait_stage(0))) return pipe.finish(rc);
    mark("uploads queued");
    ChainSplit sp;
    std::vector<HostPipe::Seg> segs;
    if (split_top) {
        // (the lane's copy stream: this call's pipe does its transfers on the compute stream, and the lane is this call's alone)
        sp.top = split_top; sp.side = L->copy;
        sp.ev_rest = pipe.join_events(0)[0]; sp.ev_top = pipe.join_events(1)[0]; sp.ev_fork = pipe.join_events(1)[1];
        sp.d_tab = dtab.as<int64_t>(); sp.h_tab = h_tab.data(); sp.d_packed = dpacked.as<int32_t>();
        sp.n_arrays = n_arrays;
        int a = 0;
        sp.src[a++] = ds.as<int32_t>(); sp.src[a++] = dp.as<int32_t>();
        if (target) sp.src[a++] = dt.as<int32_t>();
        if (peak) sp.src[a++] = dk.as<int32_t>();
        for (; a < 4; ++a) sp.src[a] = ds.as<int32_t>();
    }
    rc = chain_launch_split(n_calls, na, doff.as<int64_t>(), dx.as<uint64_t>(), dy.as<uint64_t>(), dh.as<gbx_chain_call>(),
                            ds.as<int32_t>(), dp.as<int32_t>(), dt.as<int32_t>(), dk.as<int32_t>(), dw.p, wb, lane.l->compute, split_top ? &sp : nullptr);
    if (rc) return pipe.finish(rc);
    pipe.fetch(0, score, ds.p, na * 4);
    pipe.fetch(0, parent, dp.p, na * 4);
    if (target) pipe.fetch(0, target, dt.p, na * 4);
    if (peak) pipe.fetch(0, peak, dk.p, na * 4);
    if ((rc = pipe.chunk_launched(0, split_top ? 1 : 0))) return pipe.finish(rc);
    if (split_top) {
        // the longest jobs' results: one packed buffer, job after job, array after array - delivered over what the first download
        // left in their stretches of the caller's arrays
        const int m = (int)h_tab[0];
        int32_t *const outs[4] = {score, parent, target ? target : peak, target ? peak : nullptr};
        size_t bytes = 0;
        for (int k = 0; k < m; ++k) {
            const int64_t start = h_tab[(size_t)1 + 3 * k], n = h_tab[(size_t)2 + 3 * k];
            if (start < 0 || n < 0 || start + n > na || n > max_call) { set_error("gbx_chain_host: inconsistent job table"); return pipe.finish(GBX_ERR_HIP); }
            for (int a = 0; a < n_arrays; ++a) { segs.push_back(HostPipe::Seg{(char *)(outs[a] + start), (size_t)n * 4}); bytes += (size_t)n * 4; }
        }
        if (bytes) pipe.fetch_scatter(1, dpacked.p, bytes, &segs);
        if ((rc = pipe.chunk_launched(1, 1))) return pipe.finish(rc);
    }
    mark("kernels queued");
    rc = pipe.finish();
    mark("results downloaded");
    return rc;
}

// The host entry: one device, or the calls cut into contiguous ranges of equal anchor counts (every anchor looks back
// over a bounded window: a call's work is linear in its length) over the devices of gbx_host_set_devices / GBX_GPUS -
// host_chain_kernel's OpenMP loop over calls (host_kernel.cpp:98-107) as a loop over devices.  parent / target values
// are indices inside a call, so a shard's results are the job's.
int gbx_chain_host(int64_t n_calls, const int64_t *anchor_off, const uint64_t *ax, const uint64_t *ay,
                   const gbx_chain_call *hdr, int32_t *score, int32_t *parent, int32_t *target, int32_t *peak)
{
    if (!host_multi_wanted() || n_calls <= 0 || !anchor_off || !hdr || !score || !parent || anchor_off[0] != 0)
        return chain_host_one(n_calls, anchor_off, ax, ay, hdr, score, parent, target, peak);
    for (int64_t c = 0; c < n_calls; ++c)
        if (anchor_off[c + 1] < anchor_off[c]) return chain_host_one(n_calls, anchor_off, ax, ay, hdr, score, parent, target, peak);   // names the call
    if (anchor_off[n_calls] > 0 && (!ax || !ay)) return chain_host_one(n_calls, anchor_off, ax, ay, hdr, score, parent, target, peak);
    int map[MAX_HOST_DEVICES];
    const int n_dev = host_device_set(map);
    if (n_dev < 0) return n_dev;
    int parts = shard_parts(n_dev, anchor_off[n_calls], 1 << 20);      // a million anchors per shard at least
    if (parts > n_calls) parts = (int)n_calls;
    if (parts == 1) {
        DeviceGuard g;
        int rc = g.set(map[host_next_small_call_device(n_dev)]);
        return rc ? rc : chain_host_one(n_calls, anchor_off, ax, ay, hdr, score, parent, target, peak);
    }
    const std::vector<int64_t> cuts = split_by_cost(n_calls, parts, [&](int64_t c) { return (double)(anchor_off[c + 1] - anchor_off[c]); });
    return run_on_devices(parts, map, "gbx_chain_host", [&](int k) -> int {
        const int64_t lo = cuts[(size_t)k], hi = cuts[(size_t)k + 1], m = hi - lo;
        if (m == 0) return GBX_OK;
        const int64_t a = anchor_off[lo];
        std::vector<int64_t> off2((size_t)m + 1);
        for (int64_t c = 0; c <= m; ++c) off2[(size_t)c] = anchor_off[lo + c] - a;
        return chain_host_one(m, off2.data(), ax ? ax + a : ax, ay ? ay + a : ay, hdr + lo, score + a, parent + a, target ? target + a : nullptr,
                              peak ? peak + a : nullptr, lo);
    });
}

int gbx_chain_evaluated_pairs(const void *d_work, int64_t *pairs, void *stream)
{
    if (!d_work || !pairs) { set_error("gbx_chain_evaluated_pairs: null pointer"); return GBX_ERR_ARG; }
    return chain_read_evaluated(d_work, pairs, (hipStream_t)stream);
}

int gbx_chain_job_stats(const void *d_work, int64_t n_calls, int64_t n_anchors, int64_t *jobs, int64_t *longest_job, void *stream)
{
    if (!d_work || !jobs || !longest_job) { set_error("gbx_chain_job_stats: null pointer"); return GBX_ERR_ARG; }
    return chain_read_job_stats(d_work, n_calls, n_anchors, jobs, longest_job, (hipStream_t)stream);
}


}  // extern "C"
