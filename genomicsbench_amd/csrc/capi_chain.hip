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
    // A large staged call is as long as its longest calls: each of them a lone wavefront for tens of milliseconds (on 'large': 66 ms of
    // the kernels' 68), in front of which the whole job was uploaded (12.5 ms) and behind which the whole result came home (13 ms).
    // Such a call now runs as two launches.  Its `top` longest calls go first: their anchors are uploaded ahead of everything (a few
    // per cent of the bytes) into arrays of their own and their launch starts on the lane's second stream while the rest of the job
    // is still on its way; the main launch takes every other call (chain_launch_skip), and its results are downloaded behind IT, with
    // the longest calls still at work; theirs follow, scattered into their stretches of the caller's arrays.
    // GBX_CHAIN_SPLIT_TOP=<calls> (0: one launch; default 32), from GBX_CHAIN_SPLIT_MIN anchors on (default 4 Mi).
    int top = 0;
    {
        const char *e = getenv("GBX_CHAIN_SPLIT_TOP"), *em = getenv("GBX_CHAIN_SPLIT_MIN");     /* read per call: the tests vary them */
        const int want = e ? atoi(e) : 32;
        const int64_t min_anchors = em ? atoll(em) : (int64_t)4 << 20;
        if (pipe.staged && want > 0 && na >= min_anchors && n_calls > want) top = want < 4096 ? want : 4096;
    }
    std::vector<int64_t> tcall, toff;                           // the longest calls, by position; their offsets in the arrays of their own
    std::vector<uint8_t> skip;
    std::vector<gbx_chain_call> thdr;
    int64_t tna = 0;
    if (top) {
        std::vector<int64_t> idx((size_t)n_calls);
        for (int64_t c = 0; c < n_calls; ++c) idx[(size_t)c] = c;
        auto len = [&](int64_t c) { return anchor_off[c + 1] - anchor_off[c]; };
        std::nth_element(idx.begin(), idx.begin() + top, idx.end(), [&](int64_t a, int64_t b) { return len(a) != len(b) ? len(a) > len(b) : a < b; });
        tcall.assign(idx.begin(), idx.begin() + top);
        std::sort(tcall.begin(), tcall.end());
        skip.assign((size_t)n_calls, 0);
        toff.assign((size_t)top + 1, 0);
        thdr.resize((size_t)top);
        for (int k = 0; k < top; ++k) {
            const int64_t c = tcall[(size_t)k];
            skip[(size_t)c] = 1; thdr[(size_t)k] = hdr[c];
            toff[(size_t)k + 1] = toff[(size_t)k] + len(c);
        }
        tna = toff[(size_t)top];
        if (tna == 0) top = 0;
    }
    DevBuf toffd(L), txd(L), tyd(L), thd(L), tsd(L), tpd(L), ttd(L), tkd(L), twd(L), skipd(L);
    const size_t twb = top ? chain_workspace_bytes(top, tna) : 0;
    if (top && ((rc = toffd.alloc((size_t)(top + 1) * 8)) || (rc = txd.alloc(tna * 8)) || (rc = tyd.alloc(tna * 8)) ||
                (rc = thd.alloc((size_t)top * sizeof(gbx_chain_call))) || (rc = tsd.alloc(tna * 4)) || (rc = tpd.alloc(tna * 4)) ||
                (rc = ttd.alloc(tna * 4)) || (rc = tkd.alloc(tna * 4)) || (rc = twd.alloc(twb)) || (rc = skipd.alloc((size_t)n_calls))))
        return rc;
    // chunks of the pipe: 0 = the longest calls' uploads, 1 = the job's uploads and the main launch's results, 2 = the longest calls' results
    const int64_t cm = top ? 1 : 0;
    if ((rc = pipe.prepare(top ? 3 : 1))) return rc;
    if (top) {
        pipe.stage(0, toffd.p, toff.data(), (size_t)(top + 1) * 8);
        pipe.stage(0, thd.p, thdr.data(), (size_t)top * sizeof(gbx_chain_call));
        for (int k = 0; k < top; ++k) {
            const int64_t o = anchor_off[tcall[(size_t)k]], n = toff[(size_t)k + 1] - toff[(size_t)k];
            if (!n) continue;
            pipe.stage(0, txd.as<uint64_t>() + toff[(size_t)k], ax + o, (size_t)n * 8);
            pipe.stage(0, tyd.as<uint64_t>() + toff[(size_t)k], ay + o, (size_t)n * 8);
        }
        pipe.stage(cm, skipd.p, skip.data(), (size_t)n_calls);
    }
    pipe.stage(cm, doff.p, anchor_off, (n_calls + 1) * 8);
    pipe.stage(cm, dh.p, hdr, n_calls * sizeof(gbx_chain_call));
    pipe.stage(cm, dx.p, ax, na * 8);
    pipe.stage(cm, dy.p, ay, na * 8);
    mark("device buffers ready");
    pipe.start();
    std::vector<HostPipe::Seg> segs[4];
    if (top) {
        if ((rc = pipe.wait_stage(0))) return pipe.finish(rc);
        // (this pipe's transfers run on the compute stream: the event lies behind the longest calls' uploads and whatever of the
        // job's the workers have queued since - a few pieces)
        hipEvent_t up = pipe.join_events(0)[0], done = pipe.join_events(2)[0];
        GBX_HIP(hipEventRecord(up, L->compute));
        GBX_HIP(hipStreamWaitEvent(L->copy, up, 0));
        rc = chain_launch(top, tna, toffd.as<int64_t>(), txd.as<uint64_t>(), tyd.as<uint64_t>(), thd.as<gbx_chain_call>(),
                          tsd.as<int32_t>(), tpd.as<int32_t>(), target ? ttd.as<int32_t>() : nullptr, peak ? tkd.as<int32_t>() : nullptr,
                          twd.p, twb, L->copy);
        if (rc) return pipe.finish(rc);
        GBX_HIP(hipEventRecord(done, L->copy));
        if ((rc = pipe.chunk_launched(0, 1))) return pipe.finish(rc);
        mark("longest calls queued");
    }
    if ((rc = pipe.wait_stage(cm))) return pipe.finish(rc);
    mark("uploads queued");
    rc = chain_launch_skip(n_calls, na, doff.as<int64_t>(), dx.as<uint64_t>(), dy.as<uint64_t>(), dh.as<gbx_chain_call>(),
                           ds.as<int32_t>(), dp.as<int32_t>(), dt.as<int32_t>(), dk.as<int32_t>(), dw.p, wb, lane.l->compute,
                           top ? skipd.as<uint8_t>() : nullptr);
    if (rc) return pipe.finish(rc);
    pipe.fetch(cm, score, ds.p, na * 4);
    pipe.fetch(cm, parent, dp.p, na * 4);
    if (target) pipe.fetch(cm, target, dt.p, na * 4);
    if (peak) pipe.fetch(cm, peak, dk.p, na * 4);
    if ((rc = pipe.chunk_launched(cm))) return pipe.finish(rc);
    if (top) {
        int32_t *const outs[4] = {score, parent, target, peak};
        const void *const devs[4] = {tsd.p, tpd.p, ttd.p, tkd.p};
        for (int a = 0; a < 4; ++a) {
            if (!outs[a]) continue;
            for (int k = 0; k < top; ++k) {
                const size_t n = (size_t)(toff[(size_t)k + 1] - toff[(size_t)k]) * 4;
                if (n) segs[a].push_back(HostPipe::Seg{(char *)(outs[a] + anchor_off[tcall[(size_t)k]]), n});
            }
            pipe.fetch_scatter(2, devs[a], (size_t)tna * 4, &segs[a]);
        }
        if ((rc = pipe.chunk_launched(2, 1))) return pipe.finish(rc);
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
