// capi_bsw.hip — bsw entries of the C-ABI (include/gbx.h): device entry, host-buffer entry, the SeqPair drop-in.
#include "capi_common.h"

static constexpr int BSW_HOST_WORKERS = 4;        // upload / validation helpers of a bsw host call (host_workers: GBX_HOST_THREADS overrides)

using namespace gbx;

extern "C" {

/* --------------------------------------------------------------------- bsw */
void gbx_bsw_fill_scmat(int a, int b, int ambig, int8_t mat[25])
{
    int k = 0;
    for (int i = 0; i < 4; ++i) {
        for (int j = 0; j < 4; ++j) mat[k++] = (int8_t)(i == j ? a : -b);
        mat[k++] = (int8_t)ambig;
    }
    for (int j = 0; j < 5; ++j) mat[k++] = (int8_t)ambig;
}

void gbx_bsw_default_params(gbx_bsw_params *p)
{
    memset(p, 0, sizeof(*p));
    p->o_del = p->o_ins = 6; p->e_del = p->e_ins = 1;
    p->zdrop = 100; p->end_bonus = 5; p->w = 100;
    gbx_bsw_fill_scmat(1, 4, -1, p->mat);
}

size_t gbx_bsw_workspace_bytes(int64_t n) { return bsw_workspace_bytes(n); }

int gbx_bsw_extend_device(const gbx_bsw_params *p, int64_t n,
                          const uint8_t *d_ref, const uint8_t *d_qer,
                          const int64_t *d_idr, const int64_t *d_idq,
                          const int32_t *d_len1, const int32_t *d_len2,
                          const int32_t *d_h0, gbx_bsw_result *d_out,
                          void *d_work, size_t work_bytes, void *stream)
{
    if (!p || n < 0) { set_error("gbx_bsw_extend_device: bad argument"); return GBX_ERR_ARG; }
    if (n == 0) return GBX_OK;
    if (!d_ref || !d_qer || !d_idr || !d_idq || !d_len1 || !d_len2 || !d_h0 || !d_out || !d_work) {
        set_error("gbx_bsw_extend_device: null pointer");
        return GBX_ERR_ARG;
    }
    int rc = require_device();
    if (rc) return rc;
    return bsw_launch(p, n, d_ref, d_qer, d_idr, d_idq, d_len1, d_len2, d_h0, d_out, d_work, work_bytes,
                      (hipStream_t)stream);
}

// One device (the calling thread's current one).  `base` = index of pairs[0] in the caller's job (error texts only).
static int bsw_host_one(const gbx_bsw_params *p, int64_t n,
                        const uint8_t *ref, int64_t ref_bytes,
                        const uint8_t *qer, int64_t qer_bytes,
                        const int64_t *idr, const int64_t *idq,
                        const int32_t *len1, const int32_t *len2,
                        const int32_t *h0, gbx_bsw_result *out, int64_t base = 0)
{
    RoctxRange range_("gbx_bsw_extend_host");
    if (!p || n < 0 || ref_bytes < 0 || qer_bytes < 0) { set_error("gbx_bsw_extend_host: bad argument"); return GBX_ERR_ARG; }
    if (n == 0) return GBX_OK;
    if (!ref || !qer || !idr || !idq || !len1 || !len2 || !h0 || !out) {
        set_error("gbx_bsw_extend_host: null pointer");
        return GBX_ERR_ARG;
    }
    const bool trace = getenv("GBX_HOST_TRACE") != nullptr;     /* timeline of this call on stderr */
    const double t_begin = wall_s();
    // one pass over the pairs: validation, and per pipeline chunk the furthest arena byte its pairs need
    const std::vector<int64_t> cut = bsw_host_cuts(n);          // chunk c = pairs [cut[c], cut[c + 1])
    const int64_t n_chunks = (int64_t)cut.size() - 1;
    int64_t chunk = 0;                                          // the largest chunk
    for (int64_t c = 0; c < n_chunks; ++c) chunk = cut[(size_t)c + 1] - cut[(size_t)c] > chunk ? cut[(size_t)c + 1] - cut[(size_t)c] : chunk;
    std::vector<int64_t> need_r((size_t)n_chunks), need_q((size_t)n_chunks);
    // slices of 32 Ki pairs, a few threads when there are many; the lowest failing pair is reported
    const int64_t SL = 32768, n_slices = (n + SL - 1) / SL;
    std::vector<int64_t> slice_r((size_t)n_slices), slice_q((size_t)n_slices), slice_bad((size_t)n_slices, -1);
    std::vector<int64_t> slice_rows((size_t)n_slices, 0);   // pairs the lane kernels will not take (BswChunkPrep::rows_pairs)
    std::vector<int64_t> slice_cls((size_t)n_slices * 10, 0);      // ... and the others per lane launch (BswChunkPrep::class_pairs)
    // (counted only when the switch that uses the counts is on: the count is ten nanoseconds a pair on the call's critical path -
    // the first chunk's pairs are checked before anything is uploaded - and took that check from 0.25 to 0.86 ms on 'large')
    const bool count_classes = getenv("GBX_BSW_SKIP_EMPTY") && atoi(getenv("GBX_BSW_SKIP_EMPTY")) == 1;
    BswLaneRule rule = {0, 0, 0, 0};
    if (bsw_lane_rule(p, chunk, &rule) != GBX_OK) rule.on = 0;      // (bad parameters: the launch reports them)
    std::vector<int> slice_plain((size_t)n_slices, 0);      // longest query if every pair has 1 <= qlen <= 256, tlen >= 1 and a small h0 (bsw_launch_direct), else 0
    auto check_slice = [&](int64_t sl) {
        const int64_t a = sl * SL, b = a + SL < n ? a + SL : n;
        int64_t mr = 0, mq = 0, rows = 0;
        int64_t cls[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        bool plain = true;
        int maxq = 1;
        for (int64_t k = a; k < b; ++k) {
            const bool takes = bsw_lane_takes(rule, len2[k], len1[k], h0[k]);
            if (count_classes && takes) ++cls[bsw_lane_class(rule, len2[k], h0[k])];
            rows += !takes && len1[k] != 0 && len2[k] != 0;
            plain = plain && len2[k] >= 1 && len2[k] <= 256 && len1[k] >= 1 && h0[k] < 1000000;
            maxq = len2[k] > maxq ? len2[k] : maxq;
            const int64_t er = idr[k] + len1[k], eq = idq[k] + len2[k];
            if (len1[k] < 0 || len2[k] < 0 || idr[k] < 0 || idq[k] < 0 || er > ref_bytes || eq > qer_bytes ||
                len2[k] > GBX_BSW_MAX_QLEN || len1[k] > GBX_BSW_MAX_TLEN) {
                slice_bad[(size_t)sl] = k;
                return;
            }
            mr = er > mr ? er : mr; mq = eq > mq ? eq : mq;
        }
        slice_r[(size_t)sl] = mr; slice_q[(size_t)sl] = mq; slice_plain[(size_t)sl] = plain ? maxq : 0; slice_rows[(size_t)sl] = rows;
        for (int c = 0; c < 10; ++c) slice_cls[(size_t)sl * 10 + (size_t)c] = cls[c];
    };
    // slices [s0, s1): checked by a few threads; the lowest failing pair is reported
    auto validate = [&](int64_t s0, int64_t s1) -> int {
        const int64_t cnt = s1 - s0;
        const int vt = cnt >= 24 ? 12 : cnt >= 8 ? (int)(cnt / 2) : 1;      // 2.5 ns a pair and thread: 0.9 ms with 4 threads at 2 M pairs
        std::vector<Helper> th;
        for (int t = 1; t < vt; ++t) th.emplace_back([&, t] { for (int64_t sl = s0 + t; sl < s1; sl += vt) check_slice(sl); }, true);
        for (int64_t sl = s0; sl < s1; sl += vt) check_slice(sl);
        for (auto &x : th) x.join();
        for (int64_t sl = s0; sl < s1; ++sl) {
            const int64_t k = slice_bad[(size_t)sl];
            if (k < 0) continue;
            if (len1[k] >= 0 && len2[k] >= 0 && idr[k] >= 0 && idq[k] >= 0 && idr[k] + len1[k] <= ref_bytes &&
                idq[k] + len2[k] <= qer_bytes) {
                set_error("gbx_bsw_extend_host: pair %lld exceeds GBX_BSW_MAX_QLEN/TLEN", (long long)(base + k));
                return GBX_ERR_UNSUPPORTED;
            }
            set_error("gbx_bsw_extend_host: pair %lld lies outside the arenas", (long long)(base + k));
            return GBX_ERR_ARG;
        }
        return GBX_OK;
    };
    // A pipelined call checks the first chunk's pairs, starts its upload, and checks the rest while it is on its way (the
    // whole pass is 0.75 ms at 2 M pairs, and nothing else of the call can start before the first chunk is on the device).
    const int64_t s_first = n_chunks > 1 && (cut[1] + SL - 1) / SL < n_slices ? (cut[1] + SL - 1) / SL : n_slices;
    int rc = validate(0, s_first);
    if (rc) return rc;
    std::vector<int64_t> rows_pairs((size_t)n_chunks, -1), cls_pairs((size_t)n_chunks * 10, 0);
    auto chunk_needs = [&](int64_t c) {
        int64_t mr = 0, mq = 0, rows = 0;
        // chunks are multiples of 64 pairs, slices of 32768: a slice may straddle two chunks, which only makes
        // the earlier chunk wait for a few more bytes
        for (int64_t sl = cut[(size_t)c] / SL; sl < n_slices && sl * SL < cut[(size_t)c + 1]; ++sl) {
            mr = slice_r[(size_t)sl] > mr ? slice_r[(size_t)sl] : mr;
            mq = slice_q[(size_t)sl] > mq ? slice_q[(size_t)sl] : mq;
            rows += slice_rows[(size_t)sl];
            for (int k = 0; k < 10; ++k) cls_pairs[(size_t)c * 10 + (size_t)k] += slice_cls[(size_t)sl * 10 + (size_t)k];
        }
        need_r[(size_t)c] = mr; need_q[(size_t)c] = mq;
        // (an upper bound: a slice that straddles two chunks counts for both; a short last chunk may run without the lane path)
        BswLaneRule last = rule;
        const int64_t m = cut[(size_t)c + 1] - cut[(size_t)c];
        if (m != chunk && bsw_lane_rule(p, m, &last) != GBX_OK) last.on = 0;
        rows_pairs[(size_t)c] = rule.on && last.on ? rows : -1;
    };
    chunk_needs(0);
    if ((rc = require_device())) return rc;
    auto mark = [&](const char *what, int64_t k) { if (trace) fprintf(stderr, "[gbx host] %8.3f ms %s %lld\n", (wall_s() - t_begin) * 1e3, what, (long long)k); };
    mark("validated", s_first * SL < n ? s_first * SL : n);
    // Upload, compute and download are pipelined over chunks of pairs (host_pipeline.h).  Chunk k's bases and
    // index slices go up while earlier chunks run; the arenas are uploaded front to back up to the furthest
    // byte any pair seen so far needs (a running maximum), which is right for every offset layout and streams
    // perfectly for the usual monotone one.  The chunks are queued back to back without a barrier between them
    // (own workspace each; the launch records the events a chunk's download waits for), and their results come
    // back while later chunks run.  Chunks are multiples of 64 pairs.
    const size_t wb1 = (bsw_workspace_bytes(chunk) + 255) & ~(size_t)255;
    HostLane lane;
    if ((rc = lane.acquire())) return rc;
    Lane *L = lane.l;
    DevBuf dref(L), dqer(L), didr(L), didq(L), dl1(L), dl2(L), dh0(L), dout(L), dwork(L);
    if ((rc = dref.alloc((size_t)ref_bytes)) || (rc = dqer.alloc((size_t)qer_bytes)) ||
        (rc = didr.alloc(n * 8)) || (rc = didq.alloc(n * 8)) || (rc = dl1.alloc(n * 4)) ||
        (rc = dl2.alloc(n * 4)) || (rc = dh0.alloc(n * 4)) || (rc = dout.alloc(n * sizeof(gbx_bsw_result))) ||
        (rc = dwork.alloc(wb1 * (size_t)n_chunks)))
        return rc;
    mark("allocated", 0);
    HostPipe pipe(L, (size_t)ref_bytes + (size_t)qer_bytes + (size_t)n * 28, n_chunks > 1, BSW_HOST_WORKERS);
    if ((rc = pipe.prepare(n_chunks))) return rc;
    // Staged (large) calls send the bases two per byte: the upload workers pack them on their way into the pinned slabs
    // (host_pipeline.h: pack4), the device expands them into the byte arenas the kernels read (bsw_unpack4) - the
    // arenas are most of the upload (2 M pairs: 590 of 640 MB), and PCIe is the longest leg of the call.
    const bool pack_bases = pipe.staged && !(getenv("GBX_BSW_PACK") && atoi(getenv("GBX_BSW_PACK")) == 0);
    DevBuf dref_p(L), dqer_p(L);
    if (pack_bases && ((rc = dref_p.alloc((size_t)ref_bytes / 2 + 16)) || (rc = dqer_p.alloc((size_t)qer_bytes / 2 + 16)))) return rc;
    std::vector<int64_t> lo_r((size_t)n_chunks), hi_r((size_t)n_chunks), lo_q((size_t)n_chunks), hi_q((size_t)n_chunks);
    int64_t up_r = 0, up_q = 0;
    int64_t unp_r = 0, unp_q = 0;       // the byte arenas are expanded up to here (BswChunkPrep::unp_r)
    // Two upload stages per chunk: its index arrays (stage 2c: all the preparing passes read) and then its bases (2c + 1)
    pipe.upload_stages(2 * n_chunks);
    auto stage_chunk = [&](int64_t c) {
        const int64_t a = cut[(size_t)c], m = cut[(size_t)c + 1] - a;
        pipe.stage(2 * c, didr.as<int64_t>() + a, idr + a, m * 8);
        pipe.stage(2 * c, didq.as<int64_t>() + a, idq + a, m * 8);
        pipe.stage(2 * c, dl1.as<int32_t>() + a, len1 + a, m * 4);
        pipe.stage(2 * c, dl2.as<int32_t>() + a, len2 + a, m * 4);
        pipe.stage(2 * c, dh0.as<int32_t>() + a, h0 + a, m * 4);
        int64_t nr = need_r[(size_t)c] > up_r ? need_r[(size_t)c] : up_r;
        int64_t nq = need_q[(size_t)c] > up_q ? need_q[(size_t)c] : up_q;
        if (pack_bases) {
            // packed ranges start at even offsets: round the ends up to even while the arena allows it
            if ((nr & 1) && nr < ref_bytes) ++nr;
            if ((nq & 1) && nq < qer_bytes) ++nq;
            pipe.stage_pack4(2 * c + 1, dref_p.as<uint8_t>() + up_r / 2, ref + up_r, (size_t)(nr - up_r));
            pipe.stage_pack4(2 * c + 1, dqer_p.as<uint8_t>() + up_q / 2, qer + up_q, (size_t)(nq - up_q));
            lo_r[(size_t)c] = up_r; hi_r[(size_t)c] = nr; lo_q[(size_t)c] = up_q; hi_q[(size_t)c] = nq;
        } else {
            pipe.stage(2 * c + 1, dref.as<uint8_t>() + up_r, ref + up_r, (size_t)(nr - up_r));
            pipe.stage(2 * c + 1, dqer.as<uint8_t>() + up_q, qer + up_q, (size_t)(nq - up_q));
        }
        up_r = nr; up_q = nq;
    };
    stage_chunk(0);
    if (n_chunks > 1) pipe.keep_open();
    pipe.start();
    mark("pipeline started, chunks", n_chunks);
    if (n_chunks > 1) {
        if ((rc = validate(s_first, n_slices))) return pipe.finish(rc);
        for (int64_t c = 1; c < n_chunks; ++c) { chunk_needs(c); stage_chunk(c); }
        pipe.seal();
        mark("all chunks staged", n);
    }
    // small jobs of plain pairs: one kernel launch instead of the binning passes and the class kernels
    bool direct = n <= 16384 && n_chunks == 1 && !(getenv("GBX_BSW_DIRECT") && atoi(getenv("GBX_BSW_DIRECT")) == 0);
    int direct_q = 1;
    for (int64_t sl = 0; sl < n_slices && direct; ++sl) {
        direct = slice_plain[(size_t)sl] != 0;
        direct_q = slice_plain[(size_t)sl] > direct_q ? slice_plain[(size_t)sl] : direct_q;
    }
    if (direct) {
        if ((rc = pipe.wait_stage(0)) || (rc = pipe.wait_stage(1))) return pipe.finish(rc);
        if (pack_bases &&
            ((rc = bsw_unpack4(dref_p.as<uint8_t>(), dref.as<uint8_t>(), lo_r[0], hi_r[0], L->compute)) ||
             (rc = bsw_unpack4(dqer_p.as<uint8_t>(), dqer.as<uint8_t>(), lo_q[0], hi_q[0], L->compute))))
            return pipe.finish(rc);
        rc = bsw_launch_direct(p, n, direct_q, dref.as<uint8_t>(), dqer.as<uint8_t>(), didr.as<int64_t>(), didq.as<int64_t>(),
                               dl1.as<int32_t>(), dl2.as<int32_t>(), dh0.as<int32_t>(), dout.as<gbx_bsw_result>(), L->compute);
        if (!rc) { pipe.fetch(0, out, dout.p, n * sizeof(gbx_bsw_result)); rc = pipe.chunk_launched(0, 0); }
        return pipe.finish(rc);
    }
    for (int64_t c = 0; c < n_chunks; ++c) {
        const int64_t a = cut[(size_t)c], m = cut[(size_t)c + 1] - a;
        // pipelined calls: no barrier between the chunks, the launch records one event per kernel stream, and what
        // prepares a chunk (classify, the lane sort; unpacking, if it has row-kernel pairs) waits for its uploads only
        // (BswChunkPrep) - in two calls where the launch allows it: the preparing passes behind the index arrays, while
        // the bases are still on their way, the kernels behind the bases
        hipEvent_t *je = n_chunks > 1 ? pipe.join_events(c) : nullptr;
        BswChunkPrep prep = {nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0, rows_pairs[(size_t)c], 0, L->ev_pre, L->ev_aux};
        if (pack_bases) {
            prep.ref_packed = dref_p.as<uint8_t>(); prep.ref_bytes = dref.as<uint8_t>();
            prep.qer_packed = dqer_p.as<uint8_t>(); prep.qer_bytes = dqer.as<uint8_t>();
            prep.lo_r = lo_r[(size_t)c]; prep.hi_r = hi_r[(size_t)c]; prep.lo_q = lo_q[(size_t)c]; prep.hi_q = hi_q[(size_t)c];
            prep.unp_r = &unp_r; prep.unp_q = &unp_q;
        }
        if (count_classes && rows_pairs[(size_t)c] >= 0) {   // (counted with the rule this chunk's launch applies)
            prep.class_known = 1;
            for (int k = 0; k < 10; ++k) prep.class_pairs[k] = cls_pairs[(size_t)c * 10 + (size_t)k];
        }
        auto launch = [&]() {
            return bsw_launch(p, m, dref.as<uint8_t>(), dqer.as<uint8_t>(), didr.as<int64_t>() + a, didq.as<int64_t>() + a,
                              dl1.as<int32_t>() + a, dl2.as<int32_t>() + a, dh0.as<int32_t>() + a,
                              dout.as<gbx_bsw_result>() + a, (char *)dwork.p + wb1 * (size_t)c, wb1, L->compute, je, &prep);
        };
        static const bool split_off = getenv("GBX_BSW_SPLIT_PREP") && atoi(getenv("GBX_BSW_SPLIT_PREP")) == 0;
        bool split = je && pipe.stage_event() && !split_off;
        if (split) {
            if ((rc = pipe.wait_stage(2 * c, L->ev_part))) return pipe.finish(rc);
            mark("index arrays queued, chunk", c);
            prep.phase = 1; prep.uploaded = L->ev_part;
            rc = launch();
            if (rc == 1) { split = false; rc = GBX_OK; }
            if (rc) return pipe.finish(rc);
        } else if ((rc = pipe.wait_stage(2 * c))) return pipe.finish(rc);
        if ((rc = pipe.wait_stage(2 * c + 1))) return pipe.finish(rc);
        mark("uploads queued, chunk", c);
        prep.phase = split ? 2 : 0;
        prep.uploaded = je ? pipe.stage_event() : nullptr;
        rc = launch();
        if (!rc) {
            pipe.fetch(c, out + a, dout.as<gbx_bsw_result>() + a, m * sizeof(gbx_bsw_result));
            rc = pipe.chunk_launched(c, je ? Lane::JOIN_EVENTS : 0);
        }
        if (rc) return pipe.finish(rc);
    }
    mark("kernels queued", n_chunks);
    rc = pipe.finish();
    mark("results downloaded", 0);
    return rc;
}

// The host entry: one device, or the pairs cut into contiguous ranges of equal nominal cells (len1 x len2, the reference's
// own cell count, main_banded.cpp:183,323) over the devices of gbx_host_set_devices / GBX_GPUS - the reference's per-thread
// slices (main_banded.cpp:279-291) as per-device slices.  Each shard's bases are the byte range of the arenas its pairs
// span, sent by that device's own lane; results are written in place.
static int bsw_host_entry(const gbx_bsw_params *p, int64_t n,
                          const uint8_t *ref, int64_t ref_bytes,
                          const uint8_t *qer, int64_t qer_bytes,
                          const int64_t *idr, const int64_t *idq,
                          const int32_t *len1, const int32_t *len2,
                          const int32_t *h0, gbx_bsw_result *out)
{
    if (!host_multi_wanted() || !p || n <= 0 || !ref || !qer || !idr || !idq || !len1 || !len2 || !h0 || !out || ref_bytes < 0 || qer_bytes < 0)
        return bsw_host_one(p, n, ref, ref_bytes, qer, qer_bytes, idr, idq, len1, len2, h0, out);
    {   // argument errors come first and read as on one device: a job with a bad pair takes the one-device path, which names it
        const int T = host_workers(BSW_HOST_WORKERS);
        std::vector<char> bad((size_t)T, 0);
        parallel_ranges(n, T, [&](int t, int64_t lo, int64_t hi) {
            for (int64_t j = lo; j < hi; ++j)
                if (idr[j] < 0 || idq[j] < 0 || len1[j] < 0 || len2[j] < 0 || idr[j] + len1[j] > ref_bytes || idq[j] + len2[j] > qer_bytes ||
                    len2[j] > GBX_BSW_MAX_QLEN || len1[j] > GBX_BSW_MAX_TLEN) { bad[(size_t)t] = 1; return; }
        });
        for (char b : bad) if (b) return bsw_host_one(p, n, ref, ref_bytes, qer, qer_bytes, idr, idq, len1, len2, h0, out);
    }
    int map[MAX_HOST_DEVICES];
    const int n_dev = host_device_set(map);
    if (n_dev < 0) return n_dev;
    const int parts = shard_parts(n_dev, n, 131072);
    if (parts == 1) {
        DeviceGuard g;
        int rc = g.set(map[host_next_small_call_device(n_dev)]);
        return rc ? rc : bsw_host_one(p, n, ref, ref_bytes, qer, qer_bytes, idr, idq, len1, len2, h0, out);
    }
    const std::vector<int64_t> cuts = split_by_cost(n, parts, [&](int64_t k) {
        return len1[k] > 0 && len2[k] > 0 ? (double)len1[k] * (double)len2[k] : 0.0; });
    return run_on_devices(parts, map, "gbx_bsw_extend_host", [&](int k) -> int {
        const int64_t lo = cuts[(size_t)k], hi = cuts[(size_t)k + 1], m = hi - lo;
        if (m == 0) return GBX_OK;
        // the byte range of each arena this shard's pairs span (offsets re-based to its start)
        int64_t ar = ref_bytes, br = 0, aq = qer_bytes, bq = 0;
        for (int64_t j = lo; j < hi; ++j) {
            ar = idr[j] < ar ? idr[j] : ar; br = idr[j] + len1[j] > br ? idr[j] + len1[j] : br;
            aq = idq[j] < aq ? idq[j] : aq; bq = idq[j] + len2[j] > bq ? idq[j] + len2[j] : bq;
        }
        if (br < ar) br = ar;
        if (bq < aq) bq = aq;
        std::vector<int64_t> r2((size_t)m), q2((size_t)m);
        for (int64_t j = 0; j < m; ++j) { r2[(size_t)j] = idr[lo + j] - ar; q2[(size_t)j] = idq[lo + j] - aq; }
        return bsw_host_one(p, m, ref + ar, br - ar, qer + aq, bq - aq, r2.data(), q2.data(), len1 + lo, len2 + lo, h0 + lo, out + lo, lo);
    });
}

}  // extern "C"

// ---- small concurrent calls combined (host_combine.h).  The reference's driver calls getScores16 once per 512 pairs from
// every OpenMP thread (main_banded.cpp:279-291): the calls that are pending together become one job - the pairs of all
// requests end to end, their bases gathered into two compact arenas (4-byte aligned per pair, as the SeqPair entry does for
// the driver's strided slots) - and every caller gets its own slice of the results.
namespace {
struct BswReq : CombineReq {
    const gbx_bsw_params *p; int64_t n;
    const uint8_t *ref; int64_t ref_bytes; const uint8_t *qer; int64_t qer_bytes;
    const int64_t *idr, *idq; const int32_t *len1, *len2, *h0; gbx_bsw_result *out;
    int64_t cr, cq;                       // bytes of its pairs in the compact arenas
};
struct BswScratch {
    Scratch<uint8_t> ref, qer; Scratch<int64_t> idr, idq; Scratch<int32_t> l1, l2, h0; Scratch<gbx_bsw_result> out;
};
constexpr int64_t BSW_COMBINE_MAX_CALL = 65536, BSW_COMBINE_MAX_JOB = (int64_t)1 << 20;
}
namespace gbx { Combiner &combiner_bsw() { static Combiner *c = new Combiner(); return *c; } }

static void bsw_run_alone(BswReq *r)
{
    r->rc = bsw_host_entry(r->p, r->n, r->ref, r->ref_bytes, r->qer, r->qer_bytes, r->idr, r->idq, r->len1, r->len2, r->h0, r->out);
    if (r->rc) r->err = gbx_last_error();
}

static void bsw_run_combined(const std::vector<CombineReq *> &batch, int slot)
{
    if (batch.size() == 1) { bsw_run_alone((BswReq *)batch[0]); return; }
    static BswScratch *slots = new BswScratch[Combiner::MAX_LEADERS];      // one per leader in flight (Combiner::submit)
    BswScratch *S = slots + slot;
    const size_t nb = batch.size();
    std::vector<int64_t> p0(nb + 1, 0), r0(nb + 1, 0), q0(nb + 1, 0);
    for (size_t k = 0; k < nb; ++k) {
        const BswReq *r = (const BswReq *)batch[k];
        p0[k + 1] = p0[k] + r->n; r0[k + 1] = r0[k] + r->cr; q0[k + 1] = q0[k] + r->cq;
    }
    const int64_t N = p0[nb], R = r0[nb], Q = q0[nb];
    uint8_t *mref = S->ref.get((size_t)R + 16), *mqer = S->qer.get((size_t)Q + 16);
    int64_t *midr = S->idr.get((size_t)N), *midq = S->idq.get((size_t)N);
    int32_t *ml1 = S->l1.get((size_t)N), *ml2 = S->l2.get((size_t)N), *mh0 = S->h0.get((size_t)N);
    gbx_bsw_result *mout = S->out.get((size_t)N);
    combine_parallel((int64_t)nb, host_workers(), [&](int64_t k) {
        const BswReq *r = (const BswReq *)batch[(size_t)k];
        int64_t pr = r0[(size_t)k], pq = q0[(size_t)k];
        const int64_t a = p0[(size_t)k];
        for (int64_t j = 0; j < r->n; ++j) {
            memcpy(mref + pr, r->ref + r->idr[j], (size_t)r->len1[j]);
            memcpy(mqer + pq, r->qer + r->idq[j], (size_t)r->len2[j]);
            midr[a + j] = pr; midq[a + j] = pq;
            pr += (r->len1[j] + 3) & ~3; pq += (r->len2[j] + 3) & ~3;
        }
        memcpy(ml1 + a, r->len1, (size_t)r->n * 4); memcpy(ml2 + a, r->len2, (size_t)r->n * 4); memcpy(mh0 + a, r->h0, (size_t)r->n * 4);
    });
    const BswReq *lead = (const BswReq *)batch[0];
    const int rc = bsw_host_entry(lead->p, N, mref, R + 8, mqer, Q + 8, midr, midq, ml1, ml2, mh0, mout);
    if (rc) {                                       // redone one by one: every caller gets the status of its own call
        for (CombineReq *q : batch) bsw_run_alone((BswReq *)q);
        return;
    }
    combine_parallel((int64_t)nb, nb >= 8 ? 4 : 1, [&](int64_t k) {
        BswReq *r = (BswReq *)batch[(size_t)k];
        memcpy(r->out, mout + p0[(size_t)k], (size_t)r->n * sizeof(gbx_bsw_result));
        r->rc = GBX_OK;
    });
}

extern "C" {

int gbx_bsw_extend_host(const gbx_bsw_params *p, int64_t n,
                        const uint8_t *ref, int64_t ref_bytes,
                        const uint8_t *qer, int64_t qer_bytes,
                        const int64_t *idr, const int64_t *idq,
                        const int32_t *len1, const int32_t *len2,
                        const int32_t *h0, gbx_bsw_result *out)
{
    auto plain = [&] { return bsw_host_entry(p, n, ref, ref_bytes, qer, qer_bytes, idr, idq, len1, len2, h0, out); };
    if (!p || n <= 0 || n > BSW_COMBINE_MAX_CALL || !ref || !qer || !idr || !idq || !len1 || !len2 || !h0 || !out || ref_bytes < 0 || qer_bytes < 0 ||
        !combine_enabled() || profile_active())
        return plain();
    BswReq r;
    r.p = p; r.n = n; r.ref = ref; r.ref_bytes = ref_bytes; r.qer = qer; r.qer_bytes = qer_bytes;
    r.idr = idr; r.idq = idq; r.len1 = len1; r.len2 = len2; r.h0 = h0; r.out = out; r.units = n; r.cr = r.cq = 0;
    for (int64_t k = 0; k < n; ++k) {               // a call with a bad pair goes its own way: its error names the pair
        if (len1[k] < 0 || len2[k] < 0 || idr[k] < 0 || idq[k] < 0 || idr[k] + len1[k] > ref_bytes || idq[k] + len2[k] > qer_bytes ||
            len2[k] > GBX_BSW_MAX_QLEN || len1[k] > GBX_BSW_MAX_TLEN)
            return plain();
        r.cr += (len1[k] + 3) & ~3; r.cq += (len2[k] + 3) & ~3;
    }
    if (hipGetDevice(&r.dev) != hipSuccess) { (void)hipGetLastError(); return plain(); }
    return combiner_bsw().submit(&r, BSW_COMBINE_MAX_JOB, Combiner::max_leaders(1),
        [](const CombineReq *a, const CombineReq *b) { return memcmp(((const BswReq *)a)->p, ((const BswReq *)b)->p, offsetof(gbx_bsw_params, pad_)) == 0; },
        bsw_run_combined);
}

int gbx_bsw_extend_seqpairs(const gbx_bsw_params *p, gbx_seqpair *pairs, int64_t n,
                            const uint8_t *ref, int64_t ref_bytes,
                            const uint8_t *qer, int64_t qer_bytes)
{
    RoctxRange range_("gbx_bsw_extend_seqpairs");
    if (!p || n < 0) { set_error("gbx_bsw_extend_seqpairs: bad argument"); return GBX_ERR_ARG; }
    if (n == 0) return GBX_OK;
    if (!pairs) { set_error("gbx_bsw_extend_seqpairs: null pointer"); return GBX_ERR_ARG; }
    if (!ref || !qer || ref_bytes < 0 || qer_bytes < 0) { set_error("gbx_bsw_extend_seqpairs: bad arena"); return GBX_ERR_ARG; }
    // The reference's driver gives every pair a fixed-stride slot in the two buffers (MAX_SEQ_LEN_REF / _QER bytes,
    // main_banded.cpp:56-58,160-172), so the arenas are mostly holes: 2 M pairs span 4.6 GB for 0.6 GB of bases.
    // The flat arrays are extracted with a few threads, and when the layout is that sparse the bases are gathered
    // into packed arenas first instead of sending the holes over PCIe.
    const int T = host_workers(BSW_HOST_WORKERS);
    std::vector<int64_t> idr(n), idq(n);
    std::vector<int32_t> l1(n), l2(n), h0(n);
    std::vector<gbx_bsw_result> out(n);
    std::vector<int64_t> part_r((size_t)T + 1, 0), part_q((size_t)T + 1, 0), part_bad((size_t)T, -1);
    parallel_ranges(n, T, [&](int t, int64_t lo, int64_t hi) {
        int64_t sr = 0, sq = 0;
        for (int64_t k = lo; k < hi; ++k) {
            const gbx_seqpair &sp = pairs[k];
            if (sp.len1 < 0 || sp.len2 < 0 || sp.idr < 0 || sp.idq < 0 || sp.idr + sp.len1 > ref_bytes ||
                sp.idq + sp.len2 > qer_bytes) { part_bad[(size_t)t] = k; return; }
            idr[k] = sp.idr; idq[k] = sp.idq; l1[k] = sp.len1; l2[k] = sp.len2; h0[k] = sp.h0;
            sr += (sp.len1 + 3) & ~3; sq += (sp.len2 + 3) & ~3;
        }
        part_r[(size_t)t + 1] = sr; part_q[(size_t)t + 1] = sq;
    });
    for (int t = 0; t < T; ++t)
        if (part_bad[(size_t)t] >= 0) {
            set_error("gbx_bsw_extend_seqpairs: pair %lld lies outside the arenas", (long long)part_bad[(size_t)t]);
            return GBX_ERR_ARG;
        }
    for (int t = 0; t < T; ++t) { part_r[(size_t)t + 1] += part_r[(size_t)t]; part_q[(size_t)t + 1] += part_q[(size_t)t]; }
    const int64_t packed_r = part_r[(size_t)T], packed_q = part_q[(size_t)T];
    std::vector<uint8_t> cref, cqer;
    // (small calls too: the driver's 512-pair batch spans 1.1 MB of slots for 180 KB of bases, and a call that size is mostly the
    // copy into the pinned slab and the DMA)
    const bool sparse = (ref_bytes + qer_bytes) > 2 * (packed_r + packed_q) + ((int64_t)64 << 10);
    if (sparse) {
        cref.resize((size_t)packed_r + 8); cqer.resize((size_t)packed_q + 8);
        // same thread ranges as above, so every thread knows where its pairs start in the packed arenas
        parallel_ranges(n, T, [&](int t, int64_t lo, int64_t hi) {
            int64_t pr = part_r[(size_t)t], pq = part_q[(size_t)t];
            for (int64_t k = lo; k < hi; ++k) {
                memcpy(&cref[(size_t)pr], ref + idr[k], (size_t)l1[k]);
                memcpy(&cqer[(size_t)pq], qer + idq[k], (size_t)l2[k]);
                idr[k] = pr; idq[k] = pq;
                pr += (l1[k] + 3) & ~3; pq += (l2[k] + 3) & ~3;
            }
        });
        ref = cref.data(); ref_bytes = packed_r + 8; qer = cqer.data(); qer_bytes = packed_q + 8;
    }
    int rc = gbx_bsw_extend_host(p, n, ref, ref_bytes, qer, qer_bytes, idr.data(), idq.data(), l1.data(),
                                 l2.data(), h0.data(), out.data());
    if (rc) return rc;
    parallel_ranges(n, T, [&](int, int64_t lo, int64_t hi) {
        for (int64_t k = lo; k < hi; ++k) {
            pairs[k].score = out[k].score; pairs[k].tle = out[k].tle; pairs[k].gtle = out[k].gtle;
            pairs[k].qle = out[k].qle; pairs[k].gscore = out[k].gscore; pairs[k].max_off = out[k].max_off;
        }
    });
    return GBX_OK;
}


}  // extern "C"
