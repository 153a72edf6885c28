// capi_abea.hip — abea entries of the C-ABI (include/gbx.h).
#include "capi_common.h"

using namespace gbx;

extern "C" {

/* -------------------------------------------------------------------- abea */
int gbx_abea_plan_host(int64_t n_reads, const int32_t *seq_len, const int64_t *event_off,
                       int64_t *band_off, int32_t *order, double *lp)
{
    if (n_reads < 0 || (n_reads > 0 && (!seq_len || !event_off || !band_off || !order || !lp))) {
        set_error("gbx_abea_plan_host: bad argument");
        return GBX_ERR_ARG;
    }
    band_off[0] = 0;
    std::vector<std::pair<int64_t, int32_t>> by_len((size_t)n_reads);
    for (int64_t r = 0; r < n_reads; ++r) {
        const int64_t n_events = event_off[r + 1] - event_off[r], n_kmers = (int64_t)seq_len[r] - GBX_ABEA_KMER + 1;
        if (n_events < 1 || n_kmers < 1) {
            set_error("gbx_abea_plan_host: read %lld needs at least one event and %d bases", (long long)r, GBX_ABEA_KMER);
            return GBX_ERR_ARG;
        }
        if (n_events > 0x3fffffff || n_kmers > 0x3fffffff) { set_error("gbx_abea_plan_host: read %lld is too long", (long long)r); return GBX_ERR_UNSUPPORTED; }
        const int64_t n_bands = (n_events + 1) + (n_kmers + 1);                       /* align.c:209-211 */
        band_off[r + 1] = band_off[r] + n_bands;
        by_len[(size_t)r] = std::make_pair(-n_bands, (int32_t)r);
        /* transition penalties, align.c:195-204: the host C library's log / exp, as the reference */
        const double events_per_kmer = (double)n_events / (double)n_kmers;
        const double p_stay = 1 - (1 / (events_per_kmer + 1));
        const double epsilon = 1e-10;
        const double lp_skip = log(epsilon), lp_stay = log(p_stay);
        lp[2 * r] = lp_stay;
        lp[2 * r + 1] = log(1.0 - exp(lp_skip) - exp(lp_stay));
    }
    std::sort(by_len.begin(), by_len.end());                                          /* longest first; ties in input order */
    for (int64_t r = 0; r < n_reads; ++r) order[r] = by_len[(size_t)r].second;
    return GBX_OK;
}

size_t gbx_abea_workspace_bytes(int64_t n_reads, int64_t n_kmers_total, int64_t n_bands_total)
{
    return abea_workspace_bytes(n_reads, n_kmers_total, n_bands_total);
}

int gbx_abea_cells(const void *d_work, int64_t *cells, void *stream)
{
    if (!d_work || !cells) { set_error("gbx_abea_cells: null pointer"); return GBX_ERR_ARG; }
    return abea_read_cells(d_work, cells, (hipStream_t)stream);
}

int gbx_abea_align_device(int64_t n_reads, const int64_t *d_seq_off, const int32_t *d_seq_len, const char *d_seq_arena,
                          const int64_t *d_event_off, const float *d_event_mean, const gbx_abea_model *d_models,
                          const float *d_scale, const float *d_shift, const int64_t *d_band_off, const int32_t *d_order,
                          const double *d_lp, int64_t n_kmers_total, int64_t n_bands_total,
                          gbx_abea_pair *d_out, int32_t *d_n_pairs, void *d_work, size_t work_bytes, void *stream)
{
    if (n_reads < 0 || n_kmers_total < 0 || n_bands_total < 0) { set_error("gbx_abea_align_device: bad argument"); return GBX_ERR_ARG; }
    if (n_reads == 0) return GBX_OK;
    if (!d_seq_off || !d_seq_len || !d_seq_arena || !d_event_off || !d_event_mean || !d_models || !d_scale || !d_shift ||
        !d_band_off || !d_order || !d_lp || !d_out || !d_n_pairs || !d_work) {
        set_error("gbx_abea_align_device: null pointer");
        return GBX_ERR_ARG;
    }
    int rc = require_device();
    if (rc) return rc;
    return abea_launch(n_reads, d_seq_off, d_seq_len, d_seq_arena, d_event_off, d_event_mean, d_models, d_scale, d_shift,
                       d_band_off, d_order, d_lp, n_kmers_total, n_bands_total, d_out, d_n_pairs, d_work, work_bytes,
                       (hipStream_t)stream);
}

// One device (the calling thread's current one).  `base` = index of read 0 in the caller's job (error texts only).
static int abea_host_one(int64_t n_reads, const int64_t *seq_off, const int32_t *seq_len, const char *seq_arena,
                         int64_t seq_bytes, const int64_t *event_off, const gbx_abea_event *events,
                         const gbx_abea_model *models, const float *scale, const float *shift,
                         gbx_abea_pair *out, int32_t *n_pairs, int64_t base = 0)
{
    RoctxRange range_("gbx_abea_align_host");
    const bool trace = getenv("GBX_HOST_TRACE") != nullptr;
    const double t_begin = wall_s();
    auto mark = [&](const char *what) { if (trace) fprintf(stderr, "[gbx abea host] %9.3f ms %s\n", (wall_s() - t_begin) * 1e3, what); };
    if (n_reads < 0 || seq_bytes < 0) { set_error("gbx_abea_align_host: bad argument"); return GBX_ERR_ARG; }
    if (n_reads == 0) return GBX_OK;
    if (!seq_off || !seq_len || !seq_arena || !event_off || !events || !models || !scale || !shift || !out || !n_pairs) {
        set_error("gbx_abea_align_host: null pointer");
        return GBX_ERR_ARG;
    }
    for (int64_t r = 0; r < n_reads; ++r) {
        if (seq_off[r] < 0 || seq_len[r] < 0 || seq_off[r] + seq_len[r] > seq_bytes) {
            set_error("gbx_abea_align_host: read %lld lies outside the arena", (long long)(base + r));
            return GBX_ERR_ARG;
        }
        if (event_off[r + 1] < event_off[r] || event_off[r] < 0) { set_error("gbx_abea_align_host: event_off not monotone at read %lld", (long long)(base + r)); return GBX_ERR_ARG; }
    }
    std::vector<int64_t> band_off((size_t)n_reads + 1);
    std::vector<int32_t> order((size_t)n_reads);
    std::vector<double> lp((size_t)n_reads * 2);
    int rc = gbx_abea_plan_host(n_reads, seq_len, event_off, band_off.data(), order.data(), lp.data());
    if (rc) return rc;
    if ((rc = require_device())) return rc;
    const int64_t e0 = event_off[0], n_ev = event_off[n_reads] - e0;
    int64_t n_kmers_total = 0;
    for (int64_t r = 0; r < n_reads; ++r) n_kmers_total += (int64_t)seq_len[r] - GBX_ABEA_KMER + 1;
    // only the means of the events are read (align.c:125): the upload workers gather them from the 24-byte records
    // straight into the pinned slabs (no compact host copy)
    mark("planned");
    std::vector<int64_t> eoff((size_t)n_reads + 1);
    for (int64_t r = 0; r <= n_reads; ++r) eoff[(size_t)r] = event_off[r] - e0;
    HostLane lane;
    if ((rc = lane.acquire())) return rc;
    Lane *L = lane.l;
    const size_t wb = abea_workspace_bytes(n_reads, n_kmers_total, band_off[(size_t)n_reads]);
    DevBuf dso(L), dsl(L), dsq(L), deo(L), dem(L), dmo(L), dsc(L), dsh(L), dbo(L), dor(L), dlp(L), dout(L), dnp(L), dw(L);
    if ((rc = dso.alloc(n_reads * 8)) || (rc = dsl.alloc(n_reads * 4)) || (rc = dsq.alloc((size_t)seq_bytes)) ||
        (rc = deo.alloc((n_reads + 1) * 8)) || (rc = dem.alloc((size_t)n_ev * 4 + 16)) || (rc = dmo.alloc(GBX_ABEA_NMODEL * sizeof(gbx_abea_model))) ||
        (rc = dsc.alloc(n_reads * 4)) || (rc = dsh.alloc(n_reads * 4)) || (rc = dbo.alloc((n_reads + 1) * 8)) || (rc = dor.alloc(n_reads * 4)) ||
        (rc = dlp.alloc(n_reads * 16)) || (rc = dout.alloc((size_t)n_ev * 2 * sizeof(gbx_abea_pair) + 16)) || (rc = dnp.alloc(n_reads * 4)) || (rc = dw.alloc(wb)))
        return rc;
    HostPipe pipe(L, (size_t)seq_bytes + (size_t)n_ev * 4 + (size_t)n_reads * 60, false);
    if ((rc = pipe.prepare(1))) return rc;
    pipe.stage(0, dso.p, seq_off, n_reads * 8); pipe.stage(0, dsl.p, seq_len, n_reads * 4); pipe.stage(0, dsq.p, seq_arena, (size_t)seq_bytes);
    pipe.stage(0, deo.p, eoff.data(), (n_reads + 1) * 8); if (n_ev) pipe.stage_field4(0, dem.p, &events[e0].mean, (size_t)n_ev, (int)sizeof(gbx_abea_event));
    pipe.stage(0, dmo.p, models, GBX_ABEA_NMODEL * sizeof(gbx_abea_model));
    pipe.stage(0, dsc.p, scale, n_reads * 4); pipe.stage(0, dsh.p, shift, n_reads * 4);
    pipe.stage(0, dbo.p, band_off.data(), (n_reads + 1) * 8); pipe.stage(0, dor.p, order.data(), n_reads * 4);
    pipe.stage(0, dlp.p, lp.data(), n_reads * 16);
    mark("device buffers ready");
    pipe.start();
    if ((rc = pipe.wait_stage(0))) return pipe.finish(rc);
    mark("uploads queued");
    rc = abea_launch(n_reads, dso.as<int64_t>(), dsl.as<int32_t>(), dsq.as<char>(), deo.as<int64_t>(), dem.as<float>(),
                     dmo.as<gbx_abea_model>(), dsc.as<float>(), dsh.as<float>(), dbo.as<int64_t>(), dor.as<int32_t>(), dlp.as<double>(),
                     n_kmers_total, band_off[(size_t)n_reads], dout.as<gbx_abea_pair>(), dnp.as<int32_t>(), dw.p, wb, lane.l->compute);
    if (rc) return pipe.finish(rc);
    // the caller's pair array is indexed by its own event_off (out + 2*event_off[r])
    std::vector<HostPipe::Seg> segs;
    std::vector<int64_t> prefix;
    DevBuf dpre(L);
    if (pipe.staged) {
        // large calls: half of the 2 x n_events slots are slack, so the counts come first (the calling thread waits for the
        // kernel here instead of in finish()), the pairs are packed on the device and their download is scattered to
        // the reads' places by the copy-out threads
        // (an asynchronous fault of abea_kernel surfaces at this synchronize: every error leaves through pipe.finish(),
        // which cancels the downloader thread - a bare return here would leave it waiting for chunk 0 for ever)
        hipError_t he = hipMemcpyAsync(n_pairs, dnp.p, (size_t)n_reads * 4, hipMemcpyDeviceToHost, lane.l->compute);
        if (he == hipSuccess) he = hipStreamSynchronize(lane.l->compute);
        if (he != hipSuccess) return pipe.finish(hip_fail(he, "gbx_abea_align_host: kernel / pair counts"));
        mark("kernel done, counts on the host");
        prefix.resize((size_t)n_reads + 1);
        segs.resize((size_t)n_reads);
        int64_t tot = 0;
        for (int64_t r = 0; r < n_reads; ++r) {
            prefix[(size_t)r] = tot;
            const int64_t np = n_pairs[r] > 0 ? n_pairs[r] : 0;
            segs[(size_t)r] = HostPipe::Seg{(char *)(out + 2 * event_off[r]), (size_t)np * sizeof(gbx_abea_pair)};
            tot += np;
        }
        prefix[(size_t)n_reads] = tot;
        if ((rc = dpre.alloc((size_t)(n_reads + 1) * 8))) return pipe.finish(rc);
        he = hipMemcpyAsync(dpre.p, prefix.data(), (size_t)(n_reads + 1) * 8, hipMemcpyHostToDevice, lane.l->compute);
        if (he != hipSuccess) return pipe.finish(hip_fail(he, "gbx_abea_align_host: pair prefix upload"));
        gbx_abea_pair *packed = nullptr;
        if ((rc = abea_pack_pairs(n_reads, deo.as<int64_t>(), dout.as<gbx_abea_pair>(), dnp.as<int32_t>(), dpre.as<int64_t>(), dw.p,
                                  n_kmers_total, &packed, lane.l->compute)))
            return pipe.finish(rc);
        if (tot) pipe.fetch_scatter(0, packed, (size_t)tot * sizeof(gbx_abea_pair), &segs);
    } else {
        pipe.fetch(0, out + 2 * e0, dout.p, (size_t)n_ev * 2 * sizeof(gbx_abea_pair));
        pipe.fetch(0, n_pairs, dnp.p, n_reads * 4);
    }
    if ((rc = pipe.chunk_launched(0))) return pipe.finish(rc);
    mark("kernel queued");
    rc = pipe.finish();
    mark("results downloaded");
    return rc;
}


// The host entry: one device, or the reads cut into contiguous ranges of equal band counts (events + k-mers + 2 bands of
// 100 cells each, align.c:209-211) over the devices of gbx_host_set_devices / GBX_GPUS - align_db's loop over the reads of
// a batch (f5c.c:1350-1370) as a loop over devices.  The one-device path indexes events and pairs by the caller's absolute
// event_off, so a shard is the same arrays entered at read `lo`; only the bases are cut down to the shard's byte range.
int gbx_abea_align_host(int64_t n_reads, const int64_t *seq_off, const int32_t *seq_len, const char *seq_arena,
                        int64_t seq_bytes, const int64_t *event_off, const gbx_abea_event *events,
                        const gbx_abea_model *models, const float *scale, const float *shift,
                        gbx_abea_pair *out, int32_t *n_pairs)
{
    auto one = [&]() { return abea_host_one(n_reads, seq_off, seq_len, seq_arena, seq_bytes, event_off, events, models, scale, shift, out, n_pairs); };
    if (!host_multi_wanted() || n_reads <= 0 || seq_bytes < 0 || !seq_off || !seq_len || !seq_arena || !event_off || !events || !models ||
        !scale || !shift || !out || !n_pairs)
        return one();
    for (int64_t r = 0; r < n_reads; ++r)
        if (seq_off[r] < 0 || seq_len[r] < 0 || seq_off[r] + seq_len[r] > seq_bytes || event_off[r + 1] < event_off[r] || event_off[r] < 0 ||
            event_off[r + 1] - event_off[r] < 1 || seq_len[r] < GBX_ABEA_KMER)
            return one();
    int map[MAX_HOST_DEVICES];
    const int n_dev = host_device_set(map);
    if (n_dev < 0) return n_dev;
    const int parts = shard_parts(n_dev, n_reads, 128);
    if (parts == 1) {
        DeviceGuard g;
        int rc = g.set(map[host_next_small_call_device(n_dev)]);
        return rc ? rc : one();
    }
    const std::vector<int64_t> cuts = split_by_cost(n_reads, parts, [&](int64_t r) {
        return (double)((event_off[r + 1] - event_off[r] + 1) + ((int64_t)seq_len[r] - GBX_ABEA_KMER + 2)); });
    return run_on_devices(parts, map, "gbx_abea_align_host", [&](int k) -> int {
        const int64_t lo = cuts[(size_t)k], hi = cuts[(size_t)k + 1], m = hi - lo;
        if (m == 0) return GBX_OK;
        int64_t a0 = seq_bytes, a1 = 0;
        for (int64_t r = lo; r < hi; ++r) { a0 = seq_off[r] < a0 ? seq_off[r] : a0; a1 = seq_off[r] + seq_len[r] > a1 ? seq_off[r] + seq_len[r] : a1; }
        std::vector<int64_t> so((size_t)m);
        for (int64_t r = 0; r < m; ++r) so[(size_t)r] = seq_off[lo + r] - a0;
        return abea_host_one(m, so.data(), seq_len + lo, seq_arena + a0, a1 - a0, event_off + lo, events, models, scale + lo, shift + lo, out,
                             n_pairs + lo, lo);
    });
}

}  // extern "C"
