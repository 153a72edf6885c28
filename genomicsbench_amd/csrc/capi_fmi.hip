// capi_fmi.hip — fmi entries of the C-ABI (include/gbx.h).
#include "capi_common.h"

using namespace gbx;

extern "C" {

// -------------------------------------------------------------------- fmi
void gbx_fmi_default_params(gbx_fmi_params *p, int32_t min_seed_len)
{
    if (!p) return;
    p->min_seed_len = min_seed_len;                                 // fmi.cpp:135
    p->split_width = 10;                                            // :138
    p->max_mem_intv = 20;                                           // :139
    p->split_len = (int32_t)(min_seed_len * 1.5 + .499);            // :140,178
}

size_t gbx_fmi_index_bytes(int64_t ref_seq_len) { return fmi_index_bytes(ref_seq_len); }

int gbx_fmi_index_build(const gbx_fmi_index *idx, void *d_index, size_t index_bytes, void *stream)
{
    if (!idx || !idx->cp_occ || !d_index) { set_error("gbx_fmi_index_build: null pointer"); return GBX_ERR_ARG; }
    int rc = require_device();
    if (rc) return rc;
    return fmi_index_build(idx, d_index, index_bytes, (hipStream_t)stream);
}

size_t gbx_fmi_workspace_bytes(int64_t n_reads, int32_t max_read_len, int32_t min_seed_len)
{
    return fmi_workspace_bytes(n_reads, max_read_len, min_seed_len);
}

static int fmi_check(const gbx_fmi_index *idx, const gbx_fmi_params *p, const char *who)
{
    if (!idx || !p) { set_error("%s: null pointer", who); return GBX_ERR_ARG; }
    if (p->min_seed_len < 1 || p->split_width < 0 || p->max_mem_intv < 0) { set_error("%s: bad parameters", who); return GBX_ERR_ARG; }
    if (idx->ref_seq_len < 2 || idx->count[0] != 1 || idx->count[4] != idx->ref_seq_len || idx->sentinel_index < 0 ||
        idx->sentinel_index >= idx->ref_seq_len) {
        set_error("%s: inconsistent index (count[0] must be 1, count[4] the reference length incl. the sentinel)", who);
        return GBX_ERR_ARG;
    }
    for (int c = 0; c < 4; ++c)
        if (idx->count[c] > idx->count[c + 1]) { set_error("%s: count[] not monotone", who); return GBX_ERR_ARG; }
    return GBX_OK;
}

int gbx_fmi_smem_device(const gbx_fmi_index *idx, const void *d_index, const gbx_fmi_params *p, int64_t n_reads,
                        int32_t max_read_len, const uint8_t *d_enc, const int64_t *d_read_off, const int32_t *d_read_len,
                        gbx_fmi_smem *d_out, int64_t out_cap, int64_t *d_smem_off, int64_t *d_n_out,
                        void *d_work, size_t work_bytes, void *stream)
{
    int rc = fmi_check(idx, p, "gbx_fmi_smem_device");
    if (rc) return rc;
    if (n_reads < 0 || out_cap < 0) { set_error("gbx_fmi_smem_device: bad argument"); return GBX_ERR_ARG; }
    if (!d_index || !d_smem_off || !d_n_out || !d_work || (n_reads > 0 && (!d_enc || !d_read_off || !d_read_len)) || (out_cap > 0 && !d_out)) {
        set_error("gbx_fmi_smem_device: null pointer");
        return GBX_ERR_ARG;
    }
    if ((rc = require_device())) return rc;
    return fmi_launch(idx, d_index, p, n_reads, max_read_len, d_enc, d_read_off, d_read_len, d_out, out_cap, d_smem_off, d_n_out,
                      d_work, work_bytes, (hipStream_t)stream);
}

int gbx_fmi_overflow(const void *d_work, int64_t *worst, void *stream)
{
    if (!d_work || !worst) { set_error("gbx_fmi_overflow: null pointer"); return GBX_ERR_ARG; }
    return fmi_read_overflow(d_work, worst, (hipStream_t)stream);
}

int gbx_fmi_extensions(const void *d_work, int64_t *ext, void *stream)
{
    if (!d_work || !ext) { set_error("gbx_fmi_extensions: null pointer"); return GBX_ERR_ARG; }
    return fmi_read_extensions(d_work, ext, (hipStream_t)stream);
}

// The host entry keeps the device copy of an index between calls (a reference-side caller hands over the same
// FMI_search tables for every batch of reads, fmi.cpp:218), one per device.  An entry is found by the index's CONTENT -
// its scalars and a fingerprint of 256 checkpoints spread over the table - not by the caller's address: a buffer that
// was freed and reused for another index of the same length is a different index.  Entries are reference-counted while a
// call uses them (gbx_fmi_host_release leaves those alone) and at most four idle ones are kept per device.
namespace {
struct FmiCached { int dev; int64_t len, sentinel, count[5]; uint64_t fp; void *d_index; size_t bytes; int users; uint64_t last_use; bool building; };
std::mutex g_fmi_mu;
std::condition_variable g_fmi_cv;          // an entry under construction (building) has been finished or given up
std::vector<FmiCached> g_fmi_cache;
uint64_t g_fmi_clock = 0;

uint64_t fmi_fingerprint(const gbx_fmi_index *idx)
{
    const int64_t ncp = (idx->ref_seq_len >> 6) + 1;
    uint64_t h = 1469598103934665603ull;
    auto mix = [&](const void *p, size_t n) { const unsigned char *b = (const unsigned char *)p; for (size_t k = 0; k < n; ++k) { h ^= b[k]; h *= 1099511628211ull; } };
    // one checkpoint in every 4096 (at least 256): a table edited in place between two calls is caught unless the edit misses
    // every sampled line - mutating an index that has been handed to the library is not supported (gbx_fmi_host_release forgets it)
    const int64_t samples = ncp < 256 ? ncp : std::max<int64_t>(256, ncp >> 12);
    for (int64_t k = 0; k < samples; ++k) mix(&idx->cp_occ[(size_t)(k * (ncp - 1) / (samples > 1 ? samples - 1 : 1))], sizeof(gbx_fmi_cp_occ));
    return h;
}
struct FmiUse {                       // holds a cache entry for the duration of a call
    void *d_index = nullptr;
    ~FmiUse()
    {
        if (!d_index) return;
        std::lock_guard<std::mutex> lk(g_fmi_mu);
        for (FmiCached &c : g_fmi_cache) if (c.d_index == d_index && c.users > 0) { --c.users; break; }
    }
};
}

// One device (the calling thread's current one).  `base` = index of read 0 in the caller's job (error texts only).
static int fmi_host_one(const gbx_fmi_index *idx, const gbx_fmi_params *p, int64_t n_reads, const uint8_t *enc, int64_t enc_bytes,
                        const int64_t *read_off, const int32_t *read_len, gbx_fmi_smem *out, int64_t out_cap,
                        int64_t *smem_off, int64_t *n_out, int64_t base = 0)
{
    RoctxRange range_("gbx_fmi_smem_host");
    int rc = fmi_check(idx, p, "gbx_fmi_smem_host");
    if (rc) return rc;
    if (n_reads < 0 || out_cap < 0 || enc_bytes < 0) { set_error("gbx_fmi_smem_host: bad argument"); return GBX_ERR_ARG; }
    if (!idx->cp_occ || !n_out || (n_reads > 0 && (!enc || !read_off || !read_len)) || (out_cap > 0 && !out)) {
        set_error("gbx_fmi_smem_host: null pointer");
        return GBX_ERR_ARG;
    }
    int32_t max_len = 0;
    for (int64_t r = 0; r < n_reads; ++r) {
        if (read_len[r] < 0 || read_off[r] < 0 || read_off[r] + read_len[r] > enc_bytes) {
            set_error("gbx_fmi_smem_host: read %lld lies outside the base buffer", (long long)(base + r));
            return GBX_ERR_ARG;
        }
        if (read_len[r] > max_len) max_len = read_len[r];
    }
    if ((rc = require_device())) return rc;
    int dev = 0;
    GBX_HIP(hipGetDevice(&dev));
    HostLane lane;
    if ((rc = lane.acquire())) return rc;
    Lane *L = lane.l;
    hipStream_t s = L->compute;
    // the device index: cached, or uploaded in the reference's layout and re-laid on the device
    FmiUse use;
    void *d_index = nullptr;
    {
        const uint64_t fp = fmi_fingerprint(idx);
        auto same = [&](const FmiCached &c) {
            return c.dev == dev && c.len == idx->ref_seq_len && c.sentinel == idx->sentinel_index && c.fp == fp && !memcmp(c.count, idx->count, sizeof(c.count));
        };
        bool build = false;
        {
            std::unique_lock<std::mutex> lk(g_fmi_mu);
            for (;;) {
                FmiCached *hit = nullptr;
                for (FmiCached &c : g_fmi_cache) if (same(c)) hit = &c;
                if (hit && hit->building) { g_fmi_cv.wait(lk); continue; }        // another caller is uploading this very index: wait for it
                if (hit) { d_index = hit->d_index; ++hit->users; hit->last_use = ++g_fmi_clock; }
                break;
            }
            if (!d_index) {
                // at most four idle entries per device: the least recently used goes first
                for (;;) {
                    int idle = 0, victim = -1;
                    for (size_t k = 0; k < g_fmi_cache.size(); ++k)
                        if (g_fmi_cache[k].dev == dev && g_fmi_cache[k].users == 0 && !g_fmi_cache[k].building) {
                            ++idle;
                            if (victim < 0 || g_fmi_cache[k].last_use < g_fmi_cache[(size_t)victim].last_use) victim = (int)k;
                        }
                    if (idle < 4) break;
                    (void)hipFree(g_fmi_cache[(size_t)victim].d_index);
                    g_fmi_cache.erase(g_fmi_cache.begin() + victim);
                }
                // a place-holder under the lock; the upload and the re-layout (up to a gigabyte over PCIe) run outside it, so that
                // the shards of a multi-device call build their copies side by side and gbx_fmi_host_release never waits for one
                FmiCached c{dev, idx->ref_seq_len, idx->sentinel_index, {0, 0, 0, 0, 0}, fp, nullptr, 0, 1, ++g_fmi_clock, true};
                memcpy(c.count, idx->count, sizeof(c.count));
                g_fmi_cache.push_back(c);
                build = true;
            }
        }
        if (build) {
            const size_t bytes = fmi_index_bytes(idx->ref_seq_len);
            void *d_src = nullptr;
            hipError_t e = hipMalloc(&d_index, bytes);
            if (e != hipSuccess) d_index = nullptr;
            if (e == hipSuccess) e = hipMalloc(&d_src, bytes);
            if (e == hipSuccess) e = hipMemcpyAsync(d_src, idx->cp_occ, bytes, hipMemcpyHostToDevice, s);
            int brc = e == hipSuccess ? GBX_OK : hip_fail(e, "fmi index upload");
            if (!brc) {
                gbx_fmi_index di = *idx;
                di.cp_occ = (const gbx_fmi_cp_occ *)d_src;
                brc = fmi_index_build(&di, d_index, bytes, s);
                const hipError_t e2 = hipStreamSynchronize(s);
                if (!brc && e2 != hipSuccess) brc = hip_fail(e2, "fmi index build");
            }
            if (d_src) (void)hipFree(d_src);
            if (brc && d_index) { (void)hipFree(d_index); d_index = nullptr; }
            {
                std::lock_guard<std::mutex> lk(g_fmi_mu);
                for (size_t k = 0; k < g_fmi_cache.size(); ++k)
                    if (g_fmi_cache[k].building && same(g_fmi_cache[k])) {
                        if (brc) g_fmi_cache.erase(g_fmi_cache.begin() + (long)k);
                        else { g_fmi_cache[k].d_index = d_index; g_fmi_cache[k].bytes = bytes; g_fmi_cache[k].building = false; }
                        break;
                    }
            }
            g_fmi_cv.notify_all();
            if (brc) return brc;
        }
        use.d_index = d_index;
    }
    DevBuf denc(L), doff(L), dlen(L), dout(L), dso(L), dn(L), dw(L);
    const char *cap_env = getenv("GBX_FMI_RAW_CAP");          /* test aid: records per read slot of the first pass */
    const int cap0 = cap_env && atoi(cap_env) > 0 ? atoi(cap_env) : 0;
    size_t wb = fmi_workspace_bytes(n_reads, max_len, p->min_seed_len, cap0);
    if ((rc = denc.alloc((size_t)enc_bytes)) || (rc = doff.alloc((size_t)n_reads * 8)) || (rc = dlen.alloc((size_t)n_reads * 4)) ||
        (rc = dout.alloc((size_t)out_cap * sizeof(gbx_fmi_smem))) || (rc = dso.alloc((size_t)(n_reads + 1) * 8)) || (rc = dn.alloc(8)) ||
        (rc = dw.alloc(wb)))
        return rc;
    // one pipeline chunk (host_pipeline.h): staged uploads of the reads, the kernels on the lane's compute stream, then -
    // once the total is known - staged downloads of the records and the per-read offsets
    HostPipe pipe(L, (size_t)enc_bytes + (size_t)n_reads * 12, false);
    if ((rc = pipe.prepare(1))) return rc;
    if (n_reads > 0) {
        pipe.stage(0, denc.p, enc, (size_t)enc_bytes);
        pipe.stage(0, doff.p, read_off, (size_t)n_reads * 8);
        pipe.stage(0, dlen.p, read_len, (size_t)n_reads * 4);
    }
    pipe.start();
    if ((rc = pipe.wait_stage(0))) return pipe.finish(rc);
    if ((rc = fmi_launch(idx, d_index, p, n_reads, max_len, denc.as<uint8_t>(), doff.as<int64_t>(), dlen.as<int32_t>(),
                         dout.as<gbx_fmi_smem>(), out_cap, dso.as<int64_t>(), dn.as<int64_t>(), dw.p, wb, s, cap0)))
        return pipe.finish(rc);
    int64_t total = 0, worst = 0;
    {
        hipError_t e = hipMemcpyAsync(&total, dn.p, 8, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) return pipe.finish(hip_fail(e, "gbx_fmi_smem_host"));
    }
    *n_out = total;
    if ((rc = fmi_read_overflow(dw.p, &worst, s))) return pipe.finish(rc);
    // a read with more SMEMs than its slot holds (very repetitive text, long reads with short seeds): the job runs again
    // with larger slots.  The count a pass reports for such a read is a lower bound (the re-seeding round only sees the
    // records that were kept), so the size at least doubles and the pass is checked again.
    int64_t cap_now = cap0;
    for (int attempt = 0; worst > 0; ++attempt) {
        cap_now = std::max<int64_t>(worst + 16, 2 * std::max<int64_t>(cap_now, 48));
        wb = fmi_workspace_bytes(n_reads, max_len, p->min_seed_len, (int)cap_now);
        DevBuf dw2(L);
        if (attempt >= 6 || cap_now > (1 << 20) || (rc = dw2.alloc(wb))) {
            set_error("gbx_fmi_smem_host: a read has more than %lld SMEMs and there is no workspace for slots of that size", (long long)worst);
            return pipe.finish(rc ? rc : GBX_ERR_UNSUPPORTED);
        }
        if ((rc = fmi_launch(idx, d_index, p, n_reads, max_len, denc.as<uint8_t>(), doff.as<int64_t>(), dlen.as<int32_t>(),
                             dout.as<gbx_fmi_smem>(), out_cap, dso.as<int64_t>(), dn.as<int64_t>(), dw2.p, wb, s, (int)cap_now)))
            return pipe.finish(rc);
        hipError_t e = hipMemcpyAsync(&total, dn.p, 8, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) return pipe.finish(hip_fail(e, "gbx_fmi_smem_host"));
        *n_out = total;
        if ((rc = fmi_read_overflow(dw2.p, &worst, s))) return pipe.finish(rc);
    }
    if (smem_off) pipe.fetch(0, smem_off, dso.p, (size_t)(n_reads + 1) * 8);
    const bool fits = total <= out_cap;
    if (fits && total > 0) pipe.fetch(0, out, dout.p, (size_t)total * sizeof(gbx_fmi_smem));
    if ((rc = pipe.chunk_launched(0))) return pipe.finish(rc);
    rc = pipe.finish();
    if (!rc && !fits) {
        set_error("gbx_fmi_smem_host: %lld SMEMs do not fit out_cap = %lld", (long long)total, (long long)out_cap);
        return GBX_ERR_ARG;
    }
    return rc;
}

// The host entry: one device, or the reads cut into contiguous ranges of equal base counts over the devices of
// gbx_host_set_devices / GBX_GPUS - the driver's batches of reads (fmi.cpp:193-197: contiguous rid ranges) as per-device
// ranges; every device holds the whole index.  How many SMEMs a shard yields is only known afterwards: each shard fills a
// buffer of its own (its share of out_cap plus a margin; run again with the exact size when that was too small), then the
// runs are copied behind one another into `out` with the read ids and per-read offsets moved to the job's numbering.
int gbx_fmi_smem_host(const gbx_fmi_index *idx, const gbx_fmi_params *p, int64_t n_reads, const uint8_t *enc, int64_t enc_bytes,
                      const int64_t *read_off, const int32_t *read_len, gbx_fmi_smem *out, int64_t out_cap,
                      int64_t *smem_off, int64_t *n_out)
{
    auto one = [&]() { return fmi_host_one(idx, p, n_reads, enc, enc_bytes, read_off, read_len, out, out_cap, smem_off, n_out); };
    if (!host_multi_wanted() || fmi_check(idx, p, "gbx_fmi_smem_host") || n_reads <= 0 || out_cap < 0 || enc_bytes < 0 || !idx->cp_occ || !n_out ||
        !enc || !read_off || !read_len || (out_cap > 0 && !out))
        return one();
    for (int64_t r = 0; r < n_reads; ++r)
        if (read_len[r] < 0 || read_off[r] < 0 || read_off[r] + read_len[r] > enc_bytes) return one();
    int map[MAX_HOST_DEVICES];
    const int n_dev = host_device_set(map);
    if (n_dev < 0) return n_dev;
    const int parts = shard_parts(n_dev, n_reads, 131072);
    if (parts == 1) {
        DeviceGuard g;
        int rc = g.set(map[host_next_small_call_device(n_dev)]);
        return rc ? rc : one();
    }
    const std::vector<int64_t> cuts = split_by_cost(n_reads, parts, [&](int64_t r) { return (double)read_len[r] + 1.0; });
    // (uninitialised: a vector's resize() would value-initialise 40 bytes per record slot on the shard's thread)
    struct RawBuf { gbx_fmi_smem *p = nullptr; ~RawBuf() { free(p); } gbx_fmi_smem *data() { return p; }
                    bool resize(size_t n) { free(p); p = (gbx_fmi_smem *)malloc((n ? n : 1) * sizeof(gbx_fmi_smem)); return p != nullptr; } };
    std::vector<RawBuf> bufs((size_t)parts);
    std::vector<std::vector<int64_t>> offs((size_t)parts);
    std::vector<int64_t> counts((size_t)parts, 0);
    int rc = run_on_devices(parts, map, "gbx_fmi_smem_host", [&](int k) -> int {
        const int64_t lo = cuts[(size_t)k], hi = cuts[(size_t)k + 1], m = hi - lo;
        if (m == 0) return GBX_OK;
        int64_t a0 = enc_bytes, a1 = 0;
        for (int64_t r = lo; r < hi; ++r) { a0 = read_off[r] < a0 ? read_off[r] : a0; a1 = read_off[r] + read_len[r] > a1 ? read_off[r] + read_len[r] : a1; }
        std::vector<int64_t> ro((size_t)m);
        for (int64_t r = 0; r < m; ++r) ro[(size_t)r] = read_off[lo + r] - a0;
        offs[(size_t)k].resize((size_t)m + 1);
        // twice the shard's share of out_cap (SMEMs per read are uneven: repeats concentrate; a shard that still does not fit runs
        // again with the exact size, which doubles that device's time - the margin is there to make it rare)
        int64_t cap = (int64_t)((double)out_cap * (double)m / (double)n_reads * 2.0) + 4096;
        if (cap > out_cap) cap = out_cap;
        for (int attempt = 0;; ++attempt) {
            if (!bufs[(size_t)k].resize((size_t)cap)) { set_error("gbx_fmi_smem_host: out of host memory"); return GBX_ERR_NOMEM; }
            int64_t got = 0;
            const int rc1 = fmi_host_one(idx, p, m, enc + a0, a1 - a0, ro.data(), read_len + lo, bufs[(size_t)k].data(), cap, offs[(size_t)k].data(), &got, lo);
            counts[(size_t)k] = got;
            if (rc1 == GBX_ERR_ARG && got > cap && attempt == 0 && got <= out_cap) { cap = got; continue; }   // did not fit: once more, exactly
            if (rc1 == GBX_ERR_ARG && got > cap) return GBX_OK;      // more than the whole job's out_cap: reported below with the job's total
            return rc1;
        }
    });
    if (rc) return rc;
    int64_t total = 0;
    std::vector<int64_t> first((size_t)parts + 1, 0);
    for (int k = 0; k < parts; ++k) { first[(size_t)k] = total; total += counts[(size_t)k]; }
    first[(size_t)parts] = total;
    *n_out = total;
    if (total > out_cap) {
        set_error("gbx_fmi_smem_host: %lld SMEMs do not fit out_cap = %lld", (long long)total, (long long)out_cap);
        return GBX_ERR_ARG;
    }
    std::vector<Helper> th;
    auto merge = [&](int k) {
        const int64_t lo = cuts[(size_t)k], m = cuts[(size_t)k + 1] - lo, f = first[(size_t)k];
        const gbx_fmi_smem *src = bufs[(size_t)k].data();
        for (int64_t j = 0; j < counts[(size_t)k]; ++j) { gbx_fmi_smem rec = src[j]; rec.rid += (uint32_t)lo; out[f + j] = rec; }
        if (smem_off && m > 0) for (int64_t r = 0; r < m; ++r) smem_off[lo + r] = offs[(size_t)k][(size_t)r] + f;
    };
    for (int k = 1; k < parts; ++k) th.emplace_back([&, k] { merge(k); });
    merge(0);
    for (auto &t : th) t.join();
    if (smem_off) smem_off[n_reads] = total;
    return GBX_OK;
}

// frees the device copies of the indexes gbx_fmi_smem_host keeps between calls
int gbx_fmi_host_release(void)
{
    std::lock_guard<std::mutex> lk(g_fmi_mu);
    int cur = -1;
    (void)hipGetDevice(&cur);
    for (size_t k = 0; k < g_fmi_cache.size();) {
        if (g_fmi_cache[k].users > 0) { ++k; continue; }        // a call is using it: it goes at the next release
        (void)hipSetDevice(g_fmi_cache[k].dev);
        (void)hipFree(g_fmi_cache[k].d_index);
        g_fmi_cache.erase(g_fmi_cache.begin() + (long)k);
    }
    if (cur >= 0) (void)hipSetDevice(cur);
    (void)hipGetLastError();
    return GBX_OK;
}


}  // extern "C"
