// gbx_capi.hip — the extern "C" boundary of libgbx.so (see include/gbx.h).
// Host-buffer entry points stage through device memory here; there is no CPU
// compute path in this library: without a HIP device every compute entry
// returns GBX_ERR_NO_DEVICE.
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <chrono>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>
#include "gbx_internal.h"
#include <dlfcn.h>

namespace gbx {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int hip_fail(hipError_t e, const char *what)
{
    set_error("HIP error %d (%s) in %s", (int)e, hipGetErrorString(e), what);
    (void)hipGetLastError();
    return e == hipErrorOutOfMemory ? GBX_ERR_NOMEM : GBX_ERR_HIP;
}

static int require_device()
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        set_error("no HIP device available (libgbx has no CPU fallback)");
        return GBX_ERR_NO_DEVICE;
    }
    return GBX_OK;
}

// ---- stage profiler ---------------------------------------------------------
struct StageRec { const char *name; hipEvent_t a, b; };
static thread_local bool g_prof_on = false;
static thread_local std::vector<StageRec> g_prof;

// ---- roctx ranges (optional, GBX_ROCTX=1) -------------------------------------
namespace {
struct Roctx {
    int (*push)(const char *) = nullptr;
    int (*pop)() = nullptr;
    Roctx()
    {
        const char *e = getenv("GBX_ROCTX");
        if (!e || !*e || *e == '0') return;
        void *h = dlopen("librocprofiler-sdk-roctx.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) { fprintf(stderr, "[gbx] GBX_ROCTX set but no roctx library could be loaded: %s\n", dlerror()); return; }
        push = (int (*)(const char *))dlsym(h, "roctxRangePushA");
        pop = (int (*)())dlsym(h, "roctxRangePop");
        if (!push || !pop) push = nullptr;
    }
};
const Roctx &roctx() { static Roctx r; return r; }
}  // namespace
RoctxRange::RoctxRange(const char *name) : on_(roctx().push != nullptr) { if (on_) (void)roctx().push(name); }
RoctxRange::~RoctxRange() { if (on_) (void)roctx().pop(); }

Stage::Stage(const char *name, hipStream_t s) : slot_(-1), s_(s), range_(name)
{
    if (!g_prof_on) return;
    StageRec r{name, nullptr, nullptr};
    if (hipEventCreate(&r.a) != hipSuccess) return;
    if (hipEventCreate(&r.b) != hipSuccess) { (void)hipEventDestroy(r.a); return; }
    (void)hipEventRecord(r.a, s);
    g_prof.push_back(r);
    slot_ = (int)g_prof.size() - 1;
}
Stage::~Stage()
{
    if (slot_ >= 0) (void)hipEventRecord(g_prof[slot_].b, s_);
}

int SideStreams::fork(hipStream_t main)
{
    GBX_HIP(hipEventRecord(ev_fork, main));
    for (int k = 0; k < N; ++k) GBX_HIP(hipStreamWaitEvent(side[k], ev_fork, 0));
    return GBX_OK;
}
int SideStreams::join(hipStream_t main)
{
    for (int k = 0; k < N; ++k) {
        GBX_HIP(hipEventRecord(ev_join[k], side[k]));
        GBX_HIP(hipStreamWaitEvent(main, ev_join[k], 0));
    }
    return GBX_OK;
}
int side_streams(SideStreams **out)
{
    static std::mutex mu;
    static std::vector<std::pair<int, SideStreams *>> made;
    int dev = 0;
    GBX_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(mu);
    for (auto &m : made) if (m.first == dev) { *out = m.second; return GBX_OK; }
    SideStreams *ss = new SideStreams();
    GBX_HIP(hipEventCreateWithFlags(&ss->ev_fork, hipEventDisableTiming));
    GBX_HIP(hipEventCreateWithFlags(&ss->ev_aux, hipEventDisableTiming));
    for (int k = 0; k < SideStreams::N; ++k) {
        GBX_HIP(hipStreamCreateWithFlags(&ss->side[k], hipStreamNonBlocking));
        GBX_HIP(hipEventCreateWithFlags(&ss->ev_join[k], hipEventDisableTiming));
    }
    made.emplace_back(dev, ss);
    *out = ss;
    return GBX_OK;
}

}  // namespace gbx

#include "host_pipeline.h"

using namespace gbx;

struct gbx_timer {
    hipEvent_t a, b;
};

extern "C" {

const char *gbx_version(void) { return "gbx 0.1.0 (gfx950)"; }
const char *gbx_last_error(void) { return g_err; }

int gbx_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

int gbx_set_device(int dev)
{
    int rc = require_device();
    if (rc) return rc;
    GBX_HIP(hipSetDevice(dev));
    return GBX_OK;
}

int gbx_host_prepare(void)
{
    int rc = require_device();
    if (rc) return rc;
    HostLane lane;
    if ((rc = lane.acquire())) return rc;
    return lane_prepare_staging(lane.l);
}

int gbx_host_reserve(size_t bytes)
{
    int rc = require_device();
    if (rc) return rc;
    HostLane lane;
    if ((rc = lane.acquire())) return rc;
    DevBuf b(lane.l);
    if ((rc = b.alloc(bytes))) return rc;
    // touch it: the allocation is committed lazily, and the wait for memory another process has just released
    // (seconds for 10 GB) would otherwise hit whichever call uses or allocates device memory next
    GBX_HIP(hipMemsetAsync(b.p, 0, b.cap, lane.l->compute));
    GBX_HIP(hipStreamSynchronize(lane.l->compute));
    return GBX_OK;                  // the block goes to the lane's cache when `b` goes out of scope
}

int gbx_host_release(void)
{
    std::lock_guard<std::mutex> lk(HostLane::mu());
    for (Lane *l : HostLane::idle()) {
        for (DevBlock &b : l->dev_cache) (void)hipFree(b.p);
        l->dev_cache.clear();
    }
    return GBX_OK;
}

int gbx_device_name(char *buf, size_t cap)
{
    if (!buf || cap == 0) { set_error("gbx_device_name: null buffer"); return GBX_ERR_ARG; }
    int rc = require_device();
    if (rc) return rc;
    int dev = 0;
    GBX_HIP(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    GBX_HIP(hipGetDeviceProperties(&prop, dev));
    snprintf(buf, cap, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    return GBX_OK;
}

int gbx_timer_create(gbx_timer **t)
{
    if (!t) { set_error("gbx_timer_create: null"); return GBX_ERR_ARG; }
    int rc = require_device();
    if (rc) return rc;
    gbx_timer *x = new gbx_timer;
    GBX_HIP(hipEventCreate(&x->a));
    GBX_HIP(hipEventCreate(&x->b));
    *t = x;
    return GBX_OK;
}
int gbx_timer_start(gbx_timer *t, void *stream) { GBX_HIP(hipEventRecord(t->a, (hipStream_t)stream)); return GBX_OK; }
int gbx_timer_stop(gbx_timer *t, void *stream) { GBX_HIP(hipEventRecord(t->b, (hipStream_t)stream)); return GBX_OK; }
int gbx_timer_elapsed_ms(gbx_timer *t, float *ms)
{
    GBX_HIP(hipEventSynchronize(t->b));
    GBX_HIP(hipEventElapsedTime(ms, t->a, t->b));
    return GBX_OK;
}
void gbx_timer_destroy(gbx_timer *t)
{
    if (!t) return;
    (void)hipEventDestroy(t->a);
    (void)hipEventDestroy(t->b);
    delete t;
}

int gbx_profile_begin(void)
{
    int rc = require_device();
    if (rc) return rc;
    for (auto &r : g_prof) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    g_prof.clear();
    g_prof_on = true;
    return GBX_OK;
}

int gbx_profile_end(int cap, const char **names, float *ms_sum, int *launches, int *n_stages)
{
    g_prof_on = false;
    int n = 0;
    for (auto &r : g_prof) {
        float ms = 0.f;
        if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            int k = 0;
            while (k < n && strcmp(names[k], r.name) != 0) ++k;
            if (k == n && n < cap) { names[n] = r.name; ms_sum[n] = 0.f; launches[n] = 0; ++n; }
            if (k < n) { ms_sum[k] += ms; launches[k] += 1; }      // stages beyond `cap` distinct names are dropped
        }
        (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b);
    }
    g_prof.clear();
    if (n_stages) *n_stages = n;
    return GBX_OK;
}

int gbx_malloc_device(void **p, size_t bytes)
{
    int rc = require_device();
    if (rc) return rc;
    GBX_HIP(hipMalloc(p, bytes ? bytes : 16));
    return GBX_OK;
}
int gbx_free_device(void *p) { if (p) GBX_HIP(hipFree(p)); return GBX_OK; }
int gbx_memcpy_h2d(void *dst, const void *src, size_t bytes, void *stream)
{
    if (bytes) GBX_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    return GBX_OK;
}
int gbx_memcpy_d2h(void *dst, const void *src, size_t bytes, void *stream)
{
    if (bytes) GBX_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    return GBX_OK;
}
int gbx_stream_synchronize(void *stream) { GBX_HIP(hipStreamSynchronize((hipStream_t)stream)); return GBX_OK; }

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

int gbx_bsw_extend_host(const gbx_bsw_params *p, int64_t n,
                        const uint8_t *ref, int64_t ref_bytes,
                        const uint8_t *qer, int64_t qer_bytes,
                        const int64_t *idr, const int64_t *idq,
                        const int32_t *len1, const int32_t *len2,
                        const int32_t *h0, gbx_bsw_result *out)
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
    const int64_t chunk = bsw_host_chunk(n);
    const int64_t n_chunks = (n + chunk - 1) / chunk;
    std::vector<int64_t> need_r((size_t)n_chunks), need_q((size_t)n_chunks);
    // slices of 64 Ki pairs, a few threads when there are many; the lowest failing pair is reported
    const int64_t SL = 65536, n_slices = (n + SL - 1) / SL;
    std::vector<int64_t> slice_r((size_t)n_slices), slice_q((size_t)n_slices), slice_bad((size_t)n_slices, -1);
    std::vector<int> slice_plain((size_t)n_slices, 0);      // longest query if every pair has 1 <= qlen <= 256, tlen >= 1 and a small h0 (bsw_launch_direct), else 0
    auto check_slice = [&](int64_t sl) {
        const int64_t a = sl * SL, b = a + SL < n ? a + SL : n;
        int64_t mr = 0, mq = 0;
        bool plain = true;
        int maxq = 1;
        for (int64_t k = a; k < b; ++k) {
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
        slice_r[(size_t)sl] = mr; slice_q[(size_t)sl] = mq; slice_plain[(size_t)sl] = plain ? maxq : 0;
    };
    {
        const int vt = n_slices >= 16 ? 8 : n_slices >= 8 ? 4 : 1;      // 0.9 ms with 4 threads at 2 M pairs, on the call's critical path
        std::vector<std::thread> th;
        for (int t = 1; t < vt; ++t) th.emplace_back([&, t] { for (int64_t sl = t; sl < n_slices; sl += vt) check_slice(sl); });
        for (int64_t sl = 0; sl < n_slices; sl += vt) check_slice(sl);
        for (auto &x : th) x.join();
    }
    for (int64_t sl = 0; sl < n_slices; ++sl) {
        const int64_t k = slice_bad[(size_t)sl];
        if (k < 0) continue;
        if (len1[k] >= 0 && len2[k] >= 0 && idr[k] >= 0 && idq[k] >= 0 && idr[k] + len1[k] <= ref_bytes &&
            idq[k] + len2[k] <= qer_bytes) {
            set_error("gbx_bsw_extend_host: pair %lld exceeds GBX_BSW_MAX_QLEN/TLEN", (long long)k);
            return GBX_ERR_UNSUPPORTED;
        }
        set_error("gbx_bsw_extend_host: pair %lld lies outside the arenas", (long long)k);
        return GBX_ERR_ARG;
    }
    for (int64_t c = 0; c < n_chunks; ++c) {
        int64_t mr = 0, mq = 0;
        // chunks are multiples of 64 pairs, slices of 65536: a slice may straddle two chunks, which only makes
        // the earlier chunk wait for a few more bytes
        for (int64_t sl = c * chunk / SL; sl < n_slices && sl * SL < (c + 1) * chunk; ++sl) {
            mr = slice_r[(size_t)sl] > mr ? slice_r[(size_t)sl] : mr;
            mq = slice_q[(size_t)sl] > mq ? slice_q[(size_t)sl] : mq;
        }
        need_r[(size_t)c] = mr; need_q[(size_t)c] = mq;
    }
    int rc = require_device();
    if (rc) return rc;
    auto mark = [&](const char *what, int64_t k) { if (trace) fprintf(stderr, "[gbx host] %8.3f ms %s %lld\n", (wall_s() - t_begin) * 1e3, what, (long long)k); };
    mark("validated", n);
    // Upload, compute and download are pipelined over chunks of pairs (host_pipeline.h).  Chunk k's bases and
    // index slices go up while earlier chunks run; the arenas are uploaded front to back up to the furthest
    // byte any pair seen so far needs (a running maximum), which is right for every offset layout and streams
    // perfectly for the usual monotone one.  The chunks are queued back to back without a barrier between them
    // (own workspace each; the launch records the events a chunk's download waits for), and their results come
    // back while later chunks run.  Chunks are multiples of 64 pairs.
    const size_t wb1 = (bsw_workspace_bytes(chunk < n ? chunk : n) + 255) & ~(size_t)255;
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
    HostPipe pipe(L, (size_t)ref_bytes + (size_t)qer_bytes + (size_t)n * 28, n_chunks > 1);
    if ((rc = pipe.prepare(n_chunks))) return rc;
    // Staged (large) calls send the bases two per byte: the upload workers pack them on their way into the pinned slabs
    // (host_pipeline.h: pack4), the device expands them into the byte arenas the kernels read (bsw_unpack4) - the
    // arenas are most of the upload (2 M pairs: 590 of 640 MB), and PCIe is the longest leg of the call.
    const bool pack_bases = pipe.staged && !(getenv("GBX_BSW_PACK") && atoi(getenv("GBX_BSW_PACK")) == 0);
    DevBuf dref_p(L), dqer_p(L);
    if (pack_bases && ((rc = dref_p.alloc((size_t)ref_bytes / 2 + 16)) || (rc = dqer_p.alloc((size_t)qer_bytes / 2 + 16)))) return rc;
    std::vector<int64_t> lo_r((size_t)n_chunks), hi_r((size_t)n_chunks), lo_q((size_t)n_chunks), hi_q((size_t)n_chunks);
    int64_t up_r = 0, up_q = 0;
    for (int64_t a = 0, c = 0; a < n; a += chunk, ++c) {
        const int64_t b = a + chunk < n ? a + chunk : n, m = b - a;
        int64_t nr = need_r[(size_t)c] > up_r ? need_r[(size_t)c] : up_r;
        int64_t nq = need_q[(size_t)c] > up_q ? need_q[(size_t)c] : up_q;
        if (pack_bases) {
            // packed ranges start at even offsets: round the ends up to even while the arena allows it
            if ((nr & 1) && nr < ref_bytes) ++nr;
            if ((nq & 1) && nq < qer_bytes) ++nq;
            pipe.stage_pack4(c, dref_p.as<uint8_t>() + up_r / 2, ref + up_r, (size_t)(nr - up_r));
            pipe.stage_pack4(c, dqer_p.as<uint8_t>() + up_q / 2, qer + up_q, (size_t)(nq - up_q));
            lo_r[(size_t)c] = up_r; hi_r[(size_t)c] = nr; lo_q[(size_t)c] = up_q; hi_q[(size_t)c] = nq;
        } else {
            pipe.stage(c, dref.as<uint8_t>() + up_r, ref + up_r, (size_t)(nr - up_r));
            pipe.stage(c, dqer.as<uint8_t>() + up_q, qer + up_q, (size_t)(nq - up_q));
        }
        pipe.stage(c, didr.as<int64_t>() + a, idr + a, m * 8);
        pipe.stage(c, didq.as<int64_t>() + a, idq + a, m * 8);
        pipe.stage(c, dl1.as<int32_t>() + a, len1 + a, m * 4);
        pipe.stage(c, dl2.as<int32_t>() + a, len2 + a, m * 4);
        pipe.stage(c, dh0.as<int32_t>() + a, h0 + a, m * 4);
        up_r = nr; up_q = nq;
    }
    pipe.start();
    mark("pipeline started, chunks", n_chunks);
    // small jobs of plain pairs: one kernel launch instead of the binning passes and the class kernels
    bool direct = n <= 16384 && n_chunks == 1 && !(getenv("GBX_BSW_DIRECT") && atoi(getenv("GBX_BSW_DIRECT")) == 0);
    int direct_q = 1;
    for (int64_t sl = 0; sl < n_slices && direct; ++sl) {
        direct = slice_plain[(size_t)sl] != 0;
        direct_q = slice_plain[(size_t)sl] > direct_q ? slice_plain[(size_t)sl] : direct_q;
    }
    if (direct) {
        if ((rc = pipe.wait_stage(0))) return pipe.finish(rc);
        if (pack_bases &&
            ((rc = bsw_unpack4(dref_p.as<uint8_t>(), dref.as<uint8_t>(), lo_r[0], hi_r[0], L->compute)) ||
             (rc = bsw_unpack4(dqer_p.as<uint8_t>(), dqer.as<uint8_t>(), lo_q[0], hi_q[0], L->compute))))
            return pipe.finish(rc);
        rc = bsw_launch_direct(p, n, direct_q, dref.as<uint8_t>(), dqer.as<uint8_t>(), didr.as<int64_t>(), didq.as<int64_t>(),
                               dl1.as<int32_t>(), dl2.as<int32_t>(), dh0.as<int32_t>(), dout.as<gbx_bsw_result>(), L->compute);
        if (!rc) { pipe.fetch(0, out, dout.p, n * sizeof(gbx_bsw_result)); rc = pipe.chunk_launched(0, 0); }
        return pipe.finish(rc);
    }
    for (int64_t a = 0, c = 0; a < n; a += chunk, ++c) {
        const int64_t m = (a + chunk < n ? a + chunk : n) - a;
        if ((rc = pipe.wait_stage(c))) return pipe.finish(rc);
        mark("uploads queued, chunk", c);
        if (pack_bases &&
            ((rc = bsw_unpack4(dref_p.as<uint8_t>(), dref.as<uint8_t>(), lo_r[(size_t)c], hi_r[(size_t)c], L->compute)) ||
             (rc = bsw_unpack4(dqer_p.as<uint8_t>(), dqer.as<uint8_t>(), lo_q[(size_t)c], hi_q[(size_t)c], L->compute))))
            return pipe.finish(rc);
        // pipelined calls: no barrier between the chunks, the launch records one event per kernel stream
        hipEvent_t *je = n_chunks > 1 ? pipe.join_events(c) : nullptr;
        rc = bsw_launch(p, m, dref.as<uint8_t>(), dqer.as<uint8_t>(), didr.as<int64_t>() + a, didq.as<int64_t>() + a,
                        dl1.as<int32_t>() + a, dl2.as<int32_t>() + a, dh0.as<int32_t>() + a,
                        dout.as<gbx_bsw_result>() + a, (char *)dwork.p + wb1 * (size_t)c, wb1, L->compute, je);
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
    const int T = host_workers();
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
    const bool sparse = n >= 4096 && (ref_bytes + qer_bytes) > 2 * (packed_r + packed_q) + ((int64_t)1 << 20);
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

int gbx_chain_host(int64_t n_calls, const int64_t *anchor_off, const uint64_t *ax, const uint64_t *ay,
                   const gbx_chain_call *hdr, int32_t *score, int32_t *parent, int32_t *target, int32_t *peak)
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
        if (n < 0) { set_error("gbx_chain_host: anchor_off not monotone at call %lld", (long long)c); return GBX_ERR_ARG; }
        if (n > 0x7fffffffLL) { set_error("gbx_chain_host: call %lld has more than 2^31 anchors", (long long)c); return GBX_ERR_UNSUPPORTED; }
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
    if ((rc = pipe.prepare(1))) return rc;
    pipe.stage(0, doff.p, anchor_off, (n_calls + 1) * 8);
    pipe.stage(0, dh.p, hdr, n_calls * sizeof(gbx_chain_call));
    pipe.stage(0, dx.p, ax, na * 8);
    pipe.stage(0, dy.p, ay, na * 8);
    mark("device buffers ready");
    pipe.start();
    if ((rc = pipe.wait_stage(0))) return pipe.finish(rc);
    mark("uploads queued");
    rc = chain_launch(n_calls, na, doff.as<int64_t>(), dx.as<uint64_t>(), dy.as<uint64_t>(), dh.as<gbx_chain_call>(),
                      ds.as<int32_t>(), dp.as<int32_t>(), dt.as<int32_t>(), dk.as<int32_t>(), dw.p, wb, lane.l->compute);
    if (rc) return pipe.finish(rc);
    pipe.fetch(0, score, ds.p, na * 4);
    pipe.fetch(0, parent, dp.p, na * 4);
    if (target) pipe.fetch(0, target, dt.p, na * 4);
    if (peak) pipe.fetch(0, peak, dk.p, na * 4);
    if ((rc = pipe.chunk_launched(0))) return pipe.finish(rc);
    mark("kernels queued");
    rc = pipe.finish();
    mark("results downloaded");
    return rc;
}

/* -------------------------------------------------------------------- phmm */
int gbx_phmm_init(void)
{
    int rc = require_device();
    if (rc) return rc;
    return phmm_init_tables();
}

size_t gbx_phmm_workspace_bytes(int64_t n_pairs, int64_t n_reads, int32_t max_hap_len)
{
    return phmm_workspace_bytes(n_pairs, n_reads, max_hap_len);
}

int gbx_phmm_forward_device(int64_t n_pairs, const int32_t *d_pair_read, const int32_t *d_pair_hap,
                            int64_t n_reads, const int64_t *d_read_off, const int32_t *d_read_len,
                            const uint8_t *d_rs, const uint8_t *d_q, const uint8_t *d_i, const uint8_t *d_d,
                            const uint8_t *d_c,
                            const int64_t *d_hap_off, const int32_t *d_hap_len, const uint8_t *d_hap,
                            int32_t max_hap_len, double *d_out, void *d_work, size_t work_bytes, void *stream)
{
    if (n_pairs < 0 || n_reads < 0 || max_hap_len < 0 || max_hap_len > GBX_PHMM_MAX_HAPLEN) {
        set_error("gbx_phmm_forward_device: bad argument");
        return GBX_ERR_ARG;
    }
    if (n_pairs == 0) return GBX_OK;
    if (!d_pair_read || !d_pair_hap || !d_read_off || !d_read_len || !d_rs || !d_q || !d_i || !d_d || !d_c ||
        !d_hap_off || !d_hap_len || !d_hap || !d_out || !d_work) {
        set_error("gbx_phmm_forward_device: null pointer");
        return GBX_ERR_ARG;
    }
    int rc = require_device();
    if (rc) return rc;
    return phmm_launch(n_pairs, d_pair_read, d_pair_hap, n_reads, d_read_off, d_read_len, d_rs, d_q, d_i, d_d, d_c,
                       d_hap_off, d_hap_len, d_hap, max_hap_len, d_out, d_work, work_bytes, (hipStream_t)stream);
}

int gbx_phmm_forward_host(int64_t n_pairs, const int32_t *pair_read, const int32_t *pair_hap,
                          int64_t n_reads, const int64_t *read_off, const int32_t *read_len, int64_t read_bytes,
                          const uint8_t *rs, const uint8_t *q, const uint8_t *i, const uint8_t *d, const uint8_t *c,
                          int64_t n_haps, const int64_t *hap_off, const int32_t *hap_len, int64_t hap_bytes,
                          const uint8_t *hap, double *out)
{
    RoctxRange range_("gbx_phmm_forward_host");
    const bool trace = getenv("GBX_HOST_TRACE") != nullptr;
    const double t_begin = wall_s();
    auto mark = [&](const char *what) { if (trace) fprintf(stderr, "[gbx phmm host] %9.3f ms %s\n", (wall_s() - t_begin) * 1e3, what); };
    if (n_pairs < 0 || n_reads < 0 || n_haps < 0 || read_bytes < 0 || hap_bytes < 0) {
        set_error("gbx_phmm_forward_host: bad argument");
        return GBX_ERR_ARG;
    }
    if (n_pairs == 0) return GBX_OK;
    if (!pair_read || !pair_hap || !read_off || !read_len || !rs || !q || !i || !d || !c || !hap_off || !hap_len ||
        !hap || !out) {
        set_error("gbx_phmm_forward_host: null pointer");
        return GBX_ERR_ARG;
    }
    int max_h = 1;
    // exact size of the haplotype streams (sum over the pairs of haplen+1): one long haplotype must not size the
    // workspace of every pair
    int64_t stream_syms = 0;
    for (int64_t k = 0; k < n_reads; ++k)
        if (read_len[k] < 0 || read_off[k] < 0 || read_off[k] + read_len[k] > read_bytes) {
            set_error("gbx_phmm_forward_host: read %lld lies outside the arena", (long long)k);
            return GBX_ERR_ARG;
        }
    for (int64_t k = 0; k < n_haps; ++k) {
        if (hap_len[k] < 0 || hap_off[k] < 0 || hap_off[k] + hap_len[k] > hap_bytes) {
            set_error("gbx_phmm_forward_host: haplotype %lld lies outside the arena", (long long)k);
            return GBX_ERR_ARG;
        }
        if (hap_len[k] > GBX_PHMM_MAX_HAPLEN) {
            set_error("gbx_phmm_forward_host: haplotype %lld longer than GBX_PHMM_MAX_HAPLEN", (long long)k);
            return GBX_ERR_UNSUPPORTED;
        }
        if (hap_len[k] > max_h) max_h = hap_len[k];
    }
    {
        // the pair list is the long one (10.9 M entries in the 'large' job): a few threads, lowest bad index reported
        const int T = host_workers();
        std::vector<int64_t> bad((size_t)T, -1), syms((size_t)T, 0);
        parallel_ranges(n_pairs, T, [&](int t, int64_t lo, int64_t hi) {
            int64_t sum = 0;
            for (int64_t k = lo; k < hi; ++k) {
                if (pair_read[k] < 0 || pair_read[k] >= n_reads || pair_hap[k] < 0 || pair_hap[k] >= n_haps) { bad[(size_t)t] = k; return; }
                sum += (int64_t)hap_len[pair_hap[k]] + 1;
            }
            syms[(size_t)t] = sum;
        });
        for (int t = 0; t < T; ++t) stream_syms += syms[(size_t)t];
        for (int t = 0; t < T; ++t)
            if (bad[(size_t)t] >= 0) {
                set_error("gbx_phmm_forward_host: pair %lld names a read/haplotype out of range", (long long)bad[(size_t)t]);
                return GBX_ERR_ARG;
            }
    }
    mark("validated");
    int rc = require_device();
    if (rc) return rc;
    HostLane lane;
    if ((rc = lane.acquire())) return rc;
    Lane *L = lane.l;
    DevBuf dpr(L), dph(L), dro(L), drl(L), drs(L), dq(L), di(L), dd(L), dc(L), dho(L), dhl(L), dh(L), dout(L), dw(L);
    const size_t wb = phmm_workspace_bytes(n_pairs, n_reads, max_h, stream_syms);
    if ((rc = dpr.alloc(n_pairs * 4)) || (rc = dph.alloc(n_pairs * 4)) || (rc = dro.alloc(n_reads * 8)) ||
        (rc = drl.alloc(n_reads * 4)) || (rc = drs.alloc(read_bytes)) || (rc = dq.alloc(read_bytes)) ||
        (rc = di.alloc(read_bytes)) || (rc = dd.alloc(read_bytes)) || (rc = dc.alloc(read_bytes)) ||
        (rc = dho.alloc(n_haps * 8)) || (rc = dhl.alloc(n_haps * 4)) || (rc = dh.alloc(hap_bytes)) ||
        (rc = dout.alloc(n_pairs * 8)) || (rc = dw.alloc(wb)))
        return rc;
    HostPipe pipe(lane.l, (size_t)read_bytes * 5 + (size_t)hap_bytes + (size_t)n_pairs * 8 + (size_t)(n_reads + n_haps) * 12, false);
    if ((rc = pipe.prepare(1))) return rc;
    pipe.stage(0, dpr.p, pair_read, n_pairs * 4); pipe.stage(0, dph.p, pair_hap, n_pairs * 4);
    pipe.stage(0, dro.p, read_off, n_reads * 8); pipe.stage(0, drl.p, read_len, n_reads * 4);
    pipe.stage(0, drs.p, rs, read_bytes); pipe.stage(0, dq.p, q, read_bytes); pipe.stage(0, di.p, i, read_bytes);
    pipe.stage(0, dd.p, d, read_bytes); pipe.stage(0, dc.p, c, read_bytes);
    pipe.stage(0, dho.p, hap_off, n_haps * 8); pipe.stage(0, dhl.p, hap_len, n_haps * 4);
    pipe.stage(0, dh.p, hap, hap_bytes);
    mark("device buffers ready");
    pipe.start();
    if ((rc = pipe.wait_stage(0))) return pipe.finish(rc);
    mark("uploads queued");
    rc = phmm_launch(n_pairs, dpr.as<int32_t>(), dph.as<int32_t>(), n_reads, dro.as<int64_t>(), drl.as<int32_t>(),
                     drs.as<uint8_t>(), dq.as<uint8_t>(), di.as<uint8_t>(), dd.as<uint8_t>(), dc.as<uint8_t>(),
                     dho.as<int64_t>(), dhl.as<int32_t>(), dh.as<uint8_t>(), max_h, dout.as<double>(), dw.p, wb,
                     lane.l->compute, stream_syms);
    if (rc) return pipe.finish(rc);
    pipe.fetch(0, out, dout.p, n_pairs * 8);
    if ((rc = pipe.chunk_launched(0))) return pipe.finish(rc);
    mark("kernels queued");
    rc = pipe.finish();
    mark("results downloaded");
    return rc;
}

/* --------------------------------------------------------------------- poa */
void gbx_poa_default_params(gbx_poa_params *p)
{
    memset(p, 0, sizeof(*p));
    p->m = 2; p->n = -4; p->g = -6; p->e = -2; p->q = -25; p->c = -1;   /* msa_spoa_omp.cpp:156-162,184 */
}

int gbx_poa_plan_host(int64_t n_windows, const int64_t *win_first_seq, const int32_t *seq_len, gbx_poa_plan *plan)
{
    if (n_windows < 0 || !plan || (n_windows > 0 && (!win_first_seq || !seq_len))) {
        set_error("gbx_poa_plan_host: bad argument");
        return GBX_ERR_ARG;
    }
    int lmax = 1, smax = 1;
    int64_t bmax = 1, n_long = 0;
    for (int64_t w = 0; w < n_windows; ++w) {
        const int64_t a = win_first_seq[w], b = win_first_seq[w + 1];
        if (b < a) { set_error("gbx_poa_plan_host: win_first_seq not monotone at window %lld", (long long)w); return GBX_ERR_ARG; }
        if (b - a > GBX_POA_MAX_SEQS_PER_WINDOW) {
            set_error("gbx_poa_plan_host: window %lld has more than %d sequences", (long long)w, GBX_POA_MAX_SEQS_PER_WINDOW);
            return GBX_ERR_UNSUPPORTED;
        }
        int64_t bases = 0;
        bool is_long = false;
        for (int64_t s = a; s < b; ++s) {
            if (seq_len[s] < 0) { set_error("gbx_poa_plan_host: negative sequence length"); return GBX_ERR_ARG; }
            if (seq_len[s] > lmax) lmax = seq_len[s];
            if (seq_len[s] > POA_PIPE_MAXLEN) is_long = true;
            bases += seq_len[s];
        }
        if (is_long) ++n_long;
        if (b - a > smax) smax = (int)(b - a);
        if (bases > bmax) bmax = bases;
    }
    if (n_long > 0x7fffffff) { set_error("gbx_poa_plan_host: too many windows"); return GBX_ERR_UNSUPPORTED; }
    plan->max_seq_len = lmax;
    plan->max_seqs_per_window = smax < 4 ? 4 : ((smax + 3) & ~3);     /* multiple of 4: 16-byte aligned edge rows */
    int nf = 6;                                            /* typical windows stay below ~3.6x the read length */
    if (const char *e = getenv("GBX_POA_NODE_FACTOR")) { const int v = atoi(e); if (v >= 2 && v <= 16) nf = v; }   /* tuning aid */
    int64_t cap = (int64_t)nf * lmax + 256;
    if (bmax + 8 < cap) cap = bmax + 8;
    plan->node_cap = (int32_t)cap;
    int cus = 256, dev = 0;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    else (void)hipGetLastError();
    const int64_t resident = (int64_t)cus * poa_waves_per_cu(plan->node_cap);   /* one window per resident wavefront */
    const int64_t n_main = n_windows - n_long;
    plan->n_slots = (int32_t)(n_main < resident ? n_main : resident);
    plan->n_long_windows = (int32_t)n_long;
    plan->long_slots = (int32_t)(n_long < resident ? n_long : resident);
    plan->n_windows = n_windows;
    return GBX_OK;
}

size_t gbx_poa_workspace_bytes(const gbx_poa_plan *plan)
{
    if (!plan) return 0;
    return poa_workspace_bytes(plan);
}

int gbx_poa_cells(const gbx_poa_plan *plan, const void *d_work, int64_t *cells, void *stream)
{
    if (!plan || !d_work || !cells) { set_error("gbx_poa_cells: null pointer"); return GBX_ERR_ARG; }
    return poa_read_cells(d_work, poa_slot_bytes(plan->node_cap, plan->max_seqs_per_window, plan->max_seq_len, false) * (size_t)(plan->n_slots > 0 ? plan->n_slots : 0),
                          cells, (hipStream_t)stream);
}

/* development aid (scripts/dbg_poa_phases.py): byte offset of the 32-counter block inside a poa workspace */
size_t gbx_debug_poa_counter_offset(const gbx_poa_plan *plan)
{
    return poa_slot_bytes(plan->node_cap, plan->max_seqs_per_window, plan->max_seq_len, false) * (size_t)(plan->n_slots > 0 ? plan->n_slots : 0);
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

int gbx_poa_consensus_device(const gbx_poa_params *p, const gbx_poa_plan *plan, int64_t n_windows,
                             const int64_t *d_win_first_seq, const int64_t *d_seq_off, const int32_t *d_seq_len,
                             const char *d_arena, char *d_cons, int32_t *d_cons_len, int32_t *d_status,
                             int64_t cons_stride, void *d_work, size_t work_bytes, void *stream)
{
    if (!p || !plan || n_windows < 0 || cons_stride <= 0) { set_error("gbx_poa_consensus_device: bad argument"); return GBX_ERR_ARG; }
    if (n_windows == 0) return GBX_OK;
    if (!d_win_first_seq || !d_seq_off || !d_seq_len || !d_arena || !d_cons || !d_cons_len || !d_status || !d_work) {
        set_error("gbx_poa_consensus_device: null pointer");
        return GBX_ERR_ARG;
    }
    int rc = require_device();
    if (rc) return rc;
    return poa_launch(p, plan, n_windows, d_win_first_seq, d_seq_off, d_seq_len, (const uint8_t *)d_arena, (uint8_t *)d_cons, d_cons_len,
                      d_status, cons_stride, d_work, work_bytes, (hipStream_t)stream);
}

int gbx_poa_consensus_host(const gbx_poa_params *p, int64_t n_windows, const int64_t *win_first_seq,
                           int64_t n_seqs, const int64_t *seq_off, const int32_t *seq_len,
                           const char *arena, int64_t arena_bytes,
                           char *cons, int32_t *cons_len, int64_t cons_stride)
{
    RoctxRange range_("gbx_poa_consensus_host");
    if (!p || n_windows < 0 || n_seqs < 0 || arena_bytes < 0 || cons_stride <= 0) {
        set_error("gbx_poa_consensus_host: bad argument");
        return GBX_ERR_ARG;
    }
    if (n_windows == 0) return GBX_OK;
    if (!win_first_seq || !seq_off || !seq_len || !arena || !cons || !cons_len) {
        set_error("gbx_poa_consensus_host: null pointer");
        return GBX_ERR_ARG;
    }
    if (win_first_seq[0] != 0 || win_first_seq[n_windows] != n_seqs) {
        set_error("gbx_poa_consensus_host: win_first_seq must span [0, n_seqs]");
        return GBX_ERR_ARG;
    }
    for (int64_t s = 0; s < n_seqs; ++s)
        if (seq_len[s] < 0 || seq_off[s] < 0 || seq_off[s] + seq_len[s] > arena_bytes) {
            set_error("gbx_poa_consensus_host: sequence %lld lies outside the arena", (long long)s);
            return GBX_ERR_ARG;
        }
    int rc = require_device();
    if (rc) return rc;
    gbx_poa_plan plan;
    if ((rc = gbx_poa_plan_host(n_windows, win_first_seq, seq_len, &plan))) return rc;
    const size_t wb = gbx_poa_workspace_bytes(&plan);
    const bool trace = getenv("GBX_HOST_TRACE") != nullptr;
    const double t_begin = wall_s();
    auto mark = [&](const char *what) { if (trace) fprintf(stderr, "[gbx poa host] %9.3f ms %s\n", (wall_s() - t_begin) * 1e3, what); };
    HostLane lane;
    if ((rc = lane.acquire())) return rc;
    Lane *L = lane.l;
    DevBuf dwf(L), doff(L), dlen(L), dar(L), dcons(L), dcl(L), dst(L), dw(L);
    if ((rc = dwf.alloc((n_windows + 1) * 8)) || (rc = doff.alloc(n_seqs * 8)) || (rc = dlen.alloc(n_seqs * 4)) ||
        (rc = dar.alloc(arena_bytes)) || (rc = dcons.alloc(n_windows * cons_stride)) || (rc = dcl.alloc(n_windows * 4)) ||
        (rc = dst.alloc(n_windows * 4)) || (rc = dw.alloc(wb)))
        return rc;
    mark("allocated");
    std::vector<int32_t> status(n_windows);
    {
        HostPipe pipe(lane.l, (size_t)arena_bytes + (size_t)n_seqs * 12 + (size_t)n_windows * 8, false);
        if ((rc = pipe.prepare(1))) return rc;
        pipe.stage(0, dwf.p, win_first_seq, (n_windows + 1) * 8);
        pipe.stage(0, doff.p, seq_off, n_seqs * 8);
        pipe.stage(0, dlen.p, seq_len, n_seqs * 4);
        pipe.stage(0, dar.p, arena, arena_bytes);
        pipe.start();
        if ((rc = pipe.wait_stage(0))) return pipe.finish(rc);
        rc = poa_launch(p, &plan, n_windows, dwf.as<int64_t>(), doff.as<int64_t>(), dlen.as<int32_t>(), dar.as<uint8_t>(),
                        dcons.as<uint8_t>(), dcl.as<int32_t>(), dst.as<int32_t>(), cons_stride, dw.p, wb, lane.l->compute);
            if (rc) return pipe.finish(rc);
        pipe.fetch(0, cons, dcons.p, n_windows * cons_stride);
        pipe.fetch(0, cons_len, dcl.p, n_windows * 4);
        pipe.fetch(0, status.data(), dst.p, n_windows * 4);
        if ((rc = pipe.chunk_launched(0))) return pipe.finish(rc);
        mark("kernels queued");
        if ((rc = pipe.finish())) return rc;
        mark("results fetched");
    }
    // Windows whose graph outgrew the first pass's node capacity (deep or noisy windows; the plan sizes it for the
    // typical case so that the 'large' job's slots stay small) run again with room for the worst case of exactly
    // those windows: every base its own node, bounded by what int16 scores admit.  spoa has no such limit
    // (msa_spoa_omp.cpp:237-252), so only a window that cannot be represented at all fails the call.
    std::vector<int64_t> redo;
    for (int64_t w = 0; w < n_windows; ++w)
        if (status[w] & GBX_POA_ST_NODES) redo.push_back(w);      // (other bits set next to it are re-decided by the second pass)
    if (!redo.empty()) {
        std::vector<int64_t> wf(redo.size() + 1, 0), off;
        std::vector<int32_t> len;
        int64_t bmax = 1, n_long = 0;
        int lmax = 1, smax = 1;
        for (size_t k = 0; k < redo.size(); ++k) {
            const int64_t a = win_first_seq[redo[k]], b = win_first_seq[redo[k] + 1];
            int64_t bases = 0;
            bool is_long = false;
            for (int64_t sidx = a; sidx < b; ++sidx) {
                off.push_back(seq_off[sidx]); len.push_back(seq_len[sidx]);
                bases += seq_len[sidx];
                if (seq_len[sidx] > lmax) lmax = seq_len[sidx];
                if (seq_len[sidx] > POA_PIPE_MAXLEN) is_long = true;
            }
            if (is_long) ++n_long;
            if (b - a > smax) smax = (int)(b - a);
            if (bases > bmax) bmax = bases;
            wf[k + 1] = (int64_t)off.size();
        }
        int64_t cap = bmax + 8;
        while (cap > plan.node_cap && !poa_scores_fit_int16(p, cap, lmax)) cap -= (cap - plan.node_cap + 1) / 2;
        if (cap > plan.node_cap) {
            const int64_t nr = (int64_t)redo.size(), ns = (int64_t)off.size();
            gbx_poa_plan big = plan;
            big.max_seq_len = lmax;
            big.max_seqs_per_window = smax < 4 ? 4 : ((smax + 3) & ~3);
            big.node_cap = (int32_t)cap;
            big.n_windows = nr;
            big.n_long_windows = (int32_t)n_long;
            // slots: what the device has room for beside the first pass's buffers (still held), at most 16 GB, at most one per window
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); free_b = (size_t)16 << 30; }
            size_t budget = free_b / 2 < ((size_t)16 << 30) ? free_b / 2 : ((size_t)16 << 30);
            DevBuf dwf2(L), doff2(L), dlen2(L), dcons2(L), dcl2(L), dst2(L), dw2(L);
            if ((rc = dwf2.alloc((nr + 1) * 8)) || (rc = doff2.alloc(ns * 8)) || (rc = dlen2.alloc(ns * 4)) ||
                (rc = dcons2.alloc(nr * cons_stride)) || (rc = dcl2.alloc(nr * 4)) || (rc = dst2.alloc(nr * 4)))
                return rc;
            size_t wb2 = 0;
            for (;;) {                                      // fewer slots when the allocation fails
                const size_t ms = poa_slot_bytes(big.node_cap, big.max_seqs_per_window, big.max_seq_len, false);
                const size_t ls = poa_slot_bytes(big.node_cap, big.max_seqs_per_window, big.max_seq_len, true);
                int64_t slots = (int64_t)(budget / (ls ? ls : 1));
                if (slots < 1) slots = 1;
                if (slots > plan.n_slots + plan.long_slots && plan.n_slots + plan.long_slots > 0) slots = plan.n_slots + plan.long_slots;
                const int64_t n_main = nr - n_long;
                big.n_slots = (int32_t)(n_main < slots ? n_main : slots);
                big.long_slots = (int32_t)(n_long < slots ? n_long : slots);
                (void)ms;
                wb2 = gbx_poa_workspace_bytes(&big);
                if (dw2.alloc(wb2) == GBX_OK) break;
                if (budget <= ls) { set_error("gbx_poa_consensus_host: no device memory for the second pass of %lld oversized window(s)", (long long)nr); return GBX_ERR_NOMEM; }
                budget /= 2;
            }
            hipStream_t st = lane.l->compute;
            GBX_HIP(hipMemcpyAsync(dwf2.p, wf.data(), (size_t)(nr + 1) * 8, hipMemcpyHostToDevice, st));
            GBX_HIP(hipMemcpyAsync(doff2.p, off.data(), (size_t)ns * 8, hipMemcpyHostToDevice, st));
            GBX_HIP(hipMemcpyAsync(dlen2.p, len.data(), (size_t)ns * 4, hipMemcpyHostToDevice, st));
            if ((rc = poa_launch(p, &big, nr, dwf2.as<int64_t>(), doff2.as<int64_t>(), dlen2.as<int32_t>(), dar.as<uint8_t>(),
                                 dcons2.as<uint8_t>(), dcl2.as<int32_t>(), dst2.as<int32_t>(), cons_stride, dw2.p, wb2, st)))
                return rc;
            std::vector<char> c2((size_t)nr * (size_t)cons_stride);
            std::vector<int32_t> l2((size_t)nr), s2((size_t)nr);
            GBX_HIP(hipMemcpyAsync(c2.data(), dcons2.p, c2.size(), hipMemcpyDeviceToHost, st));
            GBX_HIP(hipMemcpyAsync(l2.data(), dcl2.p, (size_t)nr * 4, hipMemcpyDeviceToHost, st));
            GBX_HIP(hipMemcpyAsync(s2.data(), dst2.p, (size_t)nr * 4, hipMemcpyDeviceToHost, st));
            GBX_HIP(hipStreamSynchronize(st));
            for (int64_t k = 0; k < nr; ++k) {
                status[(size_t)redo[(size_t)k]] = s2[(size_t)k];
                if (s2[(size_t)k]) continue;
                cons_len[redo[(size_t)k]] = l2[(size_t)k];
                memcpy(cons + redo[(size_t)k] * cons_stride, c2.data() + (size_t)k * (size_t)cons_stride, (size_t)cons_stride);
            }
            mark("oversized windows redone");
        }
    }
    int64_t n_bad = 0, first_bad = -1;
    for (int64_t w = 0; w < n_windows; ++w)
        if (status[w]) { if (first_bad < 0) first_bad = w; ++n_bad; }
    if (n_bad) {
        set_error("gbx_poa_consensus_host: %lld window(s) exceeded a device capacity, first is window %lld (status bits 0x%x, "
                  "see GBX_POA_ST_*); the others' results are valid", (long long)n_bad, (long long)first_bad, status[(size_t)first_bad]);
        return GBX_ERR_UNSUPPORTED;
    }
    return GBX_OK;
}

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

int gbx_abea_align_host(int64_t n_reads, const int64_t *seq_off, const int32_t *seq_len, const char *seq_arena,
                        int64_t seq_bytes, const int64_t *event_off, const gbx_abea_event *events,
                        const gbx_abea_model *models, const float *scale, const float *shift,
                        gbx_abea_pair *out, int32_t *n_pairs)
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
            set_error("gbx_abea_align_host: read %lld lies outside the arena", (long long)r);
            return GBX_ERR_ARG;
        }
        if (event_off[r + 1] < event_off[r] || event_off[r] < 0) { set_error("gbx_abea_align_host: event_off not monotone at read %lld", (long long)r); return GBX_ERR_ARG; }
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
// FMI_search tables for every batch of reads, fmi.cpp:218): keyed by the table's address and its scalars, one per device.
namespace {
struct FmiCached { int dev; const void *host_cp; int64_t len, sentinel, count1; void *d_index; size_t bytes; };
std::mutex g_fmi_mu;
std::vector<FmiCached> g_fmi_cache;
}

int gbx_fmi_smem_host(const gbx_fmi_index *idx, const gbx_fmi_params *p, int64_t n_reads, const uint8_t *enc, int64_t enc_bytes,
                      const int64_t *read_off, const int32_t *read_len, gbx_fmi_smem *out, int64_t out_cap,
                      int64_t *smem_off, int64_t *n_out)
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
            set_error("gbx_fmi_smem_host: read %lld lies outside the base buffer", (long long)r);
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
    void *d_index = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_fmi_mu);
        for (const FmiCached &c : g_fmi_cache)
            if (c.dev == dev && c.host_cp == idx->cp_occ && c.len == idx->ref_seq_len && c.sentinel == idx->sentinel_index && c.count1 == idx->count[1])
                d_index = c.d_index;
        if (!d_index) {
            const size_t bytes = fmi_index_bytes(idx->ref_seq_len);
            void *d_src = nullptr;
            GBX_HIP(hipMalloc(&d_index, bytes));
            hipError_t e = hipMalloc(&d_src, bytes);
            if (e == hipSuccess) e = hipMemcpyAsync(d_src, idx->cp_occ, bytes, hipMemcpyHostToDevice, s);
            if (e != hipSuccess) { (void)hipFree(d_index); if (d_src) (void)hipFree(d_src); return hip_fail(e, "fmi index upload"); }
            gbx_fmi_index di = *idx;
            di.cp_occ = (const gbx_fmi_cp_occ *)d_src;
            rc = fmi_index_build(&di, d_index, bytes, s);
            hipError_t e2 = hipStreamSynchronize(s);
            (void)hipFree(d_src);
            if (rc || e2 != hipSuccess) { (void)hipFree(d_index); return rc ? rc : hip_fail(e2, "fmi index build"); }
            g_fmi_cache.push_back(FmiCached{dev, idx->cp_occ, idx->ref_seq_len, idx->sentinel_index, idx->count[1], d_index, bytes});
        }
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

// frees the device copies of the indexes gbx_fmi_smem_host keeps between calls
int gbx_fmi_host_release(void)
{
    std::lock_guard<std::mutex> lk(g_fmi_mu);
    for (FmiCached &c : g_fmi_cache) (void)hipFree(c.d_index);
    g_fmi_cache.clear();
    return GBX_OK;
}

}  // extern "C"
