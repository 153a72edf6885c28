// gbx_core.hip — what the extern "C" boundary of libgbx.so (include/gbx.h) shares between the kernels: error text, device
// helpers, the stage profiler, roctx ranges, side streams, the host entries' device set.  The per-kernel entries live in
// capi_<kernel>.hip.  There is no CPU compute path in this library: without a HIP device every compute entry returns
// GBX_ERR_NO_DEVICE.
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

int require_device()
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
bool profile_active() { return g_prof_on; }

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
    // Stream priorities (GBX_SIDE_PRIO="-1,-1,0": one number per side stream, HIP's scale - lower = more urgent; default all
    // 0).  A tuning aid that stayed one: bsw puts its longest-query classes on side streams 0 and 1 - their wavefronts hold
    // the most LDS and run at the lowest occupancy, so they should finish first and leave the tail to the short-query
    // classes - but on MI355X the priority of a stream did not move which launch gets the workgroup slots: "-1,-1,0",
    // "-1,0,1", "-1,-1,-1" and "0,0,0" all gave 5.77-5.83 ms per step on 'large' (profiles/r04h_ab.txt).
    int prio[SideStreams::N];
    {
        int least = 0, greatest = 0;
        if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) { (void)hipGetLastError(); least = greatest = 0; }
        const char *e = getenv("GBX_SIDE_PRIO");
        const char *q = e ? e : "0,0,0";
        for (int k = 0; k < SideStreams::N; ++k) {
            int v = atoi(q);
            v = v < greatest ? greatest : v > least ? least : v;
            prio[k] = v;
            while (*q && *q != ',') ++q;
            if (*q == ',') ++q;
        }
    }
    for (int k = 0; k < SideStreams::N; ++k) {
        GBX_HIP(hipStreamCreateWithPriority(&ss->side[k], hipStreamNonBlocking, prio[k]));
        GBX_HIP(hipEventCreateWithFlags(&ss->ev_join[k], hipEventDisableTiming));
    }
    {
        int least = 0, greatest = 0;
        if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) { (void)hipGetLastError(); least = greatest = 0; }
        for (int k = 0; k < 2; ++k) GBX_HIP(hipStreamCreateWithPriority(&ss->pre[k], hipStreamNonBlocking, greatest));
        GBX_HIP(hipEventCreateWithFlags(&ss->ev_pre, hipEventDisableTiming));
    }
    made.emplace_back(dev, ss);
    *out = ss;
    return GBX_OK;
}

}  // namespace gbx

#include "capi_common.h"

namespace gbx {

// ---- the host entries' device set (host_multi.h) ------------------------------
static std::atomic<int> g_host_devices{0};          // 0: not set by the caller -> GBX_GPUS, else 1
static std::atomic<unsigned> g_small_call_rr{0};

int host_device_set(int *map)
{
    int n = g_host_devices.load();
    if (n <= 0) {
        const char *e = getenv("GBX_GPUS");
        n = e && atoi(e) > 0 ? atoi(e) : 1;
    }
    if (n > MAX_HOST_DEVICES) n = MAX_HOST_DEVICES;
    int have = 0;
    if (hipGetDeviceCount(&have) != hipSuccess || have <= 0) {
        (void)hipGetLastError();
        set_error("no HIP device available (libgbx has no CPU fallback)");
        return GBX_ERR_NO_DEVICE;
    }
    for (int k = 0; k < n; ++k) map[k] = k;
    if (const char *m = getenv("GBX_DEVICE_MAP")) {      /* test aid: "0,0,0" runs three logical devices on GPU 0 */
        int k = 0;
        for (const char *q = m; *q && k < n;) {
            map[k++] = atoi(q);
            while (*q && *q != ',') ++q;
            if (*q == ',') ++q;
        }
        for (; k < n; ++k) map[k] = map[k - 1 < 0 ? 0 : k - 1];
    }
    for (int k = 0; k < n; ++k)
        if (map[k] < 0 || map[k] >= have) {
            set_error("the host entries were asked to use %d device(s) (gbx_host_set_devices / GBX_GPUS), logical device %d is HIP "
                      "device %d, but only %d exist", n, k, map[k], have);
            return GBX_ERR_ARG;
        }
    return n;
}

bool host_multi_wanted()
{
    if (getenv("GBX_DEVICE_MAP")) return true;
    int n = g_host_devices.load();
    if (n <= 0) { const char *e = getenv("GBX_GPUS"); n = e ? atoi(e) : 1; }
    return n > 1;
}

int host_next_small_call_device(int n) { return n <= 1 ? 0 : (int)(g_small_call_rr.fetch_add(1) % (unsigned)n); }

}  // namespace gbx

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

int gbx_host_set_devices(int n_gpus)
{
    if (n_gpus < 0 || n_gpus > MAX_HOST_DEVICES) { set_error("gbx_host_set_devices: 0 (default) .. %d devices", MAX_HOST_DEVICES); return GBX_ERR_ARG; }
    if (n_gpus > 1 && !getenv("GBX_DEVICE_MAP")) {
        int have = 0;
        if (hipGetDeviceCount(&have) != hipSuccess) { (void)hipGetLastError(); have = 0; }
        if (n_gpus > have) { set_error("gbx_host_set_devices: %d devices asked for, %d present", n_gpus, have); return GBX_ERR_ARG; }
    }
    g_host_devices.store(n_gpus);
    return GBX_OK;
}

int gbx_split_by_cost(int64_t n_units, const double *cost, int n_parts, int64_t *cuts)
{
    if (n_units < 0 || n_parts < 1 || !cuts || (n_units > 0 && !cost)) { set_error("gbx_split_by_cost: bad argument"); return GBX_ERR_ARG; }
    const std::vector<int64_t> c = split_by_cost(n_units, n_parts, [&](int64_t i) { return cost[i] > 0.0 ? cost[i] : 0.0; });
    for (int k = 0; k <= n_parts; ++k) cuts[k] = c[(size_t)k];
    return GBX_OK;
}

int gbx_host_devices(void)
{
    int map[MAX_HOST_DEVICES];
    const int n = host_device_set(map);
    return n < 0 ? 0 : n;
}

int gbx_host_prepare(void)
{
    int rc = require_device();
    if (rc) return rc;
    int map[MAX_HOST_DEVICES];
    const int n = host_device_set(map);
    if (n < 0) return n;
    if (n == 1 && !getenv("GBX_DEVICE_MAP")) {          // the calling thread's current device
        HostLane lane;
        if ((rc = lane.acquire())) return rc;
        return lane_prepare_staging(lane.l);
    }
    // one lane per logical device (two logical devices on one physical one: two lanes), all held at once so that each
    // acquire makes a new lane rather than re-taking the one just prepared
    std::vector<HostLane> lanes((size_t)n);
    int cur = 0;
    GBX_HIP(hipGetDevice(&cur));
    for (int k = 0; k < n && !rc; ++k) {
        hipError_t e = hipSetDevice(map[k]);
        if (e != hipSuccess) rc = hip_fail(e, "hipSetDevice");
        else if (!(rc = lanes[(size_t)k].acquire())) rc = lane_prepare_staging(lanes[(size_t)k].l);
    }
    (void)hipSetDevice(cur);
    return rc;
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
    int cur = -1;
    (void)hipGetDevice(&cur);
    for (Lane *l : HostLane::idle()) {
        (void)hipSetDevice(l->dev);
        for (DevBlock &b : l->dev_cache) (void)hipFree(b.p);
        l->dev_cache.clear();
    }
    if (cur >= 0) (void)hipSetDevice(cur);
    (void)hipGetLastError();
    return GBX_OK;
}

int gbx_host_combine_stats(int kernel, uint64_t out[4], int reset)
{
    Combiner *c = kernel == GBX_GK_BSW ? &combiner_bsw() : kernel == GBX_GK_PHMM ? &combiner_phmm() : kernel == GBX_GK_POA ? &combiner_poa() : nullptr;
    if (!c || !out) { set_error("gbx_host_combine_stats: kernel 1 (bsw), 3 (phmm) or 4 (poa), and four words"); return GBX_ERR_ARG; }
    out[0] = c->n_calls.load(); out[1] = c->n_batches.load(); out[2] = c->n_shared.load(); out[3] = c->largest.load();
    if (reset) { c->n_calls = 0; c->n_batches = 0; c->n_shared = 0; c->largest = 0; }
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

}  // extern "C"
