// host_pipeline.h — what the *_host entry points of libgbx.so use to move caller-owned (pageable) buffers to the
// device and the results back while kernels run.  Included through capi_common.h by the capi_<kernel>.hip files.
//
// A call takes a Lane (streams + pinned staging slabs, pooled per device, one per concurrent caller) and
// describes its uploads as an ordered list of stages; stage c is everything chunk c of the input needs.  Large
// calls spawn a few upload workers for the duration of the call: each takes the next piece (<= 8 MiB) of the
// list, copies it from the caller's memory into one of its two pinned slabs and queues the DMA on the lane's
// copy stream.  The caller's pages are never pinned (the runtime would pin them on first touch: about 30 ms
// per GB), the DMA engine stays busy (one thread's memcpy runs at ~29 GB/s on the GPU box's host, PCIe takes
// ~75 GB/s from pinned memory), and the calling thread is free to queue kernels as soon as a stage is complete
// on the copy stream.  A downloader thread brings finished chunks' results back through a pinned slab while
// later chunks still run.  Small calls (< 8 MiB of input) do the same steps inline with plain pageable copies.
#pragma once
#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
#include <thread>
#include <type_traits>
#if !defined(__HIP_DEVICE_COMPILE__) && defined(__x86_64__)
#include <immintrin.h>
#endif

namespace gbx {

static double wall_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// A thread from a cache of idle ones: what std::thread is used for in the host entries (a few helpers for the length of one
// call or one pass), without creating threads per call - eighteen of them cost a 2 M-pair bsw call 0.5-0.7 ms of its 10.
// Same use as std::thread: construct from a callable, join().  The operating-system threads are detached; one that has finished
// its task waits for the next (at most 64 idle ones are kept, a surplus worker ends).  They run code of this library: a process
// must not unload libgbx.so (dlclose) or fork() and go on using it in the child while workers are parked.  A task always starts at once (an idle thread, else a new one), so
// tasks that wait for each other (upload workers, the downloader) cannot block one another out.
class Helper {
    struct Worker {
        std::mutex m;
        std::condition_variable cv;
        std::function<void()> job;
        bool has = false, done = false, quit = false;
        void loop()
        {
            for (;;) {
                std::unique_lock<std::mutex> lk(m);
                cv.wait(lk, [&] { return has || quit; });
                if (!has) { lk.unlock(); delete this; return; }      // surplus to the cache (join): nobody holds a pointer to it any more
                std::function<void()> f = std::move(job);
                has = false;
                lk.unlock();
                f();
                f = nullptr;
                lk.lock();
                done = true;
                cv.notify_all();
            }
        }
    };
    struct Cache { std::mutex mu; std::vector<Worker *> idle; };
    static Cache &cache() { static Cache *c = new Cache(); return *c; }      // never destroyed: its threads outlive static destruction
    // idle workers kept for the next task; beyond that a worker that finishes its task exits (a process whose many caller threads
    // once ran the host entries at the same time would otherwise keep its peak helper count for good)
    static size_t cache_cap()
    {
        static const size_t cap = getenv("GBX_HELPER_CACHE") ? (size_t)atoll(getenv("GBX_HELPER_CACHE")) : 64;      // (the TSan harness sets a small one)
        return cap;
    }
    Worker *w = nullptr;
    bool failed_ = false;

public:
    Helper() = default;
    // inline_on_failure: a task that does not wait for other tasks (a slice of a parallel loop) runs on the calling thread when no
    // thread can be started; otherwise failed() is set and nothing has run (the pipeline's workers wait for each other)
    template <class F, class = typename std::enable_if<!std::is_same<typename std::decay<F>::type, Helper>::value>::type>
    explicit Helper(F &&f, bool inline_on_failure = false)
    {
        Cache &c = cache();
        {
            std::lock_guard<std::mutex> lk(c.mu);
            if (!c.idle.empty()) { w = c.idle.back(); c.idle.pop_back(); }
        }
        if (!w) {
            Worker *n = nullptr;
            try {
                n = new Worker();
                std::thread([n] { n->loop(); }).detach();
            } catch (...) {                                       // no thread, or no memory for it: never through an extern "C" entry
                delete n;
                if (inline_on_failure) f(); else failed_ = true;
                return;
            }
            w = n;
        }
        {
            std::lock_guard<std::mutex> lk(w->m);
            w->job = std::forward<F>(f);
            w->has = true;
            w->done = false;
        }
        w->cv.notify_all();
    }
    Helper(Helper &&o) noexcept : w(o.w), failed_(o.failed_) { o.w = nullptr; }
    Helper &operator=(Helper &&o) noexcept { if (this != &o) { join(); w = o.w; failed_ = o.failed_; o.w = nullptr; } return *this; }
    Helper(const Helper &) = delete;
    Helper &operator=(const Helper &) = delete;
    bool joinable() const { return w != nullptr; }
    bool failed() const { return failed_; }
    void join()
    {
        if (!w) return;
        {
            std::unique_lock<std::mutex> lk(w->m);
            w->cv.wait(lk, [&] { return w->done; });
        }
        Cache &c = cache();
        bool keep;
        {
            std::lock_guard<std::mutex> lk(c.mu);
            keep = c.idle.size() < cache_cap();
            if (keep) c.idle.push_back(w);
        }
        if (!keep) {                                               // told to go while we hold its lock; it frees itself, we let go of it
            std::lock_guard<std::mutex> lk(w->m);
            w->quit = true;
            w->cv.notify_all();
        }
        w = nullptr;
    }
    ~Helper() { join(); }
};

// fn(t, lo, hi) over [0, n) split into `threads` contiguous ranges (helper threads; the caller's thread takes
// range 0)
template <class F> static void parallel_ranges(int64_t n, int threads, F fn)
{
    if (threads > n / 4096) threads = (int)(n / 4096);
    if (threads <= 1) { fn(0, (int64_t)0, n); return; }
    std::vector<Helper> th;
    for (int t = 1; t < threads; ++t) th.emplace_back([=] { fn(t, n * t / threads, n * (t + 1) / threads); }, true);
    fn(0, (int64_t)0, n / threads);
    for (auto &x : th) x.join();
}

// Two base codes per byte for the PCIe leg of gbx_bsw_extend_host (codes 0..4; anything larger is clamped to 15, which
// the kernels treat like 4, the ambiguous base): byte k of the output = code 2k | code 2k+1 << 4.  The upload workers
// do this instead of their memcpy into the pinned slab, so the DMA moves half the bytes; bsw_unpack4 expands them on
// the device.  n may be odd (the last byte then holds one code).
static void pack4_scalar(uint8_t *d, const uint8_t *s, size_t n)
{
    size_t i = 0;
    for (; i + 1 < n; i += 2) {
        const unsigned a = s[i] < 15 ? s[i] : 15, b = s[i + 1] < 15 ? s[i + 1] : 15;
        d[i >> 1] = (uint8_t)(a | (b << 4));
    }
    if (i < n) d[i >> 1] = (uint8_t)(s[i] < 15 ? s[i] : 15);
}
#if !defined(__HIP_DEVICE_COMPILE__) && defined(__x86_64__)
__attribute__((target("avx2"))) static void pack4_avx2(uint8_t *d, const uint8_t *s, size_t n)
{
    const __m256i m15 = _mm256_set1_epi8(15), mul = _mm256_set1_epi16(0x1001);   // bytes (1, 16): lo + 16 * hi per pair
    size_t i = 0;
    for (; i + 64 <= n; i += 64) {
        const __m256i a = _mm256_min_epu8(_mm256_loadu_si256((const __m256i *)(s + i)), m15);
        const __m256i b = _mm256_min_epu8(_mm256_loadu_si256((const __m256i *)(s + i + 32)), m15);
        const __m256i r = _mm256_packus_epi16(_mm256_maddubs_epi16(a, mul), _mm256_maddubs_epi16(b, mul));
        _mm256_storeu_si256((__m256i *)(d + (i >> 1)), _mm256_permute4x64_epi64(r, 0xD8));
    }
    pack4_scalar(d + (i >> 1), s + i, n - i);
}
#endif
static void pack4(uint8_t *d, const uint8_t *s, size_t n)
{
#if !defined(__HIP_DEVICE_COMPILE__) && defined(__x86_64__)
    static const bool avx2 = __builtin_cpu_supports("avx2");
    if (avx2) { pack4_avx2(d, s, n); return; }
#endif
    pack4_scalar(d, s, n);
}

struct DevBlock { void *p; size_t cap; };

struct Lane {
    std::vector<DevBlock> dev_cache;       // device blocks of earlier calls of this lane, reused by DevBuf
    static constexpr int MAX_WORKERS = 8;
#ifndef GBX_PIECE_MB
#define GBX_PIECE_MB 8
#endif
    static constexpr size_t PIECE = (size_t)GBX_PIECE_MB << 20;   // upload piece = worker slab
    static constexpr size_t DOWN = (size_t)16 << 20;        // download slab
    int dev = 0;
    // One stream of ordinary priority only (`compute`).  The runtime maps the ordinary streams of a process onto four
    // hardware queues, and streams that share a queue run in order: compute + three of the kernels' side streams have a
    // queue each (measured with more: two kernel streams lost their overlap).  The transfer streams are urgent ones, which
    // get hardware queues of their own class: two for the uploads - their DMAs fill each other's gaps - and one for the
    // downloads instead of queueing between the uploads (without stream priorities all three are one ordinary stream,
    // and its DMAs' marker packets wait behind kernel launches: DESIGN section 1).  Calls that do not overlap transfer and
    // compute use `compute` for everything.
    hipStream_t compute = nullptr, copy = nullptr, copy2 = nullptr, down = nullptr;
    hipEvent_t ev_stage = nullptr, ev_stage2 = nullptr, ev_half[2] = {nullptr, nullptr};
    hipEvent_t ev_part = nullptr, ev_pre = nullptr, ev_aux = nullptr;      // bsw: a chunk's index arrays are up; its preparing passes are done
    static constexpr int JOIN_EVENTS = 1 + SideStreams::N;
    std::vector<hipEvent_t> ev_chunk;      // JOIN_EVENTS per pipeline chunk of a call, grown on demand
    char *wslab[MAX_WORKERS][2] = {};
    hipEvent_t wev[MAX_WORKERS][2] = {};
    char *dslab = nullptr;
    bool staged_ready = false;
};

// Events a host thread waits on (a worker for its slab's last DMA, the downloader for a chunk's kernels).  GBX_EVENT_BLOCKING=1:
// the waiting thread sleeps until the interrupt instead of polling - a tuning aid for hosts where the pollers exhaust a
// CPU quota (measured on the pool's 16-core cgroups: see DESIGN §8).
static unsigned host_wait_event_flags()
{
    static const bool blocking = getenv("GBX_EVENT_BLOCKING") && atoi(getenv("GBX_EVENT_BLOCKING")) != 0;
    return hipEventDisableTiming | (blocking ? hipEventBlockingSync : 0u);
}

// dflt: a kernel's own default (bsw: 4 - measured on the pool's 16-core quota, profiles/r06zj_bsw_host_threads.txt: medians 8.75-8.84 ms
// with four workers, 8.95-9.37 with six, worse with three or five: the workers alternate between the two upload streams)
static int host_workers(int dflt = 6)
{
    const char *env = getenv("GBX_HOST_THREADS");
    int t = env ? atoi(env) : dflt;
    return t < 1 ? 1 : t > Lane::MAX_WORKERS ? Lane::MAX_WORKERS : t;
}

// pinned slabs of a lane: allocated and touched once (the pages are pinned by hipHostMalloc but only mapped into
// this process on the first write, ~0.1 ms per MB)
static int lane_prepare_staging(Lane *l)
{
    if (l->staged_ready) return GBX_OK;
    for (int w = 0; w < Lane::MAX_WORKERS; ++w)
        for (int k = 0; k < 2; ++k) {
            if (!l->wslab[w][k]) GBX_HIP(hipHostMalloc((void **)&l->wslab[w][k], Lane::PIECE, hipHostMallocDefault));
            if (!l->wev[w][k]) GBX_HIP(hipEventCreateWithFlags(&l->wev[w][k], host_wait_event_flags()));
        }
    if (!l->dslab) GBX_HIP(hipHostMalloc((void **)&l->dslab, Lane::DOWN, hipHostMallocDefault));
    std::vector<Helper> th;
    for (int w = 0; w < Lane::MAX_WORKERS; ++w)
        th.emplace_back([=] { memset(l->wslab[w][0], 0, Lane::PIECE); memset(l->wslab[w][1], 0, Lane::PIECE); }, true);
    memset(l->dslab, 0, Lane::DOWN);
    for (auto &x : th) x.join();
    // one small DMA from every slab on the stream its worker will use, one to the download slab, an event behind each: what
    // the runtime sets up at the first use of a stream or of a slab by a DMA engine (measured: 15 ms over the first two
    // large calls of a process) is paid here, not by a call
    if (l->copy && l->copy != l->compute) {
        void *scratch = nullptr;
        GBX_HIP(hipMalloc(&scratch, (size_t)3 << 20));            // a megabyte per stream: nothing here touches the same bytes twice at once
        hipError_t e = hipSuccess;
        for (int w = 0; w < Lane::MAX_WORKERS && e == hipSuccess; ++w)
            for (int k = 0; k < 2 && e == hipSuccess; ++k) {
                const hipStream_t xw = (w & 1) ? l->copy2 : l->copy;
                e = hipMemcpyAsync((char *)scratch + ((size_t)(w & 1) << 20), l->wslab[w][k], (size_t)1 << 20, hipMemcpyHostToDevice, xw);
                if (e == hipSuccess) e = hipEventRecord(l->wev[w][k], xw);
            }
        if (e == hipSuccess) e = hipMemsetAsync((char *)scratch + ((size_t)2 << 20), 0, (size_t)1 << 20, l->down);
        if (e == hipSuccess) e = hipMemcpyAsync(l->dslab, (char *)scratch + ((size_t)2 << 20), (size_t)1 << 20, hipMemcpyDeviceToHost, l->down);
        if (e == hipSuccess) e = hipEventRecord(l->ev_half[0], l->down);
        for (hipStream_t x : {l->copy, l->copy2, l->down}) { const hipError_t e2 = hipStreamSynchronize(x); if (e == hipSuccess) e = e2; }
        (void)hipFree(scratch);
        if (e != hipSuccess) return hip_fail(e, "host lane warm-up");
    }
    l->staged_ready = true;
    return GBX_OK;
}

struct HostLane {
    Lane *l = nullptr;
    static std::mutex &mu() { static std::mutex m; return m; }
    static std::vector<Lane *> &idle() { static std::vector<Lane *> v; return v; }
    int acquire()
    {
        int dev = 0;
        GBX_HIP(hipGetDevice(&dev));
        {
            std::lock_guard<std::mutex> lk(mu());
            auto &v = idle();
            for (size_t k = 0; k < v.size(); ++k)
                if (v[k]->dev == dev) { l = v[k]; v.erase(v.begin() + (long)k); return GBX_OK; }
        }
        // The runtime maps streams onto its four hardware queues in the order they are created, and streams that share a
        // queue run in order.  The kernels of a call run on the lane's compute stream and the device's three side streams:
        // the side streams are created first, then compute, so that in a process of its own those four get a queue each
        // (the copy stream then shares one with a side stream, which the pipelined bsw path allows for by using three
        // kernel streams).  A process that holds other streams of its own shifts the mapping: inside bench.py (torch's
        // streams) phmm's four class kernels, 125-185 ms each, lose their overlap in the host entry (283 ms against 217
        // standalone) - a property of the calling process, not of this pipeline.
        {
            SideStreams *ss = nullptr;
            const int rc = side_streams(&ss);
            if (rc) return rc;
        }
        Lane *n = new Lane();
        n->dev = dev;
        hipError_t e = hipStreamCreateWithFlags(&n->compute, hipStreamNonBlocking);
        // the copy stream at high priority: the runtime gives such a stream a hardware queue of its own class, so that the marker
        // packets behind its DMAs (the events the upload workers wait on before they reuse a slab) are not queued behind a
        // kernel stream's long launches (round 5: uploads of chunk k+1 stood still while chunk k's kernels ran)
        if (e == hipSuccess) {
            int least = 0, greatest = 0;
            const bool hi = !(getenv("GBX_COPY_PRIO") && atoi(getenv("GBX_COPY_PRIO")) == 0);
            if (hi && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && greatest < least) {
                e = hipStreamCreateWithPriority(&n->copy, hipStreamNonBlocking, greatest);
                const int ups = getenv("GBX_COPY_STREAMS") ? atoi(getenv("GBX_COPY_STREAMS")) : 2;
                if (e == hipSuccess && ups >= 2) e = hipStreamCreateWithPriority(&n->copy2, hipStreamNonBlocking, greatest);
                if (e == hipSuccess && !(getenv("GBX_DOWN_STREAM") && atoi(getenv("GBX_DOWN_STREAM")) == 0))
                    e = hipStreamCreateWithPriority(&n->down, hipStreamNonBlocking, greatest);
            } else { (void)hipGetLastError(); e = hipStreamCreateWithFlags(&n->copy, hipStreamNonBlocking); }
            if (!n->copy2) n->copy2 = n->copy;
            if (!n->down) n->down = n->copy;
        }
        if (e == hipSuccess) e = hipEventCreateWithFlags(&n->ev_stage, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&n->ev_stage2, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&n->ev_part, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&n->ev_pre, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&n->ev_aux, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&n->ev_half[0], host_wait_event_flags());
        if (e == hipSuccess) e = hipEventCreateWithFlags(&n->ev_half[1], host_wait_event_flags());
        if (e != hipSuccess) { delete n; return hip_fail(e, "host lane"); }
        l = n;
        return GBX_OK;
    }
    ~HostLane()
    {
        if (!l) return;
        std::lock_guard<std::mutex> lk(mu());
        idle().push_back(l);
    }
};

// Device buffer of one *_host call.  Blocks come from the lane's cache when one fits (at least the size asked
// for, at most twice that plus 1 MiB) and go back to it afterwards: a driver that calls per 512-pair batch, as the
// reference's does, would otherwise spend most of each call in nine hipMalloc/hipFree pairs (hipFree also
// synchronises the device).  gbx_host_release() frees the caches of idle lanes.  Declare after the HostLane.
struct DevBuf {
    Lane *L;
    void *p = nullptr;
    size_t cap = 0;
    explicit DevBuf(Lane *l) : L(l) {}
    DevBuf(const DevBuf &) = delete;
    ~DevBuf() { if (p) L->dev_cache.push_back(DevBlock{p, cap}); }
    int alloc(size_t bytes)
    {
        bytes = (bytes + 64 + 255) & ~(size_t)255;      // 64 B slack: kernels may read a few bytes past the last base
        auto &c = L->dev_cache;
        size_t best = c.size();
        for (size_t k = 0; k < c.size(); ++k)
            if (c[k].cap >= bytes && c[k].cap <= 2 * bytes + ((size_t)1 << 20) && (best == c.size() || c[k].cap < c[best].cap))
                best = k;
        if (best < c.size()) {
            p = c[best].p; cap = c[best].cap;
            c.erase(c.begin() + (long)best);
            return GBX_OK;
        }
        size_t held = 0;
        for (const DevBlock &b : c) held += b.cap;
        if (c.size() >= 48 || held > cache_cap()) {      // sizes keep changing: do not hoard
            for (DevBlock &b : c) (void)hipFree(b.p);
            c.clear();
        }
        hipError_t e = hipMalloc(&p, bytes);
        if (e != hipSuccess) {                           // out of memory with a full cache: drop it and retry once
            (void)hipGetLastError();
            for (DevBlock &b : c) (void)hipFree(b.p);
            c.clear();
            e = hipMalloc(&p, bytes);
        }
        if (e != hipSuccess) { p = nullptr; return hip_fail(e, "hipMalloc"); }
        cap = bytes;
        return GBX_OK;
    }
    template <class T> T *as() { return (T *)p; }
    // idle blocks a lane may hold before a miss empties its cache: 8 GiB of the 288 GB, GBX_HOST_CACHE_MB overrides
    static size_t cache_cap()
    {
        const char *env = getenv("GBX_HOST_CACHE_MB");
        return env ? (size_t)atoll(env) << 20 : (size_t)8 << 30;
    }
};

// A few helper threads that live as long as one staged call and copy pieces of the pinned download slab to the caller's
// memory (one core moves about 10 GB/s, the DMA five times that).  They used to be created per delivered piece: five
// thread creations for 8 MB, i.e. about as long as the piece's DMA.  run() hands a list of copy jobs to the helpers and
// the calling thread and returns when all are done; only the downloader thread calls it.
struct CopyPool {
    struct Job { char *dst; const char *src; size_t n; };
    std::vector<Helper> th;
    std::mutex mu;
    std::condition_variable cv, done_cv;
    const std::vector<Job> *jobs = nullptr;
    size_t next = 0, pending = 0;
    uint64_t round = 0;
    bool stop = false;
    void start(int helpers)
    {
        for (int t = 0; t < helpers; ++t) th.emplace_back([this] { loop(); });
    }
    void loop()
    {
        uint64_t seen = 0;
        for (;;) {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return stop || (round != seen && jobs && next < jobs->size()); });
            if (stop) return;
            seen = round;
            work(lk);
        }
    }
    // takes jobs until the list is empty; lk is held on entry and on exit
    void work(std::unique_lock<std::mutex> &lk)
    {
        while (jobs && next < jobs->size()) {
            const Job j = (*jobs)[next++];
            lk.unlock();
            memcpy(j.dst, j.src, j.n);
            lk.lock();
            if (--pending == 0) done_cv.notify_all();
        }
    }
    void run(const std::vector<Job> &list)
    {
        if (list.empty()) return;
        if (th.empty()) { for (const Job &j : list) memcpy(j.dst, j.src, j.n); return; }
        std::unique_lock<std::mutex> lk(mu);
        jobs = &list; next = 0; pending = list.size(); ++round;
        cv.notify_all();
        work(lk);
        done_cv.wait(lk, [&] { return pending == 0; });
        jobs = nullptr;
    }
    void shutdown()
    {
        { std::lock_guard<std::mutex> lk(mu); stop = true; }
        cv.notify_all();
        for (auto &t : th) t.join();
        th.clear();
        stop = false;
    }
    ~CopyPool() { shutdown(); }
};

// One *_host call's transfers.  Usage:
//   HostPipe P(lane, total_upload_bytes);  P.prepare(n_chunks);
//   P.stage(c, dst, src, bytes) ... for every chunk c in order;  P.start();
//   for c: P.wait_stage(c) -> compute stream waits for chunk c's uploads; launch kernels (join_events(c));
//          P.fetch(c, host_results, d_results, bytes) ...; P.chunk_launched(c)  -> the downloader fetches them when
//          the chunk is done
//   P.finish()  -> joins the threads, returns the first error
struct HostPipe {
    // pack: len source bytes -> (len+1)/2 at dst; stride != 0: len / 4 four-byte fields, `stride` bytes apart at src, -> len bytes
    struct Piece { char *dst; const char *src; size_t len; int chunk; bool pack; int stride; };
    // a contiguous device range -> one host range, or (segs) -> consecutive pieces of it to different host addresses
    struct Seg { char *dst; size_t len; };
    struct Fetch { void *dst; const void *src; size_t len; int chunk; const std::vector<Seg> *segs; };
    Lane *L;
    bool staged;
    int workers;
    std::vector<Piece> pieces;
    std::vector<int> remaining;            // pieces of chunk c not yet queued on the copy stream
    std::vector<Fetch> fetches;
    std::deque<std::vector<char>> gathered;    // compact copies of strided pieces of calls too small to stage (alive until the pipe goes)
    std::vector<int> chunk_nev;            // join events recorded for chunk c
    hipStream_t xfer;                      // L->copy, or L->compute for calls that do not overlap
    hipStream_t xfer2, xdown;              // the second upload stream and the download stream (the same rule)
    std::atomic<size_t> next{0};
    std::atomic<int> hip_err{0};
    bool abort_ = false;
    int64_t launched = 0, n_chunks = 0;
    std::mutex mu;
    std::condition_variable cv;
    std::vector<Helper> threads;
    CopyPool pool;                         // the downloader's copy helpers (staged calls)
    bool started = false;
    bool open_ = false;                    // staged calls: more pieces may be staged after start() (keep_open() .. seal())

    // stage*() after start() (keep_open() calls only): the workers are reading the list
    struct StageLock {
        HostPipe *p;
        explicit StageLock(HostPipe *q) : p(q->started ? q : nullptr) { if (p) p->mu.lock(); }
        ~StageLock() { if (p) { p->mu.unlock(); p->cv.notify_all(); } }
    };
    static size_t stage_min()
    {
        const char *env = getenv("GBX_HOST_STAGE_MIN");      /* bytes; the tests set 0 to stage small inputs too */
        return env ? (size_t)atoll(env) : (size_t)8 << 20;
    }
    // small calls on a prepared lane: the pieces are packed into one pinned slab by the calling thread and sent
    // from there (a pageable hipMemcpyAsync of a few KB is a blocking staged copy of ~20 us each, and a call has
    // seven to twelve of them); results come back through the pinned download slab
    bool packed;
    size_t pack_off = 0;
    HostPipe(Lane *l, size_t total_bytes, bool overlap, int default_workers = 6)
        : L(l), staged(total_bytes >= stage_min() && !getenv("GBX_HOST_PAGEABLE")), workers(host_workers(default_workers)),
          xfer(overlap ? l->copy : l->compute), xfer2(overlap ? l->copy2 : l->compute), xdown(overlap ? l->down : l->compute),
          packed(!staged && l->staged_ready && total_bytes <= Lane::PIECE / 2 && !getenv("GBX_HOST_PAGEABLE")) {}
    // an early return between start() and the last chunk_launched() must not leave the downloader waiting for a chunk
    // that will never be launched: such a pipe is cancelled, not finished
    ~HostPipe()
    {
        bool unlaunched;
        { std::lock_guard<std::mutex> lk(mu); unlaunched = started && staged && launched < n_chunks; }
        (void)finish(unlaunched ? GBX_ERR_HIP : GBX_OK);
    }

    int prepare(int64_t chunks)
    {
        n_chunks = chunks;
        remaining.assign((size_t)chunks, 0);
        chunk_nev.assign((size_t)chunks, 0);
        while ((int64_t)L->ev_chunk.size() < chunks * Lane::JOIN_EVENTS) {
            hipEvent_t e = nullptr;
            GBX_HIP(hipEventCreateWithFlags(&e, host_wait_event_flags()));
            L->ev_chunk.push_back(e);
        }
        return staged ? lane_prepare_staging(L) : GBX_OK;
    }
    void stage(int64_t chunk, void *dst, const void *src, size_t bytes)
    {
        StageLock lk(this);
        char *d = (char *)dst;
        const char *s = (const char *)src;
        const size_t cap = staged ? Lane::PIECE : bytes;
        while (bytes) {
            const size_t len = bytes < cap ? bytes : cap;
            pieces.push_back(Piece{d, s, len, (int)chunk, false, 0});
            ++remaining[(size_t)chunk];
            d += len; s += len; bytes -= len;
        }
    }
    // like stage(), for base codes that travel two per byte (staged calls only; src_off = offset of src in its arena,
    // even): byte k of the arena's packed image on the device holds codes 2k and 2k+1
    void stage_pack4(int64_t chunk, void *dst_packed, const void *src, size_t bytes)
    {
        StageLock lk(this);
        char *d = (char *)dst_packed;
        const char *s = (const char *)src;
        while (bytes) {
            const size_t len = bytes < Lane::PIECE ? bytes : Lane::PIECE;      // PIECE is even: only the last piece can be odd
            pieces.push_back(Piece{d, s, len, (int)chunk, true, 0});
            ++remaining[(size_t)chunk];
            d += len / 2; s += len; bytes -= len;
        }
    }
    // like stage(), for one 4-byte field of an array of records (abea: the mean of a 24-byte event): the upload workers
    // gather the field into the pinned slab instead of copying, so no compact host copy is ever made (staged calls only)
    void stage_field4(int64_t chunk, void *dst, const void *first_field, size_t n_records, int stride)
    {
        StageLock lk(this);
        char *d = (char *)dst;
        const char *s = (const char *)first_field;
        const size_t per = Lane::PIECE / 4;
        while (n_records) {
            const size_t k = n_records < per ? n_records : per;
            pieces.push_back(Piece{d, s, k * 4, (int)chunk, false, stride});
            ++remaining[(size_t)chunk];
            d += k * 4; s += k * (size_t)stride; n_records -= k;
        }
    }
    void fail_hip(hipError_t e)
    {
        int zero = 0;
        hip_err.compare_exchange_strong(zero, (int)e);
        std::lock_guard<std::mutex> lk(mu);
        abort_ = true;
        cv.notify_all();
    }
    void upload_worker(int w)
    {
        if (hipSetDevice(L->dev) != hipSuccess) { fail_hip(hipErrorInvalidDevice); return; }
        bool busy[2] = {false, false};
        int slot = 0;
        const hipStream_t xw = (w & 1) ? xfer2 : xfer;
        for (;;) {
            Piece p;
            {   // the next piece, or wait for one while the caller may still stage more (the vector can grow: a copy is taken)
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return abort_ || !open_ || next.load() < pieces.size(); });
                const size_t i = next.load();
                if (abort_ || i >= pieces.size() || hip_err.load()) break;
                next.store(i + 1);
                p = pieces[i];
            }
            hipError_t e = hipSuccess;
            if (busy[slot]) e = hipEventSynchronize(L->wev[w][slot]);
            if (e == hipSuccess) {
                RoctxRange range_("gbx:h2d piece (stage into pinned slab + DMA)");
                if (p.pack) pack4((uint8_t *)L->wslab[w][slot], (const uint8_t *)p.src, p.len);
                else if (p.stride) {
                    uint32_t *o = (uint32_t *)L->wslab[w][slot];
                    const char *q = p.src;
                    for (size_t k = 0, n = p.len / 4; k < n; ++k, q += p.stride) memcpy(o + k, q, 4);
                } else memcpy(L->wslab[w][slot], p.src, p.len);
                e = hipMemcpyAsync(p.dst, L->wslab[w][slot], p.pack ? (p.len + 1) / 2 : p.len, hipMemcpyHostToDevice, xw);
            }
            if (e == hipSuccess) e = hipEventRecord(L->wev[w][slot], xw);
            if (e != hipSuccess) { fail_hip(e); return; }
            busy[slot] = true;
            slot ^= 1;
            std::lock_guard<std::mutex> lk(mu);
            if (--remaining[(size_t)p.chunk] == 0) cv.notify_all();
        }
    }
    // D2H of one chunk's results behind its join event, through the two halves of the pinned slab when staged
    // (the copy of one half to the caller's memory overlaps the DMA into the other)
    // slab -> caller memory; large pieces with a few threads (one core moves about 10 GB/s, the DMA five times that)
    void copy_out(char *dst, const char *src, size_t len)
    {
        const int T = (int)pool.th.size() + 1;
        if (len < ((size_t)2 << 20) || T <= 1) { memcpy(dst, src, len); return; }
        const size_t per = (len / (size_t)T + 63) & ~(size_t)63;
        std::vector<CopyPool::Job> jobs;
        for (size_t a = 0; a < len; a += per) jobs.push_back(CopyPool::Job{dst + a, src + a, a + per > len ? len - a : per});
        pool.run(jobs);
    }
    // slab bytes [0, len) = bytes [pos, pos + len) of a packed stream -> the segments they belong to (cursor: first segment
    // not yet complete and the bytes of it already delivered)
    void copy_out_scatter(const char *slab, size_t len, const std::vector<Seg> &segs, size_t &si, size_t &so)
    {
        // jobs of at most 256 KB, so that a few long segments do not serialise the copy (the helpers take jobs one by one)
        std::vector<CopyPool::Job> jobs;
        size_t off = 0;
        while (off < len && si < segs.size()) {
            const size_t n = segs[si].len - so < len - off ? segs[si].len - so : len - off;
            for (size_t d = 0; d < n; d += (size_t)256 << 10)
                jobs.push_back(CopyPool::Job{segs[si].dst + so + d, slab + off + d, n - d < ((size_t)256 << 10) ? n - d : (size_t)256 << 10});
            off += n; so += n;
            if (so == segs[si].len) { ++si; so = 0; }
        }
        if (len < ((size_t)2 << 20)) { for (const CopyPool::Job &j : jobs) memcpy(j.dst, j.src, j.n); return; }
        pool.run(jobs);
    }
    hipError_t fetch_chunk(int64_t c)
    {
        RoctxRange range_("gbx:d2h chunk (wait for the chunk's kernels + DMA + copy out)");
        hipError_t e = hipSuccess;
        for (int k = 0; k < chunk_nev[(size_t)c]; ++k)        // the chunk's kernels are done on every stream they ran on
            if ((e = hipEventSynchronize(L->ev_chunk[(size_t)c * Lane::JOIN_EVENTS + (size_t)k])) != hipSuccess) return e;
        std::vector<Fetch> mine;           // the launcher thread may be appending later chunks' entries
        {
            std::lock_guard<std::mutex> lk(mu);
            for (const Fetch &f : fetches) if (f.chunk == c) mine.push_back(f);
        }
        if (!staged) {
            size_t total = 0;
            for (const Fetch &f : mine) total += (f.len + 63) & ~(size_t)63;
            if (packed && total <= Lane::DOWN) {              // into the pinned slab, one wait, then out to the caller
                size_t off = 0;
                for (const Fetch &f : mine) {
                    if (f.len && (e = hipMemcpyAsync(L->dslab + off, f.src, f.len, hipMemcpyDeviceToHost, xfer)) != hipSuccess) return e;
                    off += (f.len + 63) & ~(size_t)63;
                }
                if ((e = hipStreamSynchronize(xfer)) != hipSuccess) return e;
                off = 0;
                for (const Fetch &f : mine) { memcpy(f.dst, L->dslab + off, f.len); off += (f.len + 63) & ~(size_t)63; }
                return hipSuccess;
            }
            for (const Fetch &f : mine)
                if (f.len && (e = hipMemcpyAsync(f.dst, f.src, f.len, hipMemcpyDeviceToHost, xfer)) != hipSuccess)
                    return e;
            return hipStreamSynchronize(xfer);
        }
        size_t HALF = Lane::DOWN / 2;
        if (const char *env = getenv("GBX_HOST_DOWN_PIECE")) {      /* bytes; the tests shrink it so that small jobs span pieces */
            const size_t v = (size_t)atoll(env) & ~(size_t)63;
            if (v >= 64 && v < HALF) HALF = v;
        }
        char *pend_dst = nullptr; size_t pend_len = 0; int half = 0;
        const std::vector<Seg> *pend_segs = nullptr;
        size_t si = 0, so = 0;                                 // cursor of the scatter plan being delivered
        auto deliver = [&](int h) {
            if (pend_segs) copy_out_scatter(L->dslab + h * HALF, pend_len, *pend_segs, si, so);
            else copy_out(pend_dst, L->dslab + h * HALF, pend_len);
        };
        for (const Fetch &f : mine) {
            char *d = (char *)f.dst;
            const char *s = (const char *)f.src;
            for (size_t left = f.len; left;) {
                const size_t len = left < HALF ? left : HALF;
                if ((e = hipMemcpyAsync(L->dslab + half * HALF, s, len, hipMemcpyDeviceToHost, xdown)) != hipSuccess) return e;
                if ((e = hipEventRecord(L->ev_half[half], xdown)) != hipSuccess) return e;
                if (pend_len) {
                    if ((e = hipEventSynchronize(L->ev_half[half ^ 1])) != hipSuccess) return e;
                    deliver(half ^ 1);
                }
                if (f.segs && pend_segs != f.segs) { si = 0; so = 0; }      // (the plan's first piece: after the previous delivery)
                pend_dst = d; pend_len = len; pend_segs = f.segs; half ^= 1;
                if (d) d += len;
                s += len; left -= len;
            }
        }
        if (pend_len) {
            if ((e = hipEventSynchronize(L->ev_half[half ^ 1])) != hipSuccess) return e;
            deliver(half ^ 1);
        }
        return hipSuccess;
    }
    void download_worker()
    {
        if (hipSetDevice(L->dev) != hipSuccess) { fail_hip(hipErrorInvalidDevice); return; }
        for (int64_t c = 0; c < n_chunks; ++c) {
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return launched > c || abort_; });
                if (abort_) return;
            }
            hipError_t e = fetch_chunk(c);
            if (e != hipSuccess) { fail_hip(e); return; }
        }
    }
    // Staged calls that stage their first chunk, start(), and stage the rest while it uploads (bsw: the later chunks' pairs
    // are validated meanwhile): keep_open() before start(), seal() after the last stage*() - and before wait_stage() of
    // any chunk staged after start().
    void keep_open() { open_ = staged; }
    void seal()
    {
        std::lock_guard<std::mutex> lk(mu);
        open_ = false;
        cv.notify_all();
    }
    void start()
    {
        started = true;
        if (!staged) return;
        int up_ok = 0;
        for (int w = 0; w < workers; ++w) { threads.emplace_back([this, w] { upload_worker(w); }); up_ok += threads.back().failed() ? 0 : 1; }
        pool.start((workers < 6 ? workers : 6) - 1);
        threads.emplace_back([this] { download_worker(); });
        // no thread to be had for the uploads or the downloads (fewer upload workers than asked for is fine): the call fails at
        // its next wait_stage() instead of waiting for ever
        if (up_ok == 0 || threads.back().failed()) fail_hip(hipErrorOutOfMemory);
    }
    // Upload stages finer than the chunks (after prepare(); default: one stage per chunk): stage*() and wait_stage() then
    // count stages, fetch() / join_events() / chunk_launched() chunks.  bsw: a chunk's index arrays and its bases.
    void upload_stages(int64_t k) { remaining.assign((size_t)k, 0); }
    // returns when every upload of stage c is queued on the copy stream, and makes the compute stream wait for them;
    // with `ev` (staged, overlapping calls only): records that event behind them instead, and no stream waits
    int wait_stage(int64_t c, hipEvent_t ev = nullptr)
    {
        if (!staged) {
            RoctxRange range_("gbx:h2d (direct)");
            for (; next < pieces.size() && pieces[next].chunk <= c; ++next) {
                const Piece &p = pieces[next];
                const void *src = p.src;
                if (p.stride) {                                  // a field of records: gathered here (small calls only)
                    gathered.emplace_back(p.len);
                    char *o = gathered.back().data();
                    const char *q = p.src;
                    for (size_t k = 0, n = p.len / 4; k < n; ++k, q += p.stride) memcpy(o + 4 * k, q, 4);
                    src = o;
                }
                if (packed && pack_off + p.len <= Lane::PIECE) {
                    char *slab = L->wslab[0][0] + pack_off;
                    memcpy(slab, src, p.len);
                    pack_off += (p.len + 63) & ~(size_t)63;
                    src = slab;
                }
                GBX_HIP(hipMemcpyAsync(p.dst, src, p.len, hipMemcpyHostToDevice, xfer));
            }
        } else {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return remaining[(size_t)c] == 0 || abort_; });
            if (abort_) return hip_fail((hipError_t)hip_err.load(), "host pipeline upload");
        }
        if (xfer != L->compute) {
            if (xfer2 != xfer) {                 // the first upload stream waits for the second: one event then stands for both
                GBX_HIP(hipEventRecord(L->ev_stage2, xfer2));
                GBX_HIP(hipStreamWaitEvent(xfer, L->ev_stage2, 0));
            }
            GBX_HIP(hipEventRecord(ev ? ev : L->ev_stage, xfer));
            if (!ev) GBX_HIP(hipStreamWaitEvent(L->compute, L->ev_stage, 0));
        }
        return GBX_OK;
    }
    // the event wait_stage() recorded behind the chunk's uploads (nullptr when they went on the compute stream itself)
    hipEvent_t stage_event() const { return xfer != L->compute ? L->ev_stage : nullptr; }
    // a result array of chunk c (call before chunk_launched(c))
    void fetch(int64_t c, void *host_dst, const void *dev_src, size_t bytes)
    {
        std::lock_guard<std::mutex> lk(mu);
        fetches.push_back(Fetch{host_dst, dev_src, bytes, (int)c, nullptr});
    }
    // a packed result array of chunk c whose consecutive pieces go to different places in the caller's memory (abea: a
    // read's pairs).  segs must stay alive until finish(); their lengths add up to bytes.  Staged calls only.
    void fetch_scatter(int64_t c, const void *dev_src, size_t bytes, const std::vector<Seg> *segs)
    {
        std::lock_guard<std::mutex> lk(mu);
        fetches.push_back(Fetch{nullptr, dev_src, bytes, (int)c, segs});
    }
    // the events a launch function records when chunk c's kernels are queued (one per stream they run on)
    hipEvent_t *join_events(int64_t c) { return &L->ev_chunk[(size_t)c * Lane::JOIN_EVENTS]; }
    // chunk c's kernels are queued; n_events of join_events(c) were recorded (0: everything is ordered on the
    // compute stream and one is recorded here)
    int chunk_launched(int64_t c, int n_events = 0)
    {
        if (n_events == 0) {
            GBX_HIP(hipEventRecord(join_events(c)[0], L->compute));
            n_events = 1;
        }
        chunk_nev[(size_t)c] = n_events;
        if (!staged) return GBX_OK;
        std::lock_guard<std::mutex> lk(mu);
        launched = c + 1;
        cv.notify_all();
        return GBX_OK;
    }
    void cancel()
    {
        std::lock_guard<std::mutex> lk(mu);
        abort_ = true;
        open_ = false;
        next = pieces.size();
        cv.notify_all();
    }
    // everything queued: wait for the transfers, return the first error.  Safe to call twice.
    int finish(int rc_in = GBX_OK)
    {
        if (!started) return rc_in;
        started = false;
        if (rc_in) cancel();
        else seal();
        for (auto &t : threads) t.join();
        threads.clear();
        pool.shutdown();
        int rc = rc_in;
        if (!rc && hip_err.load()) rc = hip_fail((hipError_t)hip_err.load(), "host pipeline");
        if (!rc && !staged) {
            for (int64_t c = 0; c < n_chunks && !rc; ++c) {
                hipError_t e = fetch_chunk(c);
                if (e != hipSuccess) rc = hip_fail(e, "host pipeline download");
            }
        }
        // nothing of this call may still be in flight when the caller's device buffers are freed
        (void)hipStreamSynchronize(L->copy); (void)hipStreamSynchronize(L->compute);
        if (L->copy2 != L->copy) (void)hipStreamSynchronize(L->copy2);
        if (L->down != L->copy) (void)hipStreamSynchronize(L->down);
        if (rc) (void)hipDeviceSynchronize();        // kernels on the shared side streams too
        return rc;
    }
};

// The pipeline chunks of gbx_bsw_extend_host: cuts[c] .. cuts[c + 1] are chunk c's pairs, every cut but the last a multiple
// of 64.  Every chunk is a full set of class kernels, and a launch needs several hundred thousand pairs to keep 256 CUs
// busy through the single-wavefront tails (measured: +0.7 ms per extra chunk at 2 M pairs), so: at most 3 equal chunks,
// none below 512 Ki pairs.  GBX_BSW_HOST_CHUNK=<pairs> (the tests vary it) cuts into equal chunks of that size,
// GBX_BSW_HOST_CUTS="0.2,0.6" at those fractions of the pairs (a small first chunk, so that the kernels start early, was
// measured no faster than thirds: 10.5 against 10.3 ms, profiles/r05ae_cuts.txt).
static std::vector<int64_t> bsw_host_cuts(int64_t n)
{
    std::vector<int64_t> cuts(1, 0);
    auto r64 = [](int64_t v) { return (v + 63) & ~63LL; };
    const char *env = getenv("GBX_BSW_HOST_CHUNK");      /* read per call */
    const char *frac = getenv("GBX_BSW_HOST_CUTS");
    if (env && atoll(env) > 0) {
        const int64_t c = r64(atoll(env));
        for (int64_t a = c; a < n; a += c) cuts.push_back(a);
    } else if (frac && *frac) {
        for (const char *q = frac; *q;) {
            const int64_t a = r64((int64_t)(atof(q) * (double)n));
            if (a > cuts.back() && a < n) cuts.push_back(a);
            while (*q && *q != ',') ++q;
            if (*q == ',') ++q;
        }
    } else {
        int64_t c = (n + 2) / 3;
        if (c < 524288) c = 524288;
        c = r64(c);
        for (int64_t a = c; a < n; a += c) cuts.push_back(a);
    }
    cuts.push_back(n);
    return cuts;
}

}  // namespace gbx
