// A host-only stand-in for the slice of the HIP runtime that csrc/host_pipeline.h uses, so that the host entries'
// threads / events / caches can run under -fsanitize=thread on a box without a GPU (tests/sanitize/pipe_tsan.cpp).
// TEST INFRASTRUCTURE ONLY - never built into libgbx.so.  Semantics kept: a stream is an in-order queue served by its
// own thread (so a hipMemcpyAsync really is asynchronous to its caller), an event completes when the stream reaches its
// record, hipStreamWaitEvent stalls the waiting stream, "device memory" is host memory.  mock_launch() queues a host
// function as a kernel; mock_fail_after(n) makes the n-th following hipMemcpyAsync fail (error paths).
#pragma once
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

enum hipError_t { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorInvalidDevice = 101, hipErrorUnknown = 999 };
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipEventBlockingSync = 1, hipHostMallocDefault = 0 };

struct MockStream {
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::function<void()>> q;
    bool stop = false, busy = false;
    std::thread th;
    MockStream() : th([this] { run(); }) {}
    ~MockStream() { { std::lock_guard<std::mutex> lk(mu); stop = true; } cv.notify_all(); th.join(); }
    void run()
    {
        for (;;) {
            std::function<void()> f;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return stop || !q.empty(); });
                if (q.empty()) return;
                f = std::move(q.front()); q.pop_front(); busy = true;
            }
            f();
            { std::lock_guard<std::mutex> lk(mu); busy = false; }
            cv.notify_all();
        }
    }
    void push(std::function<void()> f) { { std::lock_guard<std::mutex> lk(mu); q.push_back(std::move(f)); } cv.notify_all(); }
    void drain() { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return q.empty() && !busy; }); }
};
struct MockEvent {
    std::mutex mu;
    std::condition_variable cv;
    uint64_t recorded = 0, done = 0;
};
typedef MockStream *hipStream_t;
typedef MockEvent *hipEvent_t;

inline std::mutex &mock_registry_mu() { static std::mutex m; return m; }
inline std::vector<MockStream *> &mock_streams() { static std::vector<MockStream *> v; return v; }
inline std::atomic<long> &mock_fail_countdown() { static std::atomic<long> c{-1}; return c; }
inline void mock_fail_after(long n) { mock_fail_countdown().store(n); }
inline MockStream *mock_default_stream() { static MockStream *s = new MockStream(); return s; }
inline MockStream *mock_s(hipStream_t s) { return s ? s : mock_default_stream(); }

inline hipError_t hipGetDevice(int *d) { *d = 0; return hipSuccess; }
inline hipError_t hipSetDevice(int d) { return d == 0 ? hipSuccess : hipErrorInvalidDevice; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : "mock HIP error"; }
inline hipError_t hipMalloc(void **p, size_t n) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
inline hipError_t hipFree(void *p) { free(p); return hipSuccess; }
inline hipError_t hipHostMalloc(void **p, size_t n, unsigned) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
inline hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned)
{
    *s = new MockStream();
    std::lock_guard<std::mutex> lk(mock_registry_mu());
    mock_streams().push_back(*s);
    return hipSuccess;
}
inline hipError_t hipDeviceGetStreamPriorityRange(int *least, int *greatest) { *least = 0; *greatest = -1; return hipSuccess; }
inline hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned flags, int) { return hipStreamCreateWithFlags(s, flags); }
inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = new MockEvent(); return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
inline hipError_t hipMemcpyAsync(void *dst, const void *src, size_t n, hipMemcpyKind, hipStream_t s)
{
    long c = mock_fail_countdown().load();
    while (c >= 0 && !mock_fail_countdown().compare_exchange_weak(c, c - 1)) {}
    if (c == 0) return hipErrorUnknown;
    mock_s(s)->push([=] { memcpy(dst, src, n); });
    return hipSuccess;
}
inline hipError_t hipMemsetAsync(void *dst, int v, size_t n, hipStream_t s) { mock_s(s)->push([=] { memset(dst, v, n); }); return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t s)
{
    uint64_t gen;
    { std::lock_guard<std::mutex> lk(e->mu); gen = ++e->recorded; }
    mock_s(s)->push([e, gen] { { std::lock_guard<std::mutex> lk(e->mu); if (e->done < gen) e->done = gen; } e->cv.notify_all(); });
    return hipSuccess;
}
inline hipError_t hipEventSynchronize(hipEvent_t e)
{
    std::unique_lock<std::mutex> lk(e->mu);
    const uint64_t gen = e->recorded;
    e->cv.wait(lk, [&] { return e->done >= gen; });
    return hipSuccess;
}
inline hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned)
{
    uint64_t gen;
    { std::lock_guard<std::mutex> lk(e->mu); gen = e->recorded; }
    mock_s(s)->push([e, gen] { std::unique_lock<std::mutex> lk(e->mu); e->cv.wait(lk, [&] { return e->done >= gen; }); });
    return hipSuccess;
}
inline hipError_t hipStreamSynchronize(hipStream_t s) { mock_s(s)->drain(); return hipSuccess; }
inline hipError_t hipDeviceSynchronize()
{
    std::vector<MockStream *> v;
    { std::lock_guard<std::mutex> lk(mock_registry_mu()); v = mock_streams(); }
    for (MockStream *s : v) s->drain();
    mock_default_stream()->drain();
    return hipSuccess;
}
inline void mock_launch(hipStream_t s, std::function<void()> kernel) { mock_s(s)->push(std::move(kernel)); }
