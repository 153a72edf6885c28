// host_combine.h — concurrent small *_host calls combined into one device call (flat combining).
//
// The reference's drivers call their kernel once per small work unit from every OpenMP thread: bsw 512 pairs per
// getScores16 call (bsw/main_banded.cpp:279-291, run-cpu.sh:61 `-b 512`), phmm one computelikelihoodsboth per batch
// (phmm/PairHMMUnitTest.cpp:224-247), poa one window per generate_consensus (poa/msa_spoa_omp.cpp:230-260).  Behind a
// C entry point that is a GPU call each - upload, launch set, download, a few hundred microseconds whatever its size - and
// 64 caller threads bought nothing (round 5: the unmodified bsw driver at -t 64 -b 512 took 1.00 s on 2 M pairs, the
// reference's AVX2 code on the same cores 0.40 s, one call 0.15 s).
//
// Here a call below a kernel's threshold is *submitted*: the caller queues a request {pointers, sizes, result buffers}
// and blocks.  The first caller that finds no leader becomes one (no extra thread, no process, nothing re-executed): it
// takes everything that is pending and compatible (same device, same scoring parameters), lays the inputs end to end in
// scratch arrays (a few helper threads copy), issues ONE call of the kernel's ordinary host path - one packed upload, one
// launch set, one download -, hands every caller its slice of the results and wakes them; whoever is pending by then is
// led by one of the woken.  A call that meets nobody runs on its own arrays exactly as before (one mutex taken).  Up to
// two leaders work at a time (GBX_COMBINE_LEADERS): the host side of one combined call overlaps the device side of the other.
//
// Callers come in crowds (an OpenMP team leaves one combined call together and is back within microseconds of each other),
// but the first one back would lead alone and the crowd would wait behind its tiny call: a leader that finds fewer
// requests than recent calls had gives the others a moment to arrive (GBX_COMBINE_GATHER_US, default 150; the estimate
// decays, so a crowd that has gone costs a few waits).  Results are those of the separate calls by construction: work
// units are independent, and a combined call that fails is redone request by request, so that every caller gets the
// status and error text its own call would have produced.  GBX_COMBINE=0 switches it off; a thread that is collecting
// kernel timings (gbx_profile_begin) is never combined.
#pragma once
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <new>
#include <string>
#include <vector>

namespace gbx {

bool profile_active();                    // gbx_core.hip: the calling thread is between gbx_profile_begin and _end

struct CombineReq {
    int dev = 0;                          // the caller's current device: calls combine within one device only
    int64_t units = 0;                    // pairs / windows: what the cap on a combined call counts
    bool done = false, taken = false;
    int rc = GBX_OK;
    std::string err;
};

static bool combine_enabled()
{
    const char *e = getenv("GBX_COMBINE");              /* read per call: the tests vary it */
    return !(e && atoi(e) == 0);
}

struct Combiner {
    static constexpr int MAX_LEADERS = 4;
    std::mutex mu;
    std::condition_variable cv_done, cv_arrive;
    std::vector<CombineReq *> pending;
    bool slot_busy[MAX_LEADERS] = {false, false, false, false};
    int leaders = 0;
    double crowd = 1.0;                   // requests per combined call, lately (decaying maximum)
    // counters (gbx_host_combine_stats): calls submitted, device calls made, calls that shared one, most calls in one
    std::atomic<uint64_t> n_calls{0}, n_batches{0}, n_shared{0}, largest{0};

    static int gather_us()
    {
        const char *e = getenv("GBX_COMBINE_GATHER_US");
        const int v = e ? atoi(e) : 150;
        return v < 0 ? 0 : v > 100000 ? 100000 : v;
    }
    // Combined calls in flight at once.  Two: while one crowd's call is on the device the callers that arrive meanwhile are
    // laid out and sent by a leader of their own, on a lane of its own - the host side of one call (gathering, copying,
    // handing out results, the driver's own work between calls) overlaps the device side of the other.
    static int max_leaders(int dflt)
    {
        const char *e = getenv("GBX_COMBINE_LEADERS");
        const int v = e ? atoi(e) : dflt;
        return v < 1 ? 1 : v > MAX_LEADERS ? MAX_LEADERS : v;
    }

    // same(a, b): may b ride in a's call.  run(batch, slot): performs every request of batch (batch[0] is the leader's own) and
    // sets their rc / err; slot (0 .. MAX_LEADERS-1) is this leader's alone while it runs (scratch arrays are kept per slot);
    // it must not throw past bad_alloc.  Returns the caller's status with its error text set on its thread.
    template <class Same, class Run> int submit(CombineReq *r, int64_t max_units, int n_leaders, Same same, Run run)
    {
        std::unique_lock<std::mutex> lk(mu);
        n_calls.fetch_add(1, std::memory_order_relaxed);
        pending.push_back(r);
        cv_arrive.notify_all();
        while (!r->done) {
            if (r->taken || leaders >= n_leaders) { cv_done.wait(lk); continue; }
            // this caller leads: its request leaves the pool now, so that no other leader can take it meanwhile
            ++leaders;
            int slot = 0;
            while (slot_busy[slot]) ++slot;
            slot_busy[slot] = true;
            r->taken = true;
            for (size_t k = 0; k < pending.size(); ++k) if (pending[k] == r) { pending.erase(pending.begin() + (long)k); break; }
            const size_t want = (size_t)(crowd + 0.5);
            const int wait_us = gather_us();
            if (pending.size() + 1 < want && wait_us > 0)
                // (a deadline on the system clock: pthread_cond_timedwait, which ThreadSanitizer sees through - wait_for's
                // pthread_cond_clockwait it does not, and reports the mutex as held across the wait; a clock step only
                // shortens or lengthens one gathering pause)
                cv_arrive.wait_until(lk, std::chrono::system_clock::now() + std::chrono::microseconds(wait_us), [&] { return pending.size() + 1 >= want; });
            std::vector<CombineReq *> batch(1, r), rest;
            int64_t units = r->units;
            for (CombineReq *q : pending) {
                if (q->dev == r->dev && units + q->units <= max_units && same(r, q)) { batch.push_back(q); units += q->units; q->taken = true; }
                else rest.push_back(q);
            }
            pending.swap(rest);
            lk.unlock();
            try { run(batch, slot); }
            catch (const std::bad_alloc &) {
                for (CombineReq *q : batch) { q->rc = GBX_ERR_NOMEM; q->err = "out of host memory while combining host calls"; }
            }
            lk.lock();
            for (CombineReq *q : batch) q->done = true;
            --leaders;
            slot_busy[slot] = false;
            const double got = (double)batch.size();
            crowd = got > crowd * 0.75 ? got : crowd * 0.75;
            if (crowd < 1.0) crowd = 1.0;
            n_batches.fetch_add(1, std::memory_order_relaxed);
            if (batch.size() > 1) n_shared.fetch_add(batch.size(), std::memory_order_relaxed);
            if (batch.size() > largest.load(std::memory_order_relaxed)) largest.store(batch.size(), std::memory_order_relaxed);
            cv_done.notify_all();
        }
        lk.unlock();
        if (r->rc) set_error("%s", r->err.c_str());
        return r->rc;
    }
};

// the three kernels whose reference drivers call per small unit (capi_bsw / capi_phmm / capi_poa.hip)
Combiner &combiner_bsw();
Combiner &combiner_phmm();
Combiner &combiner_poa();

// fn(k) for k in [0, n) on up to `threads` threads (helpers + the caller), requests taken one by one
template <class F> static void combine_parallel(int64_t n, int threads, F fn)
{
    if (threads > n) threads = (int)n;
    if (threads <= 1) { for (int64_t k = 0; k < n; ++k) fn(k); return; }
    std::atomic<int64_t> next{0};
    auto body = [&] { for (int64_t k; (k = next.fetch_add(1, std::memory_order_relaxed)) < n;) fn(k); };
    std::vector<Helper> th;
    for (int t = 1; t < threads; ++t) th.emplace_back(body, true);      // (no thread to be had: the slice runs here)
    body();
    for (auto &x : th) x.join();
}

// An array the leader fills and reuses from call to call: grows, never shrinks, never zero-filled twice.
template <class T> struct Scratch {
    std::vector<T> v;
    T *get(size_t n) { if (v.size() < n) v.resize(n + n / 4 + 64); return v.data(); }
};

}  // namespace gbx
