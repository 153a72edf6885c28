// pipe_tsan.cpp — the host-entry plumbing of libgbx.so (csrc/host_pipeline.h: lanes, upload workers, the downloader,
// pinned slabs, device-block cache, events) compiled for the HOST against tests/sanitize/mock_hip and run under
// -fsanitize=thread: several caller threads drive staged / unstaged / multi-chunk / scatter / failing calls at once.
// TEST INFRASTRUCTURE ONLY.  The header under test is the product's own file, unmodified; what is mocked is the HIP
// runtime below it (streams = in-order worker threads, events, "device" memory = host memory).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <random>
#include <thread>
#include <vector>
#include "gbx_internal.h"

namespace gbx {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) { va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap); }
int hip_fail(hipError_t e, const char *what) { set_error("%s: %s", what, hipGetErrorString(e)); return GBX_ERR_HIP; }
int side_streams(SideStreams **out) { static SideStreams ss; *out = &ss; return GBX_OK; }      // not exercised here: the kernels are mock launches
RoctxRange::RoctxRange(const char *) : on_(false) {}
RoctxRange::~RoctxRange() {}
}  // namespace gbx

#include "host_pipeline.h"
// the multi-device layer of the host entries (csrc/host_multi.h, unmodified): the cut rule and the shard runner.  The
// mock runtime has one device; the logical devices of a call all map onto it, as GBX_DEVICE_MAP=0,0,0 does on a GPU box.
extern "C" const char *gbx_last_error(void) { return gbx::g_err; }
#include "host_multi.h"
namespace gbx {
int host_device_set(int *map) { for (int k = 0; k < 3; ++k) map[k] = 0; return 3; }
int host_next_small_call_device(int n) { static std::atomic<unsigned> rr{0}; return (int)(rr.fetch_add(1) % (unsigned)n); }
bool host_multi_wanted() { return true; }
bool profile_active() { return false; }
}
// the call combiner of the host entries (csrc/host_combine.h, unmodified)
#include "host_combine.h"

using namespace gbx;

static std::atomic<int> g_fail{0};
#define CHECK(cond, ...) do { if (!(cond)) { fprintf(stderr, "pipe_tsan: FAILED %s:%d: ", __FILE__, __LINE__); fprintf(stderr, __VA_ARGS__); \
                                             fprintf(stderr, "\n"); ++g_fail; } } while (0)

static std::vector<uint8_t> random_bytes(size_t n, uint64_t seed, int mod = 256)
{
    std::vector<uint8_t> v(n);
    std::mt19937_64 r(seed);
    for (size_t i = 0; i < n; ++i) v[i] = (uint8_t)(r() % (uint64_t)mod);
    return v;
}

// one call shaped like gbx_chain_host: two input arrays, one "kernel", two result arrays, one chunk
static int call_one_chunk(size_t n, uint64_t seed, bool expect_ok)
{
    std::vector<uint8_t> a = random_bytes(n, seed), b = random_bytes(n, seed + 1), sum(n, 0), dif(n, 0);
    HostLane lane;
    int rc = lane.acquire();
    if (rc) return rc;
    Lane *L = lane.l;
    DevBuf da(L), db(L), ds(L), dd(L);
    if ((rc = da.alloc(n)) || (rc = db.alloc(n)) || (rc = ds.alloc(n)) || (rc = dd.alloc(n))) return rc;
    HostPipe pipe(L, 2 * n, false);
    if ((rc = pipe.prepare(1))) return rc;
    pipe.stage(0, da.p, a.data(), n);
    pipe.stage(0, db.p, b.data(), n);
    pipe.start();
    if ((rc = pipe.wait_stage(0))) return pipe.finish(rc);
    uint8_t *pa = da.as<uint8_t>(), *pb = db.as<uint8_t>(), *ps = ds.as<uint8_t>(), *pd = dd.as<uint8_t>();
    mock_launch(L->compute, [=] { for (size_t i = 0; i < n; ++i) { ps[i] = (uint8_t)(pa[i] + pb[i]); pd[i] = (uint8_t)(pa[i] - pb[i]); } });
    pipe.fetch(0, sum.data(), ds.p, n);
    pipe.fetch(0, dif.data(), dd.p, n);
    if ((rc = pipe.chunk_launched(0))) return pipe.finish(rc);
    rc = pipe.finish();
    if (rc == GBX_OK && expect_ok)
        for (size_t i = 0; i < n; i += 1 + n / 4099) {
            CHECK(sum[i] == (uint8_t)(a[i] + b[i]) && dif[i] == (uint8_t)(a[i] - b[i]), "one-chunk result differs at %zu of %zu", i, n);
            if (g_fail) break;
        }
    return rc;
}

// shaped like gbx_bsw_extend_host: three chunks, transfers on the copy stream overlapping the kernels, bases packed two
// per byte on the way up, a second kernel stream per chunk with its own join event
// late: only the first chunk is staged before start(), the others while it uploads (keep_open() .. seal(), as the bsw entry
// does while it checks the later chunks' pairs); abandon_open: the call gives up before seal()
static int call_three_chunks(size_t per, uint64_t seed, bool late = false, bool abandon_open = false)
{
    const int C = 3;
    std::vector<uint8_t> base = random_bytes(per * C, seed, 5);
    std::vector<uint32_t> out(per * C, 0);
    HostLane lane;
    int rc = lane.acquire();
    if (rc) return rc;
    Lane *L = lane.l;
    DevBuf dpacked(L), dout(L);
    if ((rc = dpacked.alloc(per * C / 2 + 8)) || (rc = dout.alloc(per * C * 4))) return rc;
    HostPipe pipe(L, per * C, true);
    if ((rc = pipe.prepare(C))) return rc;
    for (int c = 0; c < (late ? 1 : C); ++c) pipe.stage_pack4(c, dpacked.as<uint8_t>() + (size_t)c * per / 2, base.data() + (size_t)c * per, per);
    if (late) pipe.keep_open();
    pipe.start();
    if (late) {
        std::this_thread::sleep_for(std::chrono::microseconds(200 + seed % 700));       // the workers run dry and wait
        pipe.stage_pack4(1, dpacked.as<uint8_t>() + per / 2, base.data() + per, per);
        if (abandon_open) return pipe.finish(GBX_ERR_ARG);
        pipe.stage_pack4(2, dpacked.as<uint8_t>() + per, base.data() + 2 * per, per);
        pipe.seal();
    }
    for (int c = 0; c < C; ++c) {
        if ((rc = pipe.wait_stage(c))) return pipe.finish(rc);
        const uint8_t *pk = dpacked.as<uint8_t>() + (size_t)c * per / 2;
        uint32_t *po = dout.as<uint32_t>() + (size_t)c * per;
        mock_launch(L->compute, [=] { for (size_t i = 0; i < per; ++i) po[i] = 7u * ((pk[i >> 1] >> ((i & 1) * 4)) & 15u) + (uint32_t)i; });
        hipEvent_t *ev = pipe.join_events(c);
        if (hipEventRecord(ev[0], L->compute) != hipSuccess) return pipe.finish(GBX_ERR_HIP);
        pipe.fetch(c, out.data() + (size_t)c * per, po, per * 4);
        if ((rc = pipe.chunk_launched(c, 1))) return pipe.finish(rc);
    }
    rc = pipe.finish();
    if (rc == GBX_OK)
        for (size_t i = 0; i < per * C; i += 1 + per / 1021) {
            CHECK(out[i] == 7u * base[i] + (uint32_t)(i % per), "three-chunk result differs at %zu", i);
            if (g_fail) break;
        }
    return rc;
}

// shaped like gbx_abea_align_host: a 4-byte field gathered out of 24-byte records on the way up, a packed result
// scattered to per-unit places on the way down
static int call_field_and_scatter(size_t n_rec, uint64_t seed)
{
    struct Rec { uint64_t start; float length, mean, stdv; };
    std::vector<Rec> recs(n_rec);
    std::mt19937_64 r(seed);
    for (size_t i = 0; i < n_rec; ++i) recs[i] = Rec{r(), 1.f, (float)(r() % 100000) * 0.25f, 2.f};
    // units of 1..400 records; results of a unit go to its own place (stride 512 floats)
    std::vector<size_t> first, cnt;
    for (size_t at = 0; at < n_rec;) { const size_t k = std::min<size_t>(1 + r() % 400, n_rec - at); first.push_back(at); cnt.push_back(k); at += k; }
    std::vector<float> out(first.size() * 512, -1.f);
    std::vector<HostPipe::Seg> segs;
    for (size_t u = 0; u < first.size(); ++u) segs.push_back(HostPipe::Seg{(char *)(out.data() + u * 512), cnt[u] * 4});
    HostLane lane;
    int rc = lane.acquire();
    if (rc) return rc;
    Lane *L = lane.l;
    DevBuf dm(L), dr(L);
    if ((rc = dm.alloc(n_rec * 4)) || (rc = dr.alloc(n_rec * 4))) return rc;
    HostPipe pipe(L, n_rec * 24, false);
    if ((rc = pipe.prepare(1))) return rc;
    pipe.stage_field4(0, dm.p, &recs[0].mean, n_rec, (int)sizeof(Rec));
    pipe.start();
    if ((rc = pipe.wait_stage(0))) return pipe.finish(rc);
    const float *pm = dm.as<float>();
    float *pr = dr.as<float>();
    mock_launch(L->compute, [=] { for (size_t i = 0; i < n_rec; ++i) pr[i] = pm[i] * 2.f + 1.f; });
    if (pipe.staged) pipe.fetch_scatter(0, dr.p, n_rec * 4, &segs);
    else for (size_t u = 0; u < first.size(); ++u) pipe.fetch(0, out.data() + u * 512, pr + first[u], cnt[u] * 4);
    if ((rc = pipe.chunk_launched(0))) return pipe.finish(rc);
    rc = pipe.finish();
    if (rc == GBX_OK)
        for (size_t u = 0; u < first.size(); ++u) {
            bool ok = true;
            for (size_t k = 0; k < cnt[u]; ++k) ok = ok && out[u * 512 + k] == recs[first[u] + k].mean * 2.f + 1.f;
            ok = ok && (cnt[u] == 512 || out[u * 512 + cnt[u]] == -1.f);
            CHECK(ok, "scattered result of unit %zu differs", u);
            if (g_fail) break;
        }
    return rc;
}

// an error between start() and chunk_launched(): the call must come back (no downloader left waiting)
static void call_abandoned(size_t n, uint64_t seed)
{
    std::vector<uint8_t> a = random_bytes(n, seed);
    HostLane lane;
    if (lane.acquire()) return;
    Lane *L = lane.l;
    DevBuf da(L);
    if (da.alloc(n)) return;
    HostPipe pipe(L, n, false);
    if (pipe.prepare(1)) return;
    pipe.stage(0, da.p, a.data(), n);
    pipe.start();
    (void)pipe.wait_stage(0);
    // ... a launch fails here: the function returns without chunk_launched(); ~HostPipe cancels and joins
}

static int g_racy;
int main(int argc, char **argv)
{
    if (argc > 1 && !strcmp(argv[1], "--selftest-race")) {     // the harness must see a race when there is one
        std::thread a([] { for (int i = 0; i < 100000; ++i) g_racy += i; }), b([] { for (int i = 0; i < 100000; ++i) g_racy -= i; });
        a.join(); b.join();
        printf("selftest %d\n", g_racy);
        return 0;
    }
    const int threads = argc > 1 ? atoi(argv[1]) : 4, rounds = argc > 2 ? atoi(argv[2]) : 3;
    setenv("GBX_HELPER_CACHE", "5", 1);                       // few idle helper threads are kept: the others end after their task (Helper::join)
    // (0) the chunk cuts of gbx_bsw_extend_host: 0 .. n, every inner cut a multiple of 64, ascending; the default rule (at most
    //     three equal chunks, none below 512 Ki pairs) and the two overrides
    {
        auto well_formed = [](const std::vector<int64_t> &c, int64_t n) {
            if (c.size() < 2 || c.front() != 0 || c.back() != n) return false;
            for (size_t k = 1; k < c.size(); ++k) if (c[k] <= c[k - 1] || (k + 1 < c.size() && c[k] % 64)) return false;
            return true;
        };
        unsetenv("GBX_BSW_HOST_CHUNK"); unsetenv("GBX_BSW_HOST_CUTS");
        for (int64_t n : {(int64_t)1, (int64_t)63, (int64_t)524288, (int64_t)524289, (int64_t)1048577, (int64_t)2000000, (int64_t)10000001}) {
            const std::vector<int64_t> c = bsw_host_cuts(n);
            CHECK(well_formed(c, n), "default cuts of %lld pairs are malformed", (long long)n);
            CHECK(c.size() - 1 <= 3, "more than three default chunks for %lld pairs", (long long)n);
            CHECK(c.size() == 2 || c[1] >= 524288, "a default chunk below 512 Ki pairs for %lld", (long long)n);
        }
        CHECK(bsw_host_cuts(2000000).size() == 4 && bsw_host_cuts(524288).size() == 2, "default chunk counts");
        setenv("GBX_BSW_HOST_CHUNK", "1000", 1);
        CHECK(well_formed(bsw_host_cuts(4000), 4000) && bsw_host_cuts(4000).size() == 5 && bsw_host_cuts(4000)[1] == 1024, "cuts by GBX_BSW_HOST_CHUNK");
        unsetenv("GBX_BSW_HOST_CHUNK");
        setenv("GBX_BSW_HOST_CUTS", "0.5,0.1,0.9,2.0", 1);     // out of order or out of range entries are dropped
        CHECK(well_formed(bsw_host_cuts(100000), 100000) && bsw_host_cuts(100000).size() == 4, "cuts by GBX_BSW_HOST_CUTS");
        unsetenv("GBX_BSW_HOST_CUTS");
    }
    setenv("GBX_HOST_STAGE_MIN", "65536", 1);                 // calls of 64 KB and more are staged
    setenv("GBX_HOST_DOWN_PIECE", "262144", 1);               // downloads span several half-slab pieces
    // (a) every call shape from several caller threads at once, each on its own lane; lanes and their device-block caches
    //     are reused round after round
    std::vector<std::thread> th;
    for (int t = 0; t < threads; ++t)
        th.emplace_back([=] {
            for (int r = 0; r < rounds; ++r) {
                const uint64_t seed = 1000u * (uint64_t)t + (uint64_t)r;
                CHECK(call_one_chunk((size_t)3 << 20, seed, true) == GBX_OK, "staged one-chunk call: %s", g_err);
                CHECK(call_one_chunk(5000 + 977 * (size_t)t, seed + 7, true) == GBX_OK, "small (unstaged / packed) call: %s", g_err);
                CHECK(call_three_chunks(((size_t)1 << 20) + 64 * (size_t)t, seed + 13) == GBX_OK, "three-chunk call: %s", g_err);
                CHECK(call_three_chunks(((size_t)1 << 20) + 64 * (size_t)t, seed + 14, true) == GBX_OK, "three-chunk call staged late: %s", g_err);
                CHECK(call_three_chunks(((size_t)1 << 20) + 64 * (size_t)t, seed + 15, true, true) == GBX_ERR_ARG, "three-chunk call abandoned before seal()");
                CHECK(call_field_and_scatter(150000 + 1000 * (size_t)t, seed + 17) == GBX_OK, "field / scatter call: %s", g_err);
                CHECK(call_field_and_scatter(300, seed + 19) == GBX_OK, "small field call: %s", g_err);
                call_abandoned((size_t)1 << 20, seed + 23);
            }
        });
    for (auto &x : th) x.join();
    // (b) a transfer that fails in the middle of a staged call: an error comes back, nothing hangs, the next call works
    for (int k : {0, 1, 3, 6}) {
        mock_fail_after(k);
        const int rc = call_one_chunk((size_t)24 << 20, 99 + (uint64_t)k, false);
        mock_fail_after(-1);
        CHECK(rc != GBX_OK, "a failing hipMemcpyAsync (the %d-th) must fail the call", k);
        CHECK(call_one_chunk((size_t)1 << 20, 199 + (uint64_t)k, true) == GBX_OK, "call after a failed one: %s", g_err);
    }
    // (c) downloads in full-size pieces (8 MB halves of the pinned slab): the downloader's copy helpers (CopyPool) at work,
    //     from two callers at once
    unsetenv("GBX_HOST_DOWN_PIECE");
    {
        std::thread a([] { CHECK(call_one_chunk((size_t)20 << 20, 4001, true) == GBX_OK, "large one-chunk call: %s", g_err); });
        std::thread b([] { CHECK(call_field_and_scatter(3000000, 4002) == GBX_OK, "large field / scatter call: %s", g_err); });
        a.join(); b.join();
    }
    // (d) the multi-device shape of an entry (host_multi.h): a job cut by cost into three shards, every shard a staged call on
    //     a lane of its own from a thread of its own, two such jobs in flight at once; then a job one of whose shards fails:
    //     the lowest failing shard's status and text come back on the calling thread, the other shards finish
    {
        auto multi_job = [](size_t units, uint64_t seed, int fail_shard) -> int {
            std::vector<double> cost(units);
            std::mt19937_64 r(seed);
            for (auto &c : cost) c = (double)(r() % 1000);
            const std::vector<int64_t> cuts = split_by_cost((int64_t)units, 3, [&](int64_t i) { return cost[(size_t)i]; });
            CHECK(cuts[0] == 0 && cuts[3] == (int64_t)units && cuts[1] <= cuts[2], "cuts not monotone");
            int map[MAX_HOST_DEVICES];
            const int n = host_device_set(map);
            return run_on_devices(n, map, "multi_job", [&](int k) -> int {
                if (k == fail_shard) { set_error("shard %d was told to fail", k); return GBX_ERR_ARG; }
                const size_t m = (size_t)(cuts[(size_t)k + 1] - cuts[(size_t)k]);
                return m ? call_one_chunk(m * 64, seed + (uint64_t)k, true) : GBX_OK;
            });
        };
        std::thread a([&] { CHECK(multi_job(40000, 5001, -1) == GBX_OK, "multi-device job: %s", g_err); });
        std::thread b([&] { CHECK(multi_job(30000, 5002, -1) == GBX_OK, "multi-device job: %s", g_err); });
        a.join(); b.join();
        const int rc = multi_job(20000, 5003, 1);
        CHECK(rc == GBX_ERR_ARG && strstr(g_err, "shard 1 was told to fail") && strstr(g_err, "[shard 1 of 3"), "failing shard: rc %d, text '%s'", rc, g_err);
        CHECK(call_one_chunk((size_t)1 << 20, 5004, true) == GBX_OK, "call after a failed multi-device job: %s", g_err);
    }
    // (e) the call combiner (host_combine.h): many caller threads submit small requests round after round; a leader lays a
    //     batch's inputs end to end in its slot's scratch arrays (helper threads copy), runs ONE staged call for all of them and
    //     hands every caller its slice; two leaders in flight, requests of two "scoring" classes that must not mix, one request
    //     in twenty a bad one whose combined call fails and is redone request by request: every caller gets the result and the
    //     status of its own call
    {
        struct Req : CombineReq { int cls; std::vector<uint8_t> in, out; bool bad; };
        struct Slot { Scratch<uint8_t> in, out; };
        static Combiner comb;
        static Slot slots[Combiner::MAX_LEADERS];
        static std::atomic<int> mixed{0}, combined_calls{0};
        auto alone = [](Req *r) {
            if (r->bad) { r->rc = GBX_ERR_ARG; r->err = "request was told to fail"; return; }
            for (size_t i = 0; i < r->in.size(); ++i) r->out[i] = (uint8_t)(r->in[i] * 3 + r->cls);
            r->rc = GBX_OK;
        };
        auto run = [&](const std::vector<CombineReq *> &batch, int slot) {
            if (batch.size() == 1) { alone((Req *)batch[0]); return; }
            ++combined_calls;
            std::vector<size_t> off(batch.size() + 1, 0);
            bool any_bad = false;
            for (size_t k = 0; k < batch.size(); ++k) {
                const Req *r = (const Req *)batch[k];
                off[k + 1] = off[k] + r->in.size();
                any_bad = any_bad || r->bad;
                if (r->cls != ((const Req *)batch[0])->cls) ++mixed;
            }
            uint8_t *in = slots[slot].in.get(off.back() + 1), *out = slots[slot].out.get(off.back() + 1);
            combine_parallel((int64_t)batch.size(), 4, [&](int64_t k) { const Req *r = (const Req *)batch[(size_t)k]; memcpy(in + off[(size_t)k], r->in.data(), r->in.size()); });
            if (any_bad) { for (CombineReq *q : batch) alone((Req *)q); return; }      // the combined call failed: one by one
            // the device call of the batch: a staged one-chunk upload / "kernel" / download through a lane (host_pipeline.h)
            {
                HostLane lane;
                int rc = lane.acquire();
                DevBuf din(lane.l), dout(lane.l);
                if (!rc) rc = din.alloc(off.back() + 1);
                if (!rc) rc = dout.alloc(off.back() + 1);
                if (!rc) {
                    HostPipe pipe(lane.l, off.back(), false);
                    rc = pipe.prepare(1);
                    if (!rc) {
                        pipe.stage(0, din.p, in, off.back());
                        pipe.start();
                        rc = pipe.wait_stage(0);
                        const int cls = ((const Req *)batch[0])->cls;
                        const size_t total = off.back();
                        uint8_t *a = din.as<uint8_t>(), *b = dout.as<uint8_t>();
                        if (!rc) mock_launch(lane.l->compute, [=] { for (size_t i = 0; i < total; ++i) b[i] = (uint8_t)(a[i] * 3 + cls); });
                        if (!rc) { pipe.fetch(0, out, dout.p, total); rc = pipe.chunk_launched(0); }
                        rc = pipe.finish(rc);
                    }
                }
                if (rc) { for (CombineReq *q : batch) { q->rc = rc; q->err = g_err; } return; }
            }
            for (size_t k = 0; k < batch.size(); ++k) {
                Req *r = (Req *)batch[k];
                memcpy(r->out.data(), out + off[k], r->in.size());
                r->rc = GBX_OK;
            }
        };
        setenv("GBX_COMBINE_GATHER_US", "300", 1);
        const int callers = threads * 3;
        std::vector<std::thread> ct;
        for (int t = 0; t < callers; ++t)
            ct.emplace_back([&, t] {
                std::mt19937_64 rng(7000u + (uint64_t)t);
                for (int it = 0; it < 40 * rounds; ++it) {
                    Req r;
                    r.cls = (t + it) & 1;
                    r.bad = rng() % 20 == 0;
                    r.in = random_bytes(200 + (size_t)(rng() % 30000), rng());
                    r.out.assign(r.in.size(), 0);
                    r.units = (int64_t)r.in.size();
                    const int rc = comb.submit(&r, (int64_t)1 << 20, 2,
                        [](const CombineReq *a, const CombineReq *b) { return ((const Req *)a)->cls == ((const Req *)b)->cls; }, run);
                    if (r.bad) { CHECK(rc == GBX_ERR_ARG && strstr(g_err, "told to fail"), "a bad request's status: %d '%s'", rc, g_err); continue; }
                    CHECK(rc == GBX_OK, "combined request: %s", g_err);
                    bool ok = true;
                    for (size_t i = 0; i < r.in.size() && ok; ++i) ok = r.out[i] == (uint8_t)(r.in[i] * 3 + r.cls);
                    CHECK(ok, "a combined request came back with another request's results");
                }
            });
        for (auto &x : ct) x.join();
        CHECK(mixed.load() == 0, "requests of different classes shared a call");
        CHECK(comb.n_calls.load() == (uint64_t)callers * 40u * (uint64_t)rounds, "combiner call count");
        CHECK(combined_calls.load() > 0 && comb.largest.load() >= 2, "no call was ever combined (largest %llu)", (unsigned long long)comb.largest.load());
        printf("pipe_tsan: combiner %llu calls in %llu device calls, %llu shared, largest %llu\n", (unsigned long long)comb.n_calls.load(),
               (unsigned long long)comb.n_batches.load(), (unsigned long long)comb.n_shared.load(), (unsigned long long)comb.largest.load());
    }
    if (g_fail.load()) { fprintf(stderr, "pipe_tsan: %d check(s) failed\n", g_fail.load()); return 1; }
    printf("pipe_tsan: ok (%d caller threads x %d rounds)\n", threads, rounds);
    return 0;
}
