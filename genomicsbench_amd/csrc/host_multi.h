// host_multi.h — the *_host entries over several devices of one node (every exported entry carries the number
// of GPUs; §8e: the units of all six kernels are independent, so a call is cut into contiguous cost-balanced ranges
// with no exchange step).  This is the reference's own shape moved from OpenMP threads to devices: one aligner object /
// engine per thread working on a disjoint slice of the batch (bsw/main_banded.cpp:253-258,279-291,
// chain/src/host_kernel.cpp:98-107, phmm/PairHMMUnitTest.cpp:224-247, poa/msa_spoa_omp.cpp:184-196,230-260) becomes one
// host lane per device (host_pipeline.h: streams, pinned slabs, upload / download threads) fed straight from the
// caller's memory - the inputs start on the host, so every device gets its shard by its own H2D and writes its results in
// place; nothing travels between devices.  (bench.py's scatter / gather over RCCL is the other route: inputs that
// already live in one GPU's HBM.)
//
// gbx_host_set_devices(n) / GBX_GPUS choose how many devices a call is spread over; GBX_DEVICE_MAP="0,0,1" (a test aid)
// maps the logical devices 0..n-1 onto physical ones, so that the multi-device path can be exercised on a one-GPU box.
// A call too small to be worth cutting (shard_parts) runs whole on one device, the devices taking such calls in turn:
// a reference driver whose OpenMP threads each hand over a small slice (bsw: 512 pairs) still uses every GPU.
#pragma once
#include <atomic>
#include <exception>
#include <new>
#include <string>

namespace gbx {

constexpr int MAX_HOST_DEVICES = 16;

// defined in gbx_core.hip
int host_device_set(int *map);           // n >= 1 and map[0..n) = physical device ids, or a negative status (error text set)
int host_next_small_call_device(int n);  // 0..n-1 in turn
bool host_multi_wanted();                // more than one device asked for, or a device map set (no HIP call: the one-device
                                         // path of an entry must stay exactly what it was, errors and their order included)

struct DeviceGuard {                     // selects a device for the calling thread, restores the previous one on scope exit
    int prev = -1;
    int set(int dev)
    {
        int cur = 0;
        GBX_HIP(hipGetDevice(&cur));
        if (cur == dev) return GBX_OK;
        GBX_HIP(hipSetDevice(dev));
        prev = cur;
        return GBX_OK;
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

// how many shards a call of `units` units is cut into on n devices: every shard gets at least min_units of them
// (GBX_SHARD_MIN_UNITS overrides the kernel's figure: the tests set 1 to cut small jobs)
static int shard_parts(int n_dev, int64_t units, int64_t min_units)
{
    if (const char *e = getenv("GBX_SHARD_MIN_UNITS")) { const long long v = atoll(e); if (v >= 1) min_units = v; }
    if (min_units < 1) min_units = 1;
    int64_t parts = units / min_units;
    if (parts > n_dev) parts = n_dev;
    return parts < 1 ? 1 : (int)parts;
}

// Contiguous ranges of near-equal total cost: cut k lies at the first unit index i where the cost of units [0, i) reaches
// k / parts of the total (genomicsbench_amd/shard.py:split_by_cost is the same rule: the C++ drivers and bench.py cut a
// job identically).  cost(i) >= 0, as double (all integer costs here stay below 2^53: their sums are exact).
template <class CostFn> static std::vector<int64_t> split_by_cost(int64_t n, int parts, CostFn cost)
{
    std::vector<int64_t> cuts((size_t)parts + 1, n);
    cuts[0] = 0;
    if (parts <= 1 || n == 0) { if (n == 0) for (auto &c : cuts) c = 0; return cuts; }
    // block sums by a few threads (the pair lists run to ten million entries), then a walk over the blocks
    const int64_t BL = 65536, nb = (n + BL - 1) / BL;
    std::vector<double> bsum((size_t)nb, 0.0);
    auto block_sum = [&](int64_t b) {
        double s = 0.0;
        const int64_t hi = (b + 1) * BL < n ? (b + 1) * BL : n;
        for (int64_t i = b * BL; i < hi; ++i) s += cost(i);
        bsum[(size_t)b] = s;
    };
    {
        const int T = nb >= 32 ? host_workers() : 1;
        std::vector<Helper> th;
        for (int t = 1; t < T; ++t) th.emplace_back([&, t] { for (int64_t b = t; b < nb; b += T) block_sum(b); }, true);
        for (int64_t b = 0; b < nb; b += T) block_sum(b);
        for (auto &x : th) x.join();
    }
    double total = 0.0;
    for (double s : bsum) total += s;
    double before = 0.0;           // cost of the blocks before block b
    int64_t b = 0;
    for (int k = 1; k < parts; ++k) {
        const double want = total * k / parts;
        while (b < nb && before + bsum[(size_t)b] < want) before += bsum[(size_t)b++];
        int64_t i = b * BL;
        double cum = before;
        const int64_t hi = (b + 1) * BL < n ? (b + 1) * BL : n;
        while (i < hi && cum < want) cum += cost(i++);
        cuts[(size_t)k] = b < nb ? i : n;
        if (cuts[(size_t)k] < cuts[(size_t)k - 1]) cuts[(size_t)k] = cuts[(size_t)k - 1];
    }
    return cuts;
}

// fn(k) for k in [0, parts) on parts threads, thread k with device map[k] selected; returns the status of the lowest
// failing shard and leaves its error text (with the shard named, and how many others failed) as the calling thread's.
// A C++ exception inside a shard (bad_alloc from its buffers) becomes GBX_ERR_NOMEM for that shard; a thread that cannot
// be started makes its shard run on the calling thread after the others.
template <class F> static int run_on_devices(int parts, const int *map, const char *who, F fn)
{
    std::vector<int> rcs((size_t)parts, GBX_OK);
    std::vector<std::string> errs((size_t)parts);
    auto body = [&](int k) {
        int rc;
        try {
            const hipError_t e = hipSetDevice(map[k]);
            rc = e == hipSuccess ? fn(k) : hip_fail(e, "hipSetDevice");
        } catch (const std::bad_alloc &) {
            set_error("%s: out of host memory", who);
            rc = GBX_ERR_NOMEM;
        } catch (const std::exception &ex) {
            set_error("%s: %s", who, ex.what());
            rc = GBX_ERR_NOMEM;
        }
        rcs[(size_t)k] = rc;
        if (rc) errs[(size_t)k] = gbx_last_error();
    };
    int cur = 0;
    GBX_HIP(hipGetDevice(&cur));
    std::vector<std::thread> th;
    std::vector<int> inline_shards;
    for (int k = 1; k < parts; ++k) {
        try { th.emplace_back(body, k); }
        catch (const std::exception &) { inline_shards.push_back(k); }
    }
    body(0);
    for (int k : inline_shards) body(k);
    for (auto &t : th) t.join();
    (void)hipSetDevice(cur);
    int first = -1, failed = 0;
    for (int k = 0; k < parts; ++k)
        if (rcs[(size_t)k]) { if (first < 0) first = k; ++failed; }
    if (first < 0) return GBX_OK;
    if (failed > 1) set_error("%s [shard %d of %d, device %d; %d other shard(s) failed too]", errs[(size_t)first].c_str(), first, parts, map[first], failed - 1);
    else set_error("%s [shard %d of %d, device %d]", errs[(size_t)first].c_str(), first, parts, map[first]);
    return rcs[(size_t)first];
}

}  // namespace gbx
