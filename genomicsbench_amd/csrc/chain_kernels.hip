// chain_kernels.hip — minimap2 anchor-chaining DP for gfx950 (MI355X).
//
// Semantics: chain_dp, R/benchmarks/chain/src/host_kernel.cpp:30-94, bit-exact on
// score / parent / target / peak.  host_chain_kernel (:96-108) is what
// gbx_chain_* replaces.
//
// The recurrence over anchors i is strictly sequential inside one call (f[i]
// needs every f[j] of its look-back window, and the max_skip early exit is
// loop-carried), so a call is one wavefront and thousands of calls run side
// by side.  Inside an anchor the look-back j = i-1 .. st is swept in
// descending 64-wide chunks, lane 0 = nearest predecessor:
//   phase 1  per-lane candidate score sc(j) and the reference's `continue` mask
//            (fp64 for (int)(dd*.01*avg_qspan), as the reference computes it);
//   phase 2  "targets[j]==i" (host_kernel.cpp:84) is true iff an earlier visited,
//            non-skipped j' has parents[j']==j: earlier chunks are already in
//            memory, the current chunk is resolved through a 64-entry LDS mark
//            (parents[j'] < j' always, so only earlier lanes can mark lane j);
//   phase 3  the ordered max_f / n_skip / break logic as wave scans: exclusive
//            prefix-max for "sc > max_f", and n_skip as a walk reflected at 0
//            (prefix sum + prefix min); first lane with n_skip > max_skip breaks;
//   phase 4  targets[parents[j]] = i for the lanes before the break (:89).
// Calls are handed out longest-first, round-robin over the resident wavefronts
// (the longest call bounds the kernel's makespan).
#include "gbx_internal.h"

namespace gbx {
namespace {

constexpr int NEGI = -(1 << 30);
constexpr int NBUCKET = 32;

template <int CTRL, int ROWMASK = 0xf>
__device__ inline int dppi(int old, int x)
{
    return __builtin_amdgcn_update_dpp(old, x, CTRL, ROWMASK, 0xf, false);
}

struct OpMax { static constexpr int id = NEGI; __device__ static int f(int a, int b) { return max(a, b); } };
struct OpMin { static constexpr int id = -NEGI; __device__ static int f(int a, int b) { return min(a, b); } };
struct OpAdd { static constexpr int id = 0; __device__ static int f(int a, int b) { return a + b; } };

// inclusive scan over the 64 lanes of the wave
template <class Op>
__device__ inline int wave_scan(int x)
{
    x = Op::f(x, dppi<0x111>(Op::id, x));
    x = Op::f(x, dppi<0x112>(Op::id, x));
    x = Op::f(x, dppi<0x114>(Op::id, x));
    x = Op::f(x, dppi<0x118>(Op::id, x));
    x = Op::f(x, dppi<0x142, 0xa>(Op::id, x));
    x = Op::f(x, dppi<0x143, 0xc>(Op::id, x));
    return x;
}

struct ChainWork {
    int32_t *counts;    // [NBUCKET] calls per size bucket
    int32_t *cursors;   // [NBUCKET]
    int32_t *next;      // work cursor
    int32_t *order;     // [n_calls] calls, longest bucket first
    unsigned long long *evaluated;   // predecessor pairs visited (the benchmark's "cell")
};

__device__ inline int bucket_of(int64_t n)
{
    // bucket 0 = longest calls
    const int lg = n > 0 ? 63 - __builtin_clzll((unsigned long long)n) : 0;
    return NBUCKET - 1 - min(lg, NBUCKET - 1);
}

__global__ void __launch_bounds__(256) chain_order_kernel(int64_t n_calls, const int64_t *off, ChainWork W, int pass)
{
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= n_calls) return;
    const int b = bucket_of(off[c + 1] - off[c]);
    if (pass == 0) { atomicAdd(&W.counts[b], 1); return; }
    int base = 0;
    for (int k = 0; k < b; ++k) base += W.counts[k];
    W.order[base + atomicAdd(&W.cursors[b], 1)] = (int)c;
}

__global__ void __launch_bounds__(64) chain_kernel(int n_calls, const int64_t *__restrict__ off,
                                                   const uint64_t *__restrict__ ax, const uint64_t *__restrict__ ay,
                                                   const gbx_chain_call *__restrict__ hdr,
                                                   int32_t *score, int32_t *parent, int32_t *target, int32_t *peak,
                                                   ChainWork W)
{
    __shared__ int mark[64];
    const int lane = threadIdx.x;
    const int max_iter = GBX_CHAIN_MAX_ITER, max_skip = GBX_CHAIN_MAX_SKIP;

    // static round-robin over the longest-first list (an LPT schedule); everything derived from `slot`
    // stays wave-uniform
    for (int slot = blockIdx.x; slot < n_calls; slot += gridDim.x) {
        const int call = W.order[slot];
        const int64_t o = off[call];
        const int n = (int)(off[call + 1] - o);
        const uint64_t *x = ax + o, *y = ay + o;
        int32_t *f = score + o, *p = parent + o, *t = target + o, *pk = peak + o;
        const gbx_chain_call h = hdr[call];
        const int max_dist_x = h.max_dist_x, max_dist_y = h.max_dist_y, bw = h.bw, n_segs = h.n_segs;
        const double avg_qspan = (double)h.avg_qspan;

        for (int i = lane; i < n; i += 64) t[i] = 0;          // vectors are zero-filled, host_kernel.cpp:44-47
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");

        int st = 0;
        unsigned long long visited = 0;
        for (int i = 0; i < n; ++i) {
            const uint64_t ri = x[i], yi = y[i];
            const int qi = (int)yi, q_span = (int)(yi >> 32 & 0xff);
            const int sidi = (int)(yi >> 48 & 0xff);
            // advance st (:56): first st with ri <= x[st] + max_dist_x
            while (st < i) {
                const int idx = st + lane;
                const bool far = idx < i && ri > x[idx] + (uint64_t)(int64_t)max_dist_x;
                const unsigned long long m = __ballot(far);
                const int adv = m == ~0ull ? 64 : __builtin_ctzll(~m);
                st += adv;
                if (adv < 64) break;
            }
            if (i - st > max_iter) st = i - max_iter;         // :57

            int max_f = q_span, max_j = -1, n_skip = 0;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            for (int jhi = i - 1; jhi >= st; jhi -= 64) {
                const int j = jhi - lane;
                const bool valid = j >= st;
                const int jj = valid ? j : st;
                const uint64_t xj = x[jj], yj = y[jj];
                const int fj = f[jj], pj = p[jj], tj = t[jj];
                // ---- phase 1: candidate score / `continue` mask (:59-80)
                const int64_t dr = (int64_t)(ri - xj);
                const int dq = qi - (int)yj;
                const int sidj = (int)(yj >> 48 & 0xff);
                const bool same = sidi == sidj;
                bool skip = !valid || (same && dr == 0) || dq <= 0;
                skip = skip || (same && dq > max_dist_y) || dq > max_dist_x;
                const int dd = (int)(dr > dq ? dr - dq : dq - dr);
                skip = skip || (same && dd > bw);
                skip = skip || (n_segs > 1 && same && dr > max_dist_y);
                const int min_d = dq < dr ? dq : (int)dr;
                int sc = min_d > q_span ? q_span : min_d;
                const int log_dd = dd ? 31 - __builtin_clz((unsigned)dd) : 0;
                const int c_lin = (int)((double)dd * .01 * avg_qspan);
                int gap_cost = 0;
                if (!same) {
                    if (dr == 0) ++sc;
                    else gap_cost = c_lin < log_dd ? c_lin : log_dd;
                } else {
                    gap_cost = c_lin + (log_dd >> 1);
                }
                sc -= (int)((double)gap_cost * 1.0 + .499);
                sc += fj;
                // ---- phase 2: was this j already marked as a parent during this i? (:84)
                mark[lane] = 0;
                const int tl = jhi - pj;                       // lane that holds anchor pj, if inside this chunk
                if (!skip && pj >= 0 && tl < 64) mark[tl] = 1; // tl > lane always: parents precede their children
                const bool hit = (tj == i) || mark[lane] != 0;
                // ---- phase 3: ordered max_f / n_skip / break (:81-88)
                const int cand = skip ? NEGI : sc;
                int pm = wave_scan<OpMax>(cand);               // inclusive prefix max of candidates
                int pmx = dppi<0x138>(NEGI, pm);               // exclusive (wave_shr:1)
                pmx = lane == 0 ? NEGI : pmx;
                const bool improving = !skip && sc > max(max_f, pmx);
                const bool bump = !skip && !improving && hit;
                const int d = improving ? -1 : (bump ? 1 : 0);
                const int S = n_skip + wave_scan<OpAdd>(d);
                const int mn = wave_scan<OpMin>(S);
                const int nl = S - min(0, mn);                 // n_skip after this lane (walk reflected at 0)
                const unsigned long long brk = __ballot(bump && nl > max_skip);
                const int bl = brk ? __builtin_ctzll(brk) : 64;   // first breaking lane
                const unsigned long long before = bl >= 64 ? ~0ull : ((1ull << bl) - 1);
                visited += __builtin_popcountll(__ballot(valid) & (bl >= 63 ? ~0ull : ((2ull << bl) - 1)));
                const unsigned long long imp = __ballot(improving) & before;
                if (imp) {
                    const int li = 63 - __builtin_clzll(imp);  // last improving lane before the break
                    max_j = jhi - li;
                    max_f = __builtin_amdgcn_readlane(sc, li);
                }
                // ---- phase 4: targets[parents[j]] = i for lanes visited before the break (:89)
                if (!skip && pj >= 0 && lane < bl) t[pj] = i;
                if (bl < 64) break;
                n_skip = __builtin_amdgcn_readlane(nl, 63);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            }
            if (lane == 0) {                                  // :91-92
                f[i] = max_f;
                p[i] = max_j;
                const int pkj = max_j >= 0 ? pk[max_j] : 0;
                pk[i] = (max_j >= 0 && pkj > max_f) ? pkj : max_f;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        }
        if (lane == 0) atomicAdd(W.evaluated, visited);
    }
}

}  // namespace

size_t chain_workspace_bytes(int64_t n_calls, int64_t n_anchors)
{
    // counters (3*NBUCKET ints) + order[n_calls] + optional target/peak planes
    return (size_t)(3 * NBUCKET + (n_calls > 0 ? n_calls : 0) + 2 * (n_anchors > 0 ? n_anchors : 0) + 16) * sizeof(int32_t);
}

int chain_read_evaluated(const void *d_work, int64_t *pairs, hipStream_t s)
{
    unsigned long long v = 0;
    GBX_HIP(hipMemcpyAsync(&v, (const char *)d_work + (2 * NBUCKET + 2) * sizeof(int32_t), sizeof(v), hipMemcpyDeviceToHost, s));
    GBX_HIP(hipStreamSynchronize(s));
    *pairs = (int64_t)v;
    return GBX_OK;
}

int chain_launch(int64_t n_calls, int64_t n_anchors, const int64_t *d_off,
                 const uint64_t *d_ax, const uint64_t *d_ay, const gbx_chain_call *d_hdr,
                 int32_t *d_score, int32_t *d_parent, int32_t *d_target, int32_t *d_peak,
                 void *d_work, size_t work_bytes, hipStream_t s)
{
    if (n_calls == 0) return GBX_OK;
    if (n_calls > 0x7fffffffLL - 1024) { set_error("chain: more than 2^31 calls"); return GBX_ERR_UNSUPPORTED; }
    if (work_bytes < chain_workspace_bytes(n_calls, n_anchors)) { set_error("chain: workspace too small"); return GBX_ERR_ARG; }
    int32_t *wi = (int32_t *)d_work;
    // [counts | cursors | next, evaluated(u64 at +2) | order[n_calls] | spare planes]
    ChainWork W = {wi, wi + NBUCKET, wi + 2 * NBUCKET, wi + 3 * NBUCKET, (unsigned long long *)(wi + 2 * NBUCKET + 2)};
    int32_t *spare = wi + 3 * NBUCKET + n_calls + 8;
    if (!d_target) d_target = spare;
    if (!d_peak) d_peak = spare + n_anchors;
    GBX_HIP(hipMemsetAsync(d_work, 0, 3 * NBUCKET * sizeof(int32_t), s));
    const int ob = (int)((n_calls + 255) / 256);
    {
        Stage st("chain_order", s);
        hipLaunchKernelGGL(chain_order_kernel, dim3(ob), dim3(256), 0, s, n_calls, d_off, W, 0);
        hipLaunchKernelGGL(chain_order_kernel, dim3(ob), dim3(256), 0, s, n_calls, d_off, W, 1);
    }
    int dev_id = 0, cus = 256;
    (void)hipGetDevice(&dev_id);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev_id);
    const int64_t cap = (int64_t)cus * 16;
    const int blocks = (int)(n_calls < cap ? n_calls : cap);
    {
        Stage st("chain_dp", s);
        hipLaunchKernelGGL(chain_kernel, dim3(blocks), dim3(64), 0, s, (int)n_calls, d_off, d_ax, d_ay, d_hdr,
                           d_score, d_parent, d_target, d_peak, W);
    }
    GBX_HIP(hipGetLastError());
    return GBX_OK;
}

}  // namespace gbx
