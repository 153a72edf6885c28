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

constexpr int NBUCKET = 32;

template <int CTRL, int ROWMASK = 0xf>
__device__ inline int dppi(int old, int x)
{
    return __builtin_amdgcn_update_dpp(old, x, CTRL, ROWMASK, 0xf, false);
}

// inclusive scans over the 64 lanes of the wave.  The row_shr steps use mov_dpp with bound_ctrl:1
// (lanes without a source read 0), which the compiler folds into the consumer (v_max_u32_dpp /
// v_add_u32_dpp): callers bias their values so that 0 is the identity.
template <int CTRL>
__device__ inline unsigned dppz(unsigned x) { return (unsigned)__builtin_amdgcn_mov_dpp((int)x, CTRL, 0xf, 0xf, true); }

__device__ inline unsigned wave_scan_umax(unsigned x)          // identity 0
{
    x = max(x, dppz<0x111>(x));
    x = max(x, dppz<0x112>(x));
    x = max(x, dppz<0x114>(x));
    x = max(x, dppz<0x118>(x));
    x = max(x, (unsigned)dppi<0x142, 0xa>(0, (int)x));
    x = max(x, (unsigned)dppi<0x143, 0xc>(0, (int)x));
    return x;
}
__device__ inline int wave_scan_add(int x)                     // identity 0
{
    x += (int)dppz<0x111>((unsigned)x);
    x += (int)dppz<0x112>((unsigned)x);
    x += (int)dppz<0x114>((unsigned)x);
    x += (int)dppz<0x118>((unsigned)x);
    x += dppi<0x142, 0xa>(0, x);
    x += dppi<0x143, 0xc>(0, x);
    return x;
}
constexpr unsigned UBIAS = 1u << 30;                           // scores and skip counters are far inside +-2^30

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

constexpr int RING = 256;                 // most recent anchors kept in LDS (covers the usual look-back)

// Loaded values that are produced on a rare path and consumed after the paths merge make the compiler put
// `s_waitcnt vmcnt(0)` at the merge point - which, on the common path, waits for this wavefront's
// outstanding global *stores* (gfx9 counts them in vmcnt), a microsecond per anchor.  settle() consumes the
// value inside the rare path, so the wait stays there.
__device__ inline int settle(int v) { asm volatile("" : "+v"(v)); return v; }
__device__ inline uint64_t settle(uint64_t v)
{
    unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
    asm volatile("" : "+v"(lo), "+v"(hi));
    return ((uint64_t)hi << 32) | lo;
}

__device__ inline uint64_t readlane64(uint64_t v, int k)
{
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, k);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), k);
    return ((uint64_t)hi << 32) | lo;
}

__global__ void __launch_bounds__(64) chain_kernel(int n_calls, const int64_t *__restrict__ off,
                                                   const uint64_t *__restrict__ ax, const uint64_t *__restrict__ ay,
                                                   const gbx_chain_call *__restrict__ hdr,
                                                   int32_t *score, int32_t *parent, int32_t *target, int32_t *peak,
                                                   ChainWork W)
{
    // ring of the last RING anchors (slot = index & (RING-1)): anchor words and the DP state of the reference's
    // scores / parents / targets / peak_scores vectors.  The same values also go to global memory (the outputs),
    // which serves the rare look-backs deeper than the ring.
    __shared__ uint64_t rx[RING], ry[RING];
    __shared__ int rf[RING], rp[RING], rt[RING], rk[RING];
    __shared__ int mark[64];
    const int lane = threadIdx.x;
    const int max_iter = GBX_CHAIN_MAX_ITER, max_skip = GBX_CHAIN_MAX_SKIP;

    // one call per block when the grid allows it (chain_launch), else a stride over the longest-first list;
    // everything derived from `slot` stays wave-uniform
    for (int slot = blockIdx.x; slot < n_calls; slot += gridDim.x) {
        const int call = W.order[slot];
        const int64_t o = off[call];
        const int n = (int)(off[call + 1] - o);
        const uint64_t *x = ax + o, *y = ay + o;
        int32_t *f = score + o, *p = parent + o, *t = target + o, *pk = peak + o;
        const gbx_chain_call h = hdr[call];
        const int max_dist_x = h.max_dist_x, max_dist_y = h.max_dist_y, bw = h.bw, n_segs = h.n_segs;
        const double avg_qspan = (double)h.avg_qspan;
        const uint64_t mdx = (uint64_t)(int64_t)max_dist_x;

        for (int i = lane; i < n; i += 64) t[i] = 0;          // vectors are zero-filled, host_kernel.cpp:44-47
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");

        int st = 0;
        int sb = 0;                                           // block of 64 x-values cached for the st scan
        uint64_t xs = n ? x[min(lane, n - 1)] : 0;
        unsigned long long visited = 0;
        for (int ib = 0; ib < n; ib += 64) {
            // this block's anchors, one per lane
            const uint64_t xa = settle(x[min(ib + lane, n - 1)]), ya = settle(y[min(ib + lane, n - 1)]);
            const int kmax = min(64, n - ib);
            for (int k = 0; k < kmax; ++k) {
                const int i = ib + k;
                const uint64_t ri = readlane64(xa, k), yi = readlane64(ya, k);
                const int qi = (int)yi, q_span = (int)(yi >> 32 & 0xff);
                const int sidi = (int)(yi >> 48 & 0xff);
                // advance st (:56): first st with ri <= x[st] + max_dist_x, scanning the cached block
                for (;;) {
                    if (st >= i) break;
                    if (st >= sb + 64 || st < sb) { sb = st; xs = settle(x[min(sb + lane, n - 1)]); }
                    const int idx = sb + lane;
                    const bool far = idx >= st && idx < i && ri > xs + mdx;
                    const bool stop = idx >= st && !far;                  // first lane at/after st that is not far
                    const unsigned long long m = __ballot(stop);
                    if (m) { st = sb + __builtin_ctzll(m); break; }
                    st = sb + 64;                                         // the whole rest of the block is far
                }
                if (st > i) st = i;
                if (i - st > max_iter) st = i - max_iter;                 // :57

                int max_f = q_span, max_j = -1, n_skip = 0;
                // One 64-wide chunk of the look-back, lane 0 = anchor jhi.  Returns true when the max_skip break fired.
                // The anchor words and DP state of the chunk arrive as arguments, so that the ring path below is
                // made of LDS reads only: a flat / global load here would have to wait (vmcnt) for the global
                // stores of the previous anchors, which is most of a chunk's latency.
                auto chunk = [&](int jhi, bool valid, uint64_t xj, uint64_t yj, int fj, int pj, int tj) -> bool {
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
                    // (int)((double)gap_cost * gap_scale + .499) with gap_scale = 1.0f is gap_cost itself: gap_cost >= 0
                    // for every lane that is not skipped (dd >= 0), and skipped lanes never use sc
                    sc -= gap_cost;
                    sc += fj;
                    // ---- phase 2: was this j already marked as a parent during this i? (:84)
                    mark[lane] = 0;
                    const int tl = jhi - pj;                       // lane that holds anchor pj, if inside this chunk
                    if (!skip && pj >= 0 && tl < 64) mark[tl] = 1; // tl > lane always: parents precede their children
                    const bool hit = (tj == i) || mark[lane] != 0;
                    // ---- phase 3: ordered max_f / n_skip / break (:81-88)
                    const unsigned cand = skip ? 0u : (unsigned)sc + UBIAS;      // biased: 0 = "no candidate"
                    bool improving, bump;
                    int nl;                                                       // n_skip after this lane
                    if (__ballot(cand > (unsigned)max_f + UBIAS) == 0) {
                        // no lane beats max_f (the usual case beyond the first chunk): nobody improves, n_skip only
                        // counts the marked lanes - a prefix popcount instead of three wave scans
                        improving = false;
                        bump = !skip && hit;
                        const unsigned long long bm = __ballot(bump);
                        const int below = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bm, 0u));
                        nl = n_skip + below + (bump ? 1 : 0);
                    } else {
                        const unsigned pm = wave_scan_umax(cand);                 // inclusive prefix max of candidates
                        unsigned pmx = (unsigned)dppi<0x138>(0, (int)pm);         // exclusive (wave_shr:1)
                        pmx = lane == 0 ? 0u : pmx;
                        improving = !skip && cand > max((unsigned)max_f + UBIAS, pmx);
                        bump = !skip && !improving && hit;
                        const int d = improving ? -1 : (bump ? 1 : 0);
                        const int S = n_skip + wave_scan_add(d);
                        // n_skip after this lane = walk reflected at 0: S - min(0, prefix-min S); the min via a biased max of -S
                        const unsigned mx = wave_scan_umax((unsigned)(-S) + UBIAS);
                        const int mn = -(int)(mx - UBIAS);
                        nl = S - min(0, mn);
                    }
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
                    if (!skip && pj >= 0 && lane < bl) {
                        t[pj] = i;
                        if (i - pj <= RING) rt[pj & (RING - 1)] = i;
                    }
                    n_skip = __builtin_amdgcn_readlane(nl, 63);
                    return bl < 64;
                };
                int jhi = i - 1;
                bool broke = false;
                // chunks that lie inside the ring: LDS only
                for (; jhi >= st && i - (jhi - 63) <= RING; jhi -= 64) {
                    const int j = jhi - lane;
                    const bool valid = j >= st;
                    const int rs = (valid ? j : st) & (RING - 1);
                    if ((broke = chunk(jhi, valid, rx[rs], ry[rs], rf[rs], rp[rs], rt[rs]))) break;
                }
                // deeper chunks: the global arrays (this wavefront's own earlier stores are visible in order)
                if (!broke)
                    for (; jhi >= st; jhi -= 64) {
                        const int j = jhi - lane;
                        const bool valid = j >= st;
                        const int jj = valid ? j : st;
                        if (chunk(jhi, valid, settle(x[jj]), settle(y[jj]), settle(f[jj]), settle(p[jj]), settle(t[jj]))) break;
                    }
                // :91-92, to the outputs and to the ring
                int pkj = 0;
                if (max_j >= 0) {
                    if (i - max_j <= RING) pkj = rk[max_j & (RING - 1)];
                    else pkj = settle(pk[max_j]);
                }
                const int pki = (max_j >= 0 && pkj > max_f) ? pkj : max_f;
                if (lane == 0) {
                    f[i] = max_f; p[i] = max_j; pk[i] = pki;
                    const int rs = i & (RING - 1);
                    rx[rs] = ri; ry[rs] = yi; rf[rs] = max_f; rp[rs] = max_j; rt[rs] = 0; rk[rs] = pki;
                }
            }
        }
        if (lane == 0) atomicAdd(W.evaluated, visited);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
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
    // One call per block, in the longest-first order of the list: the hardware dispatcher hands the next call to
    // whichever slot frees up, i.e. a dynamic LPT schedule (a persistent grid with a static stride over the list
    // was 102 ms on the 'large' job, this is 97 ms; calls in flight per CU: 4: 149 ms, 8: 117, 16: 103, 19 =
    // what the 8.4 KB LDS ring admits).  GBX_CHAIN_WAVES_PER_CU caps the grid at that many blocks per CU instead.
    const char *wenv = getenv("GBX_CHAIN_WAVES_PER_CU");
    const int64_t cap = wenv && atoi(wenv) > 0 ? (int64_t)cus * atoi(wenv) : (int64_t)1 << 20;
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
