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
// Work unit = a *job*: a call, or a piece of one.  For sorted x the first predecessor st(i) is non-decreasing, so
// an anchor c with st(c) == c (it looks back at nobody, :56) is a point no later anchor looks across either: the
// anchors [c, next such point) only ever read, mark (targets[parents[j]], :89) and point at (parents) anchors of
// their own piece, and the piece is an independent chain_dp over a slice of the call whose parent / target values
// are offset by c.  Real minimap2 calls (both strands and many reference ids packed into x, repeat hits far from
// the true locus) are full of such points; chain_st_kernel finds them, chain_jobs_kernel turns them into jobs (at
// most one cut per 64 anchors, so the job list is bounded), and a piece whose anchors share one upper x word and
// one segment id takes the narrow 32-bit path even when its call as a whole could not.
// Jobs are handed out longest-first from a cursor (the longest job bounds the kernel's makespan).
#include "gbx_internal.h"
#include "chain_split.h"

namespace gbx {
namespace {

constexpr int NBUCKET = 32;
constexpr int CHAIN_MAX_DEVICES = 64;
typedef __attribute__((address_space(3))) int lds_int;

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
    int32_t *counts;    // [NBUCKET] jobs per size bucket
    int32_t *cursors;   // [NBUCKET]
    int32_t *next;      // [0] the DP kernel's job cursor, [4] ring choice, [5] number of jobs
    int32_t *order;     // [max_jobs] jobs, longest bucket first
    unsigned long long *evaluated;   // predecessor pairs visited (the benchmark's "cell")
    int32_t *st;        // [n_anchors] first predecessor of every anchor (chain_st_kernel)
    int32_t *unsorted;  // [n_calls] bit 0: the call's x are not sorted (one job, the DP kernel walks st itself); bit 1:
                        // preset by GBX_CHAIN_WIDE (every job of the call on the general 64-bit path)
    // per aligned block of 64 anchors of a call (index ((off + 64 w) >> 6) + call: unique and increasing)
    int8_t *blk_cut;    // position of the block's first anchor with st(i) == i, i > 0 (a cut), or -1
    unsigned long long *blk_diff;   // lanes whose upper x word or segment id differs from the preceding anchor's
    // jobs (chain_jobs_kernel)
    int64_t *job_start; // [max_jobs] first anchor (index into the concatenated arrays)
    int32_t *job_n;     // [max_jobs] anchors
    int32_t *job_call;  // [max_jobs]
    int32_t *job_flag;  // [max_jobs] bit 1: several segment ids or upper x words inside the job (general 64-bit path)
    int32_t max_jobs;
};

__device__ inline int bucket_of(int64_t n)
{
    // bucket 0 = longest calls
    const int lg = n > 0 ? 63 - __builtin_clzll((unsigned long long)n) : 0;
    return NBUCKET - 1 - min(lg, NBUCKET - 1);
}

__global__ void __launch_bounds__(256) chain_order_kernel(ChainWork W, int pass)
{
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= W.next[5]) return;
    const int b = bucket_of(W.job_n[c]);
    if (pass == 0) { atomicAdd(&W.counts[b], 1); return; }
    int base = 0;
    for (int k = 0; k < b; ++k) base += W.counts[k];
    W.order[base + atomicAdd(&W.cursors[b], 1)] = (int)c;
}

// -DGBX_CHAIN_STAMPS: s_memtime stamps around the parts of an anchor, accumulated by the block that runs the longest
// call (slot 0) and read back with gbx_debug_chain_stamps (scripts/dbg_chain_stamps.py).  A stamp is read through
// lgkmcnt, so it also drains the LDS reads issued before it: a part is charged its own LDS latency.
#ifndef GBX_CHAIN_QUIET
#define GBX_CHAIN_QUIET 1
#endif
#ifndef GBX_CHAIN_WINDOW
#define GBX_CHAIN_WINDOW 1
#endif
#ifdef GBX_CHAIN_STAMPS
__device__ unsigned long long g_chain_stamps[16];
#define STAMP(k) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); acc_[k] += now_ - last_; last_ = now_; } while (0)
#define STAMP_RESET() do { last_ = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define STAMP(k) do { } while (0)
#define STAMP_RESET() do { } while (0)
#endif

// Ring of the most recent anchors in LDS, slabs of 64 (anchor a lives in slot a mod RING_PHYS).  While the block of
// anchors [ib, ib+64) is being computed, the anchors a >= ib - RING_LIVE are addressed in the ring (complete
// blocks + the one being filled); older ones are read from / written to global memory.
// Ring size (GBX_CHAIN_RING_LIVE to tune): measured on 'large', whose longest call is the critical path and whose
// look-backs reach past 256 anchors in a tenth of its chunks: 256 anchors (10.5 KB, 15 calls per CU) 76.7-78.0 ms,
// 512 (18.7 KB, 8 per CU) 74.1-75.0, 768 (26.9 KB, 6 per CU) 72.2-73.3.  512 keeps eight calls per CU for jobs made of
// many short calls.
// The kernel exists for two ring sizes and a job takes one of them (chain_pick_kernel): the larger when the job is as
// long as its longest call anyway.
#ifndef GBX_CHAIN_RING_LIVE
#define GBX_CHAIN_RING_LIVE 512
#endif
#ifndef GBX_CHAIN_RING_LIVE_LONG
#define GBX_CHAIN_RING_LIVE_LONG 768
#endif

// Loaded values that are produced on a rare path and consumed after the paths merge make the compiler put
// `s_waitcnt vmcnt(0)` at the merge point - which, on the common path, waits for this wavefront's
// outstanding global *stores* (gfx9 counts them in vmcnt).  settle() consumes the value inside the rare path,
// so the wait stays there.
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

// Pre-pass, one thread per anchor, a block per call slice: the first predecessor of every anchor (host_kernel.cpp:56-57)
//     st(i) = max(first s in [0,i] with x[i] <= x[s] + max_dist_x,  i - max_iter)
// which is what the reference's running `st` equals at anchor i when the anchors are sorted by x (both terms are
// non-decreasing in i).  It depends on the anchor words only, so it is computed here at full-chip throughput instead
// of on the serial path of the call's wavefront (it was a ninth of an anchor's latency there).  A call whose x are
// not sorted gets its flag set and the DP kernel walks `st` for it as the reference does.
__global__ void __launch_bounds__(256) chain_st_kernel(int n_calls, const int64_t *__restrict__ off, const uint64_t *__restrict__ ax,
                                                       const uint64_t *__restrict__ ay,
                                                       const gbx_chain_call *__restrict__ hdr, int32_t *__restrict__ st_out,
                                                       int32_t *__restrict__ unsorted, int8_t *__restrict__ blk_cut,
                                                       unsigned long long *__restrict__ blk_diff)
{
    const int call = blockIdx.x;
    if (call >= n_calls) return;
    const int64_t o = off[call];
    const int n = (int)(off[call + 1] - o);
    const uint64_t *x = ax + o, *y = ay + o;
    const uint64_t mdx = (uint64_t)(int64_t)hdr[call].max_dist_x;
    bool bad = false;
    const int lane = threadIdx.x & 63;
    // a wavefront takes an aligned block of 64 anchors of the call at a time (i0 is wave-uniform)
    for (int i0 = blockIdx.y * 256 + (threadIdx.x & ~63); i0 < n; i0 += gridDim.y * 256) {
        const int i = i0 + lane;
        bool cut = false, diff = false;
        if (i < n) {
            const uint64_t ri = x[i];
            if (i > 0) {
                const uint64_t rp = x[i - 1];
                if (rp > ri) bad = true;
                // the DP kernel's narrow path needs one segment id (y bits 48-55) and one upper x word (strand + reference
                // id, host_data.h) in the whole job
                diff = ((ri ^ rp) >> 32) || ((y[i] ^ y[i - 1]) >> 48 & 0xff);
            }
            // gallop backwards from i, then bisect: lo = last known far index (or -1), hi = known not-far index
            int hi = i, lo = -1, step = 32;
            const int floor_ = i - GBX_CHAIN_MAX_ITER > 0 ? i - GBX_CHAIN_MAX_ITER : 0;
            while (hi > floor_) {
                int probe = hi - step;
                if (probe < floor_) probe = floor_;
                if (ri > x[probe] + mdx) { lo = probe; break; }
                hi = probe;
                step <<= 1;
            }
            if (lo < 0) lo = floor_ - 1;                      // everything down to the floor is near
            while (hi - lo > 1) {
                const int mid = (hi + lo) >> 1;
                if (ri > x[mid] + mdx) lo = mid; else hi = mid;
            }
            st_out[o + i] = hi;                               // first not-far index >= floor (hi == i when all are far)
            cut = hi == i && i > 0;                           // nobody to look back at: no later anchor looks across i
        }
        const unsigned long long cm = __ballot(cut), dm = __ballot(diff);
        if (lane == 0) {
            const int64_t blk = ((o + i0) >> 6) + call;
            blk_cut[blk] = cm ? (int8_t)__builtin_ctzll(cm) : (int8_t)-1;
            blk_diff[blk] = dm;
        }
    }
    if (__syncthreads_or(bad) && threadIdx.x == 0) atomicOr(&unsorted[call], 3);    // unsorted calls take the wide path too
}

// Jobs of a call: [0, c1), [c1, c2) ... at the cuts chain_st_kernel recorded (one per block of 64 anchors at most); an
// unsorted call is one job.  One block per call; the job slots of a call are contiguous (one atomic per call), the
// order kernel sorts the jobs by size afterwards.  job_flag is zero on entry.
// skip (nullable): calls that get no job at all - the host entry runs its longest calls as a launch of their own (capi_chain.hip).
__global__ void __launch_bounds__(256) chain_jobs_kernel(int n_calls, const int64_t *__restrict__ off, ChainWork W, int split_on,
                                                         const uint8_t *__restrict__ skip)
{
    const int call = blockIdx.x;
    if (call >= n_calls) return;
    if (skip && skip[call]) return;
    const int64_t o = off[call];
    const int n = (int)(off[call + 1] - o);
    const int nb = (n + 63) >> 6;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int cflags = W.unsorted[call];
    const bool split = split_on && !(cflags & 1);
    const int64_t blk0 = (o >> 6) + call;                    // ((o + 64 b) >> 6) + call == blk0 + b
    __shared__ int s_w[4];
    __shared__ int s_base;
    int cnt = 0;
    if (split)
        for (int b = tid; b < nb; b += 256) cnt += W.blk_cut[blk0 + b] >= 0;
    for (int d = 32; d; d >>= 1) cnt += __shfl_xor(cnt, d);
    if (lane == 0) s_w[wv] = cnt;
    __syncthreads();
    const int total = s_w[0] + s_w[1] + s_w[2] + s_w[3];      // cuts = jobs - 1
    if (tid == 0) {
        const int base = atomicAdd(&W.next[5], total + 1);
        s_base = base;
        W.job_start[base] = o;
        W.job_call[base] = call;
    }
    __syncthreads();
    const int base = s_base;
    int carry = 0;
    for (int b0 = 0; b0 < nb; b0 += 256) {
        const int b = b0 + tid;
        const int p = (split && b < nb) ? W.blk_cut[blk0 + b] : -1;
        const unsigned long long dm = b < nb ? W.blk_diff[blk0 + b] : 0ull;
        const unsigned long long m = __ballot(p >= 0);
        const int below = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
        __syncthreads();
        if (lane == 0) s_w[wv] = (int)__builtin_popcountll(m);
        __syncthreads();
        int before = carry + below;
        for (int k = 0; k < wv; ++k) before += s_w[k];
        // the job open at the start of this block is base + before; a cut at position p opens base + before + 1
        if (p >= 0) {
            W.job_start[base + before + 1] = o + 64ll * b + p;
            W.job_call[base + before + 1] = call;
            if (dm & ((1ull << p) - 1)) atomicOr(&W.job_flag[base + before], 2);
            if (p < 63 && (dm >> (p + 1))) atomicOr(&W.job_flag[base + before + 1], 2);
        } else if (dm) {
            atomicOr(&W.job_flag[base + before], 2);
        }
        carry += s_w[0] + s_w[1] + s_w[2] + s_w[3];
    }
    __threadfence_block();
    __syncthreads();
    for (int j = tid; j <= total; j += 256) {
        const int64_t s0 = W.job_start[base + j];
        const int64_t s1 = j < total ? W.job_start[base + j + 1] : o + n;
        W.job_n[base + j] = (int)(s1 - s0);
    }
}

// which of the two instances of chain_kernel runs this job (W.next[4]: 0 = the short ring, 1 = the long one).  The job is
// bound by its longest call when that call alone takes longer than the whole job spread over the calls the chip keeps
// in flight with the long ring (6 per CU): longest > anchors / (CUs x 6).  The longest call is known to within a factor
// of two from the size buckets; the test uses the bucket's lower bound.
__global__ void chain_pick_kernel(ChainWork W, long long n_anchors, int cus, int force)
{
    if (threadIdx.x || blockIdx.x) return;
    if (force >= 0) { W.next[4] = force; return; }
    int top = -1;                                           // bucket 0 holds the longest calls (bucket_of)
    for (int b = NBUCKET - 1; b >= 0; --b) if (W.counts[b] > 0) top = b;
    const long long longest = top < 0 ? 0 : 1ll << (NBUCKET - 1 - top);
    W.next[4] = longest * (long long)cus * 6 > n_anchors ? 1 : 0;
}

template <int RING_LIVE>
__global__ void __launch_bounds__(64) chain_kernel(int n_calls, const int64_t *__restrict__ off,
                                                   const uint64_t *__restrict__ ax, const uint64_t *__restrict__ ay,
                                                   const gbx_chain_call *__restrict__ hdr,
                                                   int32_t *score, int32_t *parent, int32_t *target, int32_t *peak,
                                                   ChainWork W)
{
    // ring entries: the anchor words {x, y} and the DP state {score, parent, target, peak} of the reference's four
    // vectors, 16 bytes each, so that a look-back chunk is two ds_read_b128 per lane.  Scores / parents / peaks go
    // to global memory once per block of 64 anchors (coalesced), targets when their block leaves the ring.
    constexpr int RING_PHYS = RING_LIVE + 64;
    if (W.next[4] != (RING_LIVE == GBX_CHAIN_RING_LIVE ? 0 : 1)) return;      // the other instance has this job
    __shared__ uint4 rxy[RING_PHYS];
    __shared__ int4 rst[RING_PHYS + 64];    // + a dump entry per lane for phase 4
    __shared__ int mark[128];               // [0,64) the chunk's marks, [64,128) dump slots of the lanes that mark nothing
    const int lane = threadIdx.x;
    const int max_iter = GBX_CHAIN_MAX_ITER, max_skip = GBX_CHAIN_MAX_SKIP;
    const unsigned long long lane_bit = 1ull << lane;

    // the resident blocks draw jobs from the longest-first list through a cursor: a dynamic LPT schedule.  Everything
    // derived from `slot` stays wave-uniform
    const int n_jobs = W.next[5];
    for (;;) {
        int slot = 0;
        if (lane == 0) slot = atomicAdd(&W.next[0], 1);
        slot = __builtin_amdgcn_readfirstlane(slot);
        if (slot >= n_jobs) break;
        const int job = W.order[slot];
        const int call = W.job_call[job];
        const int64_t o = W.job_start[job];                   // the job's anchors are [o, o + n) of the concatenated arrays
        const int n = W.job_n[job];
        const int cbase = (int)(o - off[call]);               // ... and [cbase, cbase + n) of their call: indices below are
                                                              // job-relative, parents and targets leave with cbase added
        const uint64_t *x = ax + o, *y = ay + o;
        int32_t *f = score + o, *p = parent + o, *t = target + o, *pk = peak + o;
        const int32_t *stp = W.st + o;
        const int flags = W.unsorted[call] | W.job_flag[job];
        const bool sorted = (flags & 1) == 0;
        unsigned long long visited = 0;
        // NARROW: one segment id and one upper x word in the whole call (and sorted x): `same` is always true and the
        // reference-position differences fit 32 unsigned bits, which takes a fifth of the instructions out of a chunk
        auto run_call = [&](auto narrow_tag) {
            constexpr bool NARROW = decltype(narrow_tag)::value;
            const gbx_chain_call h = hdr[call];
            const int max_dist_x = h.max_dist_x, max_dist_y = h.max_dist_y, bw = h.bw, n_segs = h.n_segs;
            const double avg_qspan = (double)h.avg_qspan;
            const uint64_t mdx = (uint64_t)(int64_t)max_dist_x;
            // narrow path (:59-80 with the conditions folded, see the chunk)
            const unsigned dq_lim = (unsigned)max(0, min(max_dist_x, max_dist_y));
            const unsigned dr_lim = n_segs > 1 ? (max_dist_y < 0 ? 0u : (unsigned)max_dist_y) : 0xffffffffu;

    #ifdef GBX_CHAIN_STAMPS
            unsigned long long acc_[12] = {0}, last_ = 0, n_chunks_ = 0, n_quiet_ = 0, n_scan_ = 0, n_break_ = 0;
    #endif
            int st = 0;
            int sb = 0;                                           // unsorted calls: block of 64 x-values cached for the st scan
            uint64_t xs = n ? x[min(lane, n - 1)] : 0;
            int sib = 0;                                          // ib mod RING_PHYS (blocks are slab-aligned: 320 = 5 x 64)
            // The anchors of block ib + 64 are requested while block ib is computed (round 4).  gfx9 counts loads and stores in
            // one in-order counter: three loads each settled on the spot at the head of a block were three serial round
            // trips, and each of those waits was also a wait for the acknowledgement of the previous block's output stores -
            // about 6 us per 64 anchors of a kernel whose longest job decides everything.  The anchors are requested a block
            // ahead, and a block's outputs are stored at the head of the NEXT block, behind the wait for its anchors (the
            // compiler cannot count the memory operations of the inner loops, so that wait is a vmcnt(0): it must not find a
            // young store in the queue - the ones it finds are a whole block old).
            uint64_t nxa = 0, nya = 0;
            int nstv = 0;
            if (n > 0) { const int ia0 = min(lane, n - 1); nxa = x[ia0]; nya = y[ia0]; nstv = stp[ia0]; }
            int pf_ = 0, pp_ = 0, pk_ = 0, pkmax = 0;             // the previous block's outputs, not yet stored
            // Round 5: the first look-back chunk of anchor i (anchors i-1 .. i-64, lane l = anchor i-1-l) is the first chunk of
            // anchor i-1 moved up by one lane, so it lives in registers: a wave_shr:1 per word after every anchor, the anchor's
            // own words entering at lane 0 - no LDS round trip in front of the chunk every anchor has.  The target word does
            // not travel: `targets[j] == i` in the first chunk can only come from this anchor's own marks (exchange_marks).
            unsigned wxl = 0, wxh = 0, wyl = 0, wyh = 0;
            int wf = 0, wp = -1, wk = 0;
            for (int ib = 0; ib < n; ib += 64, sib = sib + 64 == RING_PHYS ? 0 : sib + 64) {
                // this block's anchors, one per lane
                const uint64_t xa = settle(nxa), ya = settle(nya);
                const int stv = settle(nstv) - cbase;             // st(i) >= cbase for every anchor of the job
                if (lane < pkmax) { f[ib - 64 + lane] = pf_; p[ib - 64 + lane] = pp_ >= 0 ? pp_ + cbase : pp_; pk[ib - 64 + lane] = pk_; }
                if (ib + 64 < n) { const int ian = min(ib + 64 + lane, n - 1); nxa = x[ian]; nya = y[ian]; nstv = stp[ian]; }
                const int kmax = min(64, n - ib);
                const int live0 = ib - RING_LIVE;                 // anchors >= live0 are addressed in the ring during this block
                const int live_lo = live0 > 0 ? live0 : 0;
                // the slab this block fills holds the anchors [ib-320, ib-256): their targets are final (no anchor of this
                // or a later block marks them through the ring), so they go to the output now, one coalesced store
                if (ib >= RING_PHYS) t[ib - RING_PHYS + lane] = rst[sib + lane].z;
                rxy[sib + lane] = make_uint4((unsigned)xa, (unsigned)(xa >> 32), (unsigned)ya, (unsigned)(ya >> 32));
                int of_ = 0, op_ = 0, ok_ = 0;                    // outputs of the block's anchors, anchor ib+k in lane k
                for (int k = 0; k < kmax; ++k) {
                    const int i = ib + k, si = sib + k;           // si = i mod RING_PHYS
                    const int iabs = i + cbase;                   // the reference's i: what targets[] holds (:89)
                    STAMP_RESET();
                    const uint64_t ri = readlane64(xa, k), yi = readlane64(ya, k);
                    const int qi = (int)yi, q_span = (int)(yi >> 32 & 0xff);
                    const int sidi = (int)(yi >> 48 & 0xff);
                    if (sorted) {
                        st = __builtin_amdgcn_readlane(stv, k);
                    } else {
                        // advance st (:56): first st with ri <= x[st] + max_dist_x, scanning the cached block
                        GBX_GUARD(gd_st, n / 64 + 4);                              // (st moves up a block of 64 per pass)
                        for (;;) {
                            if (st >= i || GBX_GUARD_TRIP(gd_st, GBX_GK_CHAIN, 1, job)) break;
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
                    }
                    // ring slot of anchor a, i - 320 < a <= i
                    auto slot_of = [&](int a) -> int { const int v = si - (i - a); return v < 0 ? v + RING_PHYS : v; };

                    STAMP(0);
                    int max_f = q_span, max_j = -1, max_pk = 0, n_skip = 0, last_bl = 64;   // max_pk: peak of anchor max_j (:92)
                    // :89 for the lanes of a chunk visited before the break (lane < bl): targets[parents[j]] = i, in the ring while the
                    // parent's block is live, straight to the output (already flushed there) when it is older
                    auto mark_targets = [&](bool skip, int pj, int bl) {
                        // live_lo = max(live0, 0): "has a parent that is still in the ring" is one compare
                        const bool vis = !skip && lane < bl;
                        rst[vis && pj >= live_lo ? slot_of(pj) : RING_PHYS + lane].z = iabs;   // non-writers: their dump entry
                        if (live0 > 0 && __ballot(vis && pj >= 0 && pj < live0))            // rare: the parent left the ring
                            if (vis && pj >= 0 && pj < live0) t[pj] = iabs;
                    };
                    // One 64-wide chunk of the look-back, lane 0 = anchor jhi.  Returns true when the max_skip break fired.
                    // The anchor words and DP state of the chunk arrive as arguments, so that the ring path below is
                    // made of LDS reads only.
                    auto chunk = [&](int jhi, bool valid, uint64_t xj, uint64_t yj, int fj, int pj, int tj, int kj) -> bool {
                        // ---- phase 1: candidate score / `continue` mask (:59-80)
    #ifdef GBX_CHAIN_STAMPS
                        { unsigned lo_ = (unsigned)xj, a_ = (unsigned)fj, b_ = (unsigned)pj, c_ = (unsigned)tj, d_ = (unsigned)yj;
                          asm volatile("" :: "v"(lo_), "v"(a_), "v"(b_), "v"(c_), "v"(d_)); ++n_chunks_; }
    #endif
                        STAMP(1);
                        int dq, dd, min_d, log_dd, c_lin, sc, gap_cost = 0, marked = 0;
                        bool skip;
                        // ---- phase 2, issued as soon as the `continue` mask is known: was this j already marked as a parent
                        // during this i (:84)?  True iff an earlier visited, non-skipped lane's parent is this lane's anchor (or
                        // tj == i from an earlier chunk).  The marks travel from lane to lane through LDS inside one wavefront
                        // (its LDS operations are performed in order); the read is volatile so that the compiler does not
                        // forward this lane's own store of 0 to it.
                        auto exchange_marks = [&](bool skip_) -> int {
                            mark[lane] = 0;
                            const unsigned tl = (unsigned)(jhi - pj);      // lane that holds anchor pj, if inside this chunk
                            // tl > lane always: parents precede their children.  pj == -1 gives tl = jhi + 1: a lane past the
                            // job's first anchor (never valid, its mark is never looked at), so "has a parent" needs no test of
                            // its own.  Lanes with nothing to mark write to a slot of their own behind the 64 marks - one
                            // v_cndmask under the lanes' mask instead of the exec-mask changes the compiler makes of a select.
                            // (One ds_permute_b32 instead of the two writes and the read was measured: 70.4 against 69.5 ms.)
                            const unsigned long long mk = __ballot(!skip_ && tl < 64u);
                            unsigned slot_;
                            asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(slot_) : "v"(64u + (unsigned)lane), "v"(tl), "s"(mk));
                            mark[slot_] = 1;
                            return ((const volatile lds_int *)mark)[lane];    // typed LDS pointer: ds_read, not a FLAT load
                        };
                        if constexpr (NARROW) {
                            // dr in [0, 2^32) as unsigned (sorted x, equal upper words); every use below either compares
                            // it with a positive int (lanes with dq <= 0 are skipped) or truncates it to 32 bits as the
                            // reference's int32 assignments do
                            const unsigned dr = (unsigned)ri - (unsigned)xj;
                            dq = qi - (int)yj;
                            // dq <= 0 || dq > max_dist_y || dq > max_dist_x as one unsigned compare (dq_lim = min of the two,
                            // at least 0); the multi-segment clause as dr > dr_lim (all ones when n_segs <= 1)
                            skip = !valid || dr == 0 || (unsigned)(dq - 1) >= dq_lim;
                            const bool gt = dr > (unsigned)dq;
                            dd = gt ? (int)(dr - (unsigned)dq) : (int)((unsigned)dq - dr);
                            skip = skip || dd > bw;
                            skip = skip || dr > dr_lim;
                            marked = exchange_marks(skip);            // the LDS round trip runs under the arithmetic below
                            min_d = gt ? dq : (int)dr;
                            sc = min_d > q_span ? q_span : min_d;
                            log_dd = dd ? 31 - __builtin_clz((unsigned)dd) : 0;
                            c_lin = (int)((double)dd * .01 * avg_qspan);
                            gap_cost = c_lin + (log_dd >> 1);
                        } else {
                            const int64_t dr = (int64_t)(ri - xj);
                            dq = qi - (int)yj;
                            const int sidj = (int)(yj >> 48 & 0xff);
                            const bool same = sidi == sidj;
                            skip = !valid || (same && dr == 0) || dq <= 0;
                            skip = skip || (same && dq > max_dist_y) || dq > max_dist_x;
                            dd = (int)(dr > dq ? dr - dq : dq - dr);
                            skip = skip || (same && dd > bw);
                            skip = skip || (n_segs > 1 && same && dr > max_dist_y);
                            marked = exchange_marks(skip);            // the LDS round trip runs under the arithmetic below
                            min_d = dq < dr ? dq : (int)dr;
                            sc = min_d > q_span ? q_span : min_d;
                            log_dd = dd ? 31 - __builtin_clz((unsigned)dd) : 0;
                            c_lin = (int)((double)dd * .01 * avg_qspan);
                            if (!same) {
                                if (dr == 0) ++sc;
                                else gap_cost = c_lin < log_dd ? c_lin : log_dd;
                            } else {
                                gap_cost = c_lin + (log_dd >> 1);
                            }
                        }
                        // (int)((double)gap_cost * gap_scale + .499) with gap_scale = 1.0f is gap_cost itself: gap_cost >= 0
                        // for every lane that is not skipped (dd >= 0), and skipped lanes never use sc
                        sc -= gap_cost;
                        sc += fj;
    #ifdef GBX_CHAIN_STAMPS
                        { unsigned a_ = (unsigned)sc, b_ = skip; asm volatile("" :: "v"(a_), "v"(b_)); }
    #endif
                        STAMP(2);
                        const bool hit = (marked != 0) | (tj == iabs);
#ifdef GBX_CHAIN_STAMPS
                        { unsigned b_ = hit; asm volatile("" :: "v"(b_)); }
    #endif
                        STAMP(3);
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
                            if (GBX_CHAIN_QUIET && bm == 0) {
                                // ... and nobody is marked either: max_f, n_skip and the break are as they were, only :89 is left
                                last_bl = 64;
    #ifdef GBX_CHAIN_STAMPS
                                ++n_quiet_;
    #endif
                                mark_targets(skip, pj, 64);
                                return false;
                            }
                            const int below = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bm, 0u));
                            nl = n_skip + below + (bump ? 1 : 0);
                        } else {
    #ifdef GBX_CHAIN_STAMPS
                            ++n_scan_;
    #endif
                            const unsigned pm = wave_scan_umax(cand);                 // inclusive prefix max of candidates
                            unsigned pmx = (unsigned)dppi<0x138>(0, (int)pm);         // exclusive (wave_shr:1)
                            pmx = lane == 0 ? 0u : pmx;
                            improving = !skip && cand > max((unsigned)max_f + UBIAS, pmx);
                            bump = !skip && !improving && hit;
                            // S = n_skip + (bumps up to and including this lane) - (improving lanes likewise): two ballots and
                            // prefix popcounts instead of a third wave scan
                            const unsigned long long bmask = __ballot(bump), imask = __ballot(improving);
                            const int bcnt = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bmask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bmask, 0u));
                            const int icnt = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(imask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)imask, 0u));
                            const int S = n_skip + bcnt - icnt + (bump ? 1 : 0) - (improving ? 1 : 0);
                            // n_skip after this lane = the walk S reflected at 0: S - min(0, prefix-min S).  The usual shape -
                            // improving lanes only in front of the first bump - needs no scan: those improvements take n_skip
                            // down towards 0 and from there on it only counts bumps.
                            const unsigned long long first_b = bmask & (0 - bmask);                  // lowest bump bit (0 if none)
                            const unsigned long long lead = first_b ? first_b - 1 : ~0ull;           // lanes before the first bump
                            if ((imask & ~lead) == 0) {
                                const int base = max(0, n_skip - (int)__builtin_popcountll(imask));  // after the leading improvements
                                nl = (lane_bit & lead) ? max(0, n_skip - icnt - (improving ? 1 : 0)) : base + bcnt + (bump ? 1 : 0);
                            } else {
                                const unsigned mx = wave_scan_umax((unsigned)(-S) + UBIAS);          // min via a biased max of -S
                                const int mn = -(int)(mx - UBIAS);
                                nl = S - min(0, mn);
                            }
                        }
#ifdef GBX_CHAIN_STAMPS
                        { unsigned a_ = (unsigned)nl; asm volatile("" :: "v"(a_)); }
    #endif
                        STAMP(4);
                        const unsigned long long brk = __ballot(bump && nl > max_skip);
                        const int bl = brk ? __builtin_ctzll(brk) : 64;   // first breaking lane
                        const unsigned long long before = bl >= 64 ? ~0ull : ((1ull << bl) - 1);
                        last_bl = bl;
                        const unsigned long long imp = __ballot(improving) & before;
                        if (imp) {
                            const int li = 63 - __builtin_clzll(imp);  // last improving lane before the break
                            max_j = jhi - li;
                            max_f = __builtin_amdgcn_readlane(sc, li);
                            max_pk = __builtin_amdgcn_readlane(kj, li);    // travels with the chunk: no LDS round trip at the anchor's end
                        }
                        STAMP(5);
                        // ---- phase 4: targets[parents[j]] = i for lanes visited before the break (:89): in the ring while
                        // the parent's block is live, straight to the output (already flushed there) when it is older
                        mark_targets(skip, pj, bl);
                        n_skip = __builtin_amdgcn_readlane(nl, 63);
                        STAMP(6);
    #ifdef GBX_CHAIN_STAMPS
                        if (bl < 64) ++n_break_;
    #endif
                        return bl < 64;
                    };
                    int jhi = i - 1;
                    bool broke = false;
                    if (GBX_CHAIN_WINDOW && jhi >= st) {                       // the first chunk: the register window
                        broke = chunk(jhi, jhi - lane >= st, ((uint64_t)wxh << 32) | wxl, ((uint64_t)wyh << 32) | wyl, wf, wp, -1, wk);
                        if (!broke) jhi -= 64;
                    }
                    // chunks that lie inside the ring: LDS only
                    for (; !broke && jhi >= st && jhi - 63 >= live0; jhi -= 64) {
                        const int j = jhi - lane;
                        const bool valid = j >= st;
                        const int rs = slot_of(valid ? j : st);
                        const uint4 wxy = rxy[rs];
                        const int4 wst = rst[rs];
                        if ((broke = chunk(jhi, valid, ((uint64_t)wxy.y << 32) | wxy.x, ((uint64_t)wxy.w << 32) | wxy.z, wst.x, wst.y, wst.z, wst.w))) break;
                    }
                    // deeper chunks: the global arrays (blocks that left the ring are complete there; a chunk that straddles
                    // the ring's edge reads its newer lanes from the ring)
                    if (!broke)
                        for (; jhi >= st; jhi -= 64) {
                            const int j = jhi - lane;
                            const bool valid = j >= st;
                            const int jj = valid ? j : st;
                            uint64_t xj, yj; int fj, pj, tj, kj;
                            if (jj >= live0) {
                                const int rs = slot_of(jj);
                                const uint4 wxy = rxy[rs];
                                const int4 wst = rst[rs];
                                xj = ((uint64_t)wxy.y << 32) | wxy.x; yj = ((uint64_t)wxy.w << 32) | wxy.z; fj = wst.x; pj = wst.y; tj = wst.z; kj = wst.w;
                            } else {
                                xj = settle(x[jj]); yj = settle(y[jj]); fj = settle(f[jj]); pj = settle(p[jj]); tj = settle(t[jj]); kj = settle(pk[jj]);
                                pj = pj >= 0 ? pj - cbase : pj;   // the output holds call-relative parents
                            }
                            if (chunk(jhi, valid, xj, yj, fj, pj, tj, kj)) break;
                        }
                    // predecessors visited (the benchmark's cell count): everything down to st, or down to the breaking lane
                    visited += (unsigned long long)(last_bl < 64 ? (i - 1 - jhi) + last_bl + 1 : i - st);
                    STAMP(7);
                    // :91-92: into the block's output registers and the ring
                    const int pki = (max_j >= 0 && max_pk > max_f) ? max_pk : max_f;
                    const bool mine = lane == k;                  // one compare, three selects: no exec-mask change, no LDS
                    of_ = mine ? max_f : of_;
                    op_ = mine ? max_j : op_;
                    ok_ = mine ? pki : ok_;
                    if (lane == 0) rst[si] = make_int4(max_f, max_j, 0, pki);
                    if (GBX_CHAIN_WINDOW) {
                        wxl = (unsigned)dppi<0x138>((int)(unsigned)ri, (int)wxl); wxh = (unsigned)dppi<0x138>((int)(unsigned)(ri >> 32), (int)wxh);
                        wyl = (unsigned)dppi<0x138>((int)(unsigned)yi, (int)wyl); wyh = (unsigned)dppi<0x138>((int)(unsigned)(yi >> 32), (int)wyh);
                        wf = dppi<0x138>(max_f, wf); wp = dppi<0x138>(max_j, wp); wk = dppi<0x138>(pki, wk);
                    }
                    STAMP(8);
                }
                pf_ = of_; pp_ = op_; pk_ = ok_; pkmax = kmax;
            }
            // the last block's outputs, and the targets still in the ring: the anchors of the last five blocks
            if (n > 0) {
                const int ibp = (n - 1) & ~63;
                if (lane < pkmax) { f[ibp + lane] = pf_; p[ibp + lane] = pp_ >= 0 ? pp_ + cbase : pp_; pk[ibp + lane] = pk_; }
                const int ibl = (n - 1) & ~63;                    // start of the last block, whose slab is sib - 64 (mod 320)
                const int sibl = sib == 0 ? RING_PHYS - 64 : sib - 64;
                for (int a = max(0, ibl - RING_LIVE) + lane; a < n; a += 64) {
                    int v = sibl + (a - ibl);
                    v = v < 0 ? v + RING_PHYS : v;
                    t[a] = rst[v].z;
                }
            }
    #ifdef GBX_CHAIN_STAMPS
            if (lane == 0 && (unsigned long long)n >= g_chain_stamps[13]) {      // the longest call finishes last
                for (int k = 0; k < 9; ++k) g_chain_stamps[k] = acc_[k];
                g_chain_stamps[12] = n_chunks_; g_chain_stamps[13] = (unsigned long long)n;
                g_chain_stamps[9] = n_quiet_; g_chain_stamps[10] = n_scan_; g_chain_stamps[11] = n_break_;
            }
    #endif
        };
        if (flags == 0) run_call(std::true_type{}); else run_call(std::false_type{});
        (void)n_calls;
        if (lane == 0) atomicAdd(W.evaluated, visited);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    }
}

}  // namespace

// Workspace, in 8-byte words first: [header 3*NBUCKET ints | blk_diff[n_blk] u64 | job_start[max_jobs] i64 |
// order[max_jobs] | job_n | job_call | job_flag | spare target / peak planes 2 x n_anchors | st[n_anchors] |
// unsorted[n_calls] | blk_cut[n_blk] bytes]
struct ChainLayout {
    int64_t n_blk, max_jobs;
    size_t o_diff, o_jstart, o_order, o_jn, o_jcall, o_jflag, o_spare, o_st, o_unsorted, o_cut, total;
};
static ChainLayout chain_layout(int64_t n_calls, int64_t n_anchors)
{
    ChainLayout L;
    if (n_calls < 0) n_calls = 0;
    if (n_anchors < 0) n_anchors = 0;
    L.n_blk = n_anchors / 64 + n_calls + 2;                  // block index ((off + 64 w) >> 6) + call
    L.max_jobs = L.n_blk + n_calls;                          // a job per call + a cut per block at most
    size_t at = 3 * NBUCKET * sizeof(int32_t);               // 384 bytes: the 64-bit arrays that follow stay 8-byte aligned
    auto take = [&](size_t bytes) { const size_t o = at; at += (bytes + 7) & ~(size_t)7; return o; };
    L.o_diff = take((size_t)L.n_blk * 8);
    L.o_jstart = take((size_t)L.max_jobs * 8);
    L.o_order = take((size_t)L.max_jobs * 4);
    L.o_jn = take((size_t)L.max_jobs * 4);
    L.o_jcall = take((size_t)L.max_jobs * 4);
    L.o_jflag = take((size_t)L.max_jobs * 4);
    L.o_spare = take((size_t)n_anchors * 8);
    L.o_st = take((size_t)n_anchors * 4);
    L.o_unsorted = take((size_t)n_calls * 4);
    L.o_cut = take((size_t)L.n_blk);
    L.total = at + 64;
    return L;
}

size_t chain_workspace_bytes(int64_t n_calls, int64_t n_anchors) { return chain_layout(n_calls, n_anchors).total; }

#ifdef GBX_CHAIN_STAMPS
extern "C" int gbx_debug_chain_stamps(unsigned long long *out16)
{
    GBX_HIP(hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_chain_stamps), sizeof(unsigned long long) * 16));
    return GBX_OK;
}
#endif

int chain_read_evaluated(const void *d_work, int64_t *pairs, hipStream_t s)
{
    unsigned long long v = 0;
    GBX_HIP(hipMemcpyAsync(&v, (const char *)d_work + (2 * NBUCKET + 2) * sizeof(int32_t), sizeof(v), hipMemcpyDeviceToHost, s));
    GBX_HIP(hipStreamSynchronize(s));
    *pairs = (int64_t)v;
    return GBX_OK;
}

int chain_read_job_stats(const void *d_work, int64_t n_calls, int64_t n_anchors, int64_t *jobs, int64_t *longest, hipStream_t s)
{
    const ChainLayout L = chain_layout(n_calls, n_anchors);
    int32_t nj = 0;
    GBX_HIP(hipMemcpyAsync(&nj, (const char *)d_work + (2 * NBUCKET + 5) * sizeof(int32_t), sizeof(nj), hipMemcpyDeviceToHost, s));
    GBX_HIP(hipStreamSynchronize(s));
    *jobs = nj; *longest = 0;
    if (nj > 0) {
        int32_t first = 0, n = 0;                              // order[0] is a job of the fullest size bucket: the longest within 2x;
        std::vector<int32_t> jn((size_t)nj);                   // the exact maximum is read from the lengths (a diagnostic, not a hot path)
        GBX_HIP(hipMemcpyAsync(jn.data(), (const char *)d_work + L.o_jn, sizeof(int32_t) * (size_t)nj, hipMemcpyDeviceToHost, s));
        GBX_HIP(hipStreamSynchronize(s));
        for (int32_t v : jn) n = v > n ? v : n;
        (void)first;
        *longest = n;
    }
    return GBX_OK;
}

int chain_launch(int64_t n_calls, int64_t n_anchors, const int64_t *d_off,
                 const uint64_t *d_ax, const uint64_t *d_ay, const gbx_chain_call *d_hdr,
                 int32_t *d_score, int32_t *d_parent, int32_t *d_target, int32_t *d_peak,
                 void *d_work, size_t work_bytes, hipStream_t s)
{
    return chain_launch_skip(n_calls, n_anchors, d_off, d_ax, d_ay, d_hdr, d_score, d_parent, d_target, d_peak, d_work, work_bytes, s, nullptr);
}

int chain_launch_skip(int64_t n_calls, int64_t n_anchors, const int64_t *d_off,
                      const uint64_t *d_ax, const uint64_t *d_ay, const gbx_chain_call *d_hdr,
                      int32_t *d_score, int32_t *d_parent, int32_t *d_target, int32_t *d_peak,
                      void *d_work, size_t work_bytes, hipStream_t s, const uint8_t *d_skip)
{
    if (n_calls == 0) return GBX_OK;
    if (n_calls > 0x7fffffffLL - 1024) { set_error("chain: more than 2^31 calls"); return GBX_ERR_UNSUPPORTED; }
    if (work_bytes < chain_workspace_bytes(n_calls, n_anchors)) { set_error("chain: workspace too small"); return GBX_ERR_ARG; }
    if (n_anchors > 0x7fffffffLL * 32) { set_error("chain: too many anchors"); return GBX_ERR_UNSUPPORTED; }
    const ChainLayout L = chain_layout(n_calls, n_anchors);
    char *wb = (char *)d_work;
    int32_t *wi = (int32_t *)d_work;
    // header: [counts | cursors | next: job cursor, evaluated (u64 at +2), ring choice (+4), number of jobs (+5)]
    int32_t *spare = (int32_t *)(wb + L.o_spare);
    ChainWork W;
    W.counts = wi; W.cursors = wi + NBUCKET; W.next = wi + 2 * NBUCKET; W.order = (int32_t *)(wb + L.o_order);
    W.evaluated = (unsigned long long *)(wi + 2 * NBUCKET + 2);
    W.st = (int32_t *)(wb + L.o_st); W.unsorted = (int32_t *)(wb + L.o_unsorted);
    W.blk_cut = (int8_t *)(wb + L.o_cut); W.blk_diff = (unsigned long long *)(wb + L.o_diff);
    W.job_start = (int64_t *)(wb + L.o_jstart); W.job_n = (int32_t *)(wb + L.o_jn); W.job_call = (int32_t *)(wb + L.o_jcall);
    W.job_flag = (int32_t *)(wb + L.o_jflag); W.max_jobs = (int32_t)L.max_jobs;
    if (!d_target) d_target = spare;
    if (!d_peak) d_peak = spare + n_anchors;
    GBX_HIP(hipMemsetAsync(d_work, 0, 3 * NBUCKET * sizeof(int32_t), s));
    GBX_HIP(hipMemsetAsync(W.job_flag, 0, (size_t)L.max_jobs * sizeof(int32_t), s));
    // GBX_CHAIN_WIDE=1 (test aid): every job takes the general 64-bit / multi-segment path (flag bit 1 preset);
    // GBX_CHAIN_NOSPLIT=1 (test aid): a call is one job, as before the cuts existed
    const char *wide_env = getenv("GBX_CHAIN_WIDE");
    const char *nosplit_env = getenv("GBX_CHAIN_NOSPLIT");
    GBX_HIP(hipMemsetAsync(W.unsorted, wide_env && atoi(wide_env) ? 2 : 0, (size_t)n_calls * sizeof(int32_t), s));
    {
        // up to 8 slices of a call side by side: long calls are not left to one block (the longest has 60 000 anchors)
        Stage st("chain_st", s);
        hipLaunchKernelGGL(chain_st_kernel, dim3((unsigned)n_calls, 8), dim3(256), 0, s, (int)n_calls, d_off, d_ax, d_ay, d_hdr, W.st, W.unsorted,
                           W.blk_cut, W.blk_diff);
    }
    {
        Stage st("chain_order", s);
        hipLaunchKernelGGL(chain_jobs_kernel, dim3((unsigned)n_calls), dim3(256), 0, s, (int)n_calls, d_off, W,
                           nosplit_env && atoi(nosplit_env) ? 0 : 1, d_skip);
        const int ob = (int)((L.max_jobs + 255) / 256);
        hipLaunchKernelGGL(chain_order_kernel, dim3(ob), dim3(256), 0, s, W, 0);
        hipLaunchKernelGGL(chain_order_kernel, dim3(ob), dim3(256), 0, s, W, 1);
    }
    int dev_id = 0, cus = 256;
    (void)hipGetDevice(&dev_id);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev_id);
    // A persistent grid of as many blocks as the chip holds (per instance: its LDS ring decides), each drawing the next
    // job of the longest-first list from a cursor - a dynamic LPT schedule, as one block per call was, for a job count
    // only the device knows (a static stride over the list was 102 ms on the 'large' job against 97; calls in flight per
    // CU: 4: 149 ms, 8: 117, 16: 103).  GBX_CHAIN_WAVES_PER_CU caps the grid at that many blocks per CU instead.
    static int occ_short[CHAIN_MAX_DEVICES], occ_long[CHAIN_MAX_DEVICES];
    if (dev_id >= 0 && dev_id < CHAIN_MAX_DEVICES && occ_short[dev_id] == 0) {
        int a_ = 0, b_ = 0;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&a_, chain_kernel<GBX_CHAIN_RING_LIVE>, 64, 0);
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&b_, chain_kernel<GBX_CHAIN_RING_LIVE_LONG>, 64, 0);
        occ_short[dev_id] = a_ > 0 ? a_ : 8; occ_long[dev_id] = b_ > 0 ? b_ : 6;
    }
    const char *wenv = getenv("GBX_CHAIN_WAVES_PER_CU");
    const int wcap = wenv && atoi(wenv) > 0 ? atoi(wenv) : 1 << 20;
    const int di = dev_id >= 0 && dev_id < CHAIN_MAX_DEVICES ? dev_id : 0;
    const int64_t blocks_s = std::min<int64_t>(L.max_jobs, (int64_t)cus * std::min(wcap, occ_short[di]));
    const int64_t blocks_l = std::min<int64_t>(L.max_jobs, (int64_t)cus * std::min(wcap, occ_long[di]));
    {
        Stage st("chain_dp", s);
        // both instances are queued; the one the job did not pick returns at once (no host round trip for the choice)
        const char *renv = getenv("GBX_CHAIN_RING");         // test / tuning aid: "short" or "long" for every job
        const int force = renv ? (renv[0] == 'l' ? 1 : renv[0] == 's' ? 0 : -1) : -1;
        hipLaunchKernelGGL(chain_pick_kernel, dim3(1), dim3(64), 0, s, W, (long long)n_anchors, cus, force);
        hipLaunchKernelGGL(chain_kernel<GBX_CHAIN_RING_LIVE>, dim3((unsigned)blocks_s), dim3(64), 0, s, (int)n_calls, d_off, d_ax, d_ay, d_hdr,
                           d_score, d_parent, d_target, d_peak, W);
        hipLaunchKernelGGL(chain_kernel<GBX_CHAIN_RING_LIVE_LONG>, dim3((unsigned)blocks_l), dim3(64), 0, s, (int)n_calls, d_off, d_ax, d_ay, d_hdr,
                           d_score, d_parent, d_target, d_peak, W);
    }
    GBX_HIP(hipGetLastError());
    GBX_GUARD_CHECK("chain");
    return GBX_OK;
}

}  // namespace gbx
