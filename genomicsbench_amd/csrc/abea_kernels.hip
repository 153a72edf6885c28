// abea_kernels.hip — adaptive banded event alignment (f5c / nanopolish) for gfx950 (MI355X).
//
// Semantics: align(), R/benchmarks/abea/src/align.c:169-548 (the suite's CPU path; called per read by
// align_single, f5c.c:1344-1349), bit for bit: float band scores and emissions, double transition penalties
// (every candidate is a double sum rounded to float, :371-373), Suzuki's band placement (:289-307), the trim
// column (:310-319), the traceback with its double emission sum (:409-500) and the three QC rules (:530-541).
// The suite's CUDA path for this step (align.cu) is not a template for this file: it uses one 128-thread block
// per read, a rolling band buffer in shared memory with a block barrier per band, and float penalties.
//
// One read per wavefront, nothing on the way of a band but registers:
//   * a band is an anti-diagonal of 100 cells; lane l owns the cells at offsets 2l and 2l+1.  The previous band and the
//     "up" / "left" arrays it was computed from stay in registers: a band's "up" and "left" are the previous band
//     shifted by +1 / 0 or 0 / -1 offsets depending on whether it moved right or down (own other cell, or the neighbour
//     lane's by DPP wave_shl / wave_shr), and its diagonal is the previous band's "up" (after a move right) or "left"
//     (after a move down) unshifted;
//   * the event means and the scaled model parameters of the lane's cells ride along in registers: a move to the right
//     shifts the k-mer parameters by one offset, a move down the event means; the one value that enters comes from a
//     64-entry look-ahead window held one per lane.  Windows are loaded per block of 16 bands, two blocks ahead;
//   * a lone wavefront issues about one instruction every 4-5 clocks whatever its type, and the longest read of a batch
//     runs alone at the end, so the band loop is written for instruction count and for never waiting: no load, no
//     vector-memory wait (gfx9 counts loads and stores in one in-order counter: a wait for a fresh load is a wait for
//     every store before it), back-pointers staged in LDS and flushed 1 KB per 16 bands, an interior-band variant
//     without bounds / trim / end-cell logic, the division reduced to its five final operations on the packed-float
//     pipe when the read's values allow (see mid_range), the longest reads at a higher wave priority;
//   * per band the wavefront keeps 64 trace bytes: two 2-bit back-pointers per lane and two bits saying whether this
//     band and the one before it moved right, which is all the traceback needs to follow the band corners; band scores
//     are never stored - the best end cell on the last k-mer's column (:416-432) is tracked on the fly;
//   * the traceback is a scalar walk over register-resident trace blocks that records 2-bit moves, then a parallel
//     pass that rebuilds the pairs from the moves, reverses them into the caller's array and sums the emissions in
//     walk order.
// The double penalties lp_stay / lp_step depend on the read (events per k-mer) and are computed on the host with the
// C library's log / exp (gbx_abea_plan_host), exactly as the reference does: device transcendental functions are not
// bit-identical to glibc's.
#include "gbx_internal.h"
#include <cmath>
#include <type_traits>

namespace gbx {
namespace {

constexpr int BW = GBX_ABEA_BANDWIDTH;
constexpr int KSZ = GBX_ABEA_KMER;
constexpr int ROW = 64;                       // trace bytes per band (one per lane)
#define ABEA_NEG_INF (-__builtin_inff())
typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
#ifndef GBX_ABEA_ASM_DIV
#define GBX_ABEA_ASM_DIV 1
#endif
typedef __attribute__((address_space(3))) uint8_t lds_u8;
typedef __attribute__((address_space(3))) v4u lds_v4u;

struct AbeaArgs {
    int n_reads;
    const int64_t *seq_off; const int32_t *seq_len; const char *seq;
    const int64_t *event_off; const float *event_mean;
    const gbx_abea_model *models;
    const float *scale, *shift;
    const int64_t *band_off; const int32_t *order;
    const double *lp;                         // [n_reads][2]: lp_stay, lp_step (host-computed)
    double lp_skip, lp_trim;
    gbx_abea_pair *out; int32_t *n_pairs;
    float4 *kp;                               // [n_kmers_total] {scaled mean, stdv, log_inv_sqrt_2pi - log stdv, 0} per k-mer
    uint8_t *trace;                           // [n_bands_total][ROW]
    unsigned *cursor; unsigned long long *cells;
    unsigned prio_cut;                        // number of SIMDs: reads [0, prio_cut) of the order get priority 3, ...
};

template <int CTRL>
__device__ inline float dppf(float old, float x)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, false));
}
__device__ inline float shr1(float x, float fill) { return dppf<0x138>(fill, x); }     // lane l <- lane l-1 (lane 0: fill)
__device__ inline float shl1(float x, float fill) { return dppf<0x130>(fill, x); }     // lane l <- lane l+1 (lane 63: fill)
__device__ inline float rlf(float v, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); }

// 0, or a float whose magnitude lies in [2^-40, 2^40]
__device__ inline bool mid_range(float x)
{
    const unsigned m = __builtin_bit_cast(unsigned, x) & 0x7fffffffu;
    return m == 0u || (m >= ((127u - 40u) << 23) && m <= ((127u + 40u) << 23));
}

__device__ inline uint32_t base_rank(char b) { return b == 'C' ? 1u : b == 'G' ? 2u : b == 'T' ? 3u : 0u; }   // align.c:10-24

// the DP of one band.  FAST (an interior band: every offset 0..BW-1 is a cell of the matrix, no trim cell, not on the
// last k-mer's column) drops the per-band bounds, the trim cell, the end-cell test and the cell count.
struct AbeaBest { float s; int ev, off, e_default; };

__global__ void __launch_bounds__(64, 4) abea_kernel(AbeaArgs A)
{
    __shared__ __attribute__((aligned(16))) uint8_t tring[16 * ROW];      // the last (up to) 16 trace rows, flushed as 1 KB
    const int lane = threadIdx.x;
    const float NINF = ABEA_NEG_INF;
    const int NINF_BITS = (int)0xff800000u;
    const bool lane_in = lane < BW / 2;                                   // both of the lane's offsets are < BW
    unsigned long long fills = 0;
#ifdef GBX_ABEA_PHASE_STATS                                                    // development aid, scripts/dbg_abea_phases.py
    unsigned long long cyc_pre = 0, cyc_dp = 0, cyc_tb = 0, cyc_p2 = 0, steps = 0;
#define ABEA_STAMP(t) const unsigned long long t = __builtin_readcyclecounter()
#else
#define ABEA_STAMP(t) do { } while (0)
#endif
    for (;;) {
        unsigned q = 0;
        if (lane == 0) q = atomicAdd(A.cursor, 1u);
        q = (unsigned)__builtin_amdgcn_readfirstlane((int)q);
        if (q >= (unsigned)A.n_reads) break;
        const int r = A.order[q];
        // longest reads first (the order is by decreasing length): the first wavefront to land on a SIMD outranks the
        // later ones, so the reads that decide the makespan run at a lone wavefront's pace from the start
        if (q < A.prio_cut) __builtin_amdgcn_s_setprio(3);
        else if (q < 2 * A.prio_cut) __builtin_amdgcn_s_setprio(2);
        else if (q < 3 * A.prio_cut) __builtin_amdgcn_s_setprio(1);
        else __builtin_amdgcn_s_setprio(0);
        ABEA_STAMP(t0);
        const char *seq = A.seq + A.seq_off[r];
        const int n_kmers = A.seq_len[r] - KSZ + 1;
        const int64_t ev0 = A.event_off[r];
        const int n_events = (int)(A.event_off[r + 1] - ev0);
        const float *evm = A.event_mean + ev0;
        const int64_t boff = A.band_off[r];
        const int64_t koff = boff - (ev0 - A.event_off[0]) - 2 * (int64_t)r;      // bands = events + k-mers + 2 per read
        float4 *kp = A.kp + koff;
        uint8_t *trace = A.trace + (boff + 32 * (int64_t)r) * ROW;      // 32 rows of slack per read
        const int n_bands = n_events + n_kmers + 2;
        const double lp_stay = A.lp[2 * r], lp_step = A.lp[2 * r + 1], lp_skip = A.lp_skip, lp_trim = A.lp_trim;
        const float scale = A.scale[r], shift = A.shift[r];

        bool tame = true;
        // ---- scaled model parameters per k-mer (kmer_ranks, :214-222 + log_probability_match_r9's per-state terms)
        // (round 4: the six bases of a k-mer come as one 8-byte load, the next k-mer's requested before this one's model is
        // looked up - six byte loads, each waited for, then the model: seven serial round trips per 64 k-mers, 3 ms of the
        // longest read's 42)
        const int slen = n_kmers + KSZ - 1;
        auto kmer_bytes = [&](int k) -> uint64_t {
            uint64_t v = 0;
            if (k + 8 <= slen) __builtin_memcpy(&v, seq + k, 8);
            else for (int i = 0; i < KSZ; ++i) v |= (uint64_t)(uint8_t)seq[k + i] << (8 * i);      // the read's last k-mers: no byte past its end
            return v;
        };
        auto rank_of = [](uint64_t kb) -> uint32_t {
            uint32_t rank = 0;
#pragma unroll
            for (int i = 0; i < KSZ; ++i) rank += base_rank((char)(kb >> (8 * (KSZ - i - 1)))) << (i << 1);
            return rank;
        };
        // two stages: the bases of k-mer k + 128 and the model of k-mer k + 64 are on their way while k-mer k is stored (its
        // store is issued after them, so waiting for them does not wait for it)
        uint64_t nextb = lane + 64 < n_kmers ? kmer_bytes(lane + 64) : 0;
        gbx_abea_model nextm = A.models[lane < n_kmers ? rank_of(kmer_bytes(lane)) : 0];
        for (int k = lane; k < n_kmers; k += 64) {
            const gbx_abea_model m = nextm;
            const uint64_t kb1 = nextb;
            if (k + 128 < n_kmers) nextb = kmer_bytes(k + 128);
            if (k + 64 < n_kmers) nextm = A.models[rank_of(kb1)];
            const float gm = scale * m.level_mean + shift, gs = m.level_stdv * 1;
            // the refined reciprocal the float division starts from (v_rcp_f32 + one Newton step, exactly the first
            // three operations of the compiler's own x / gs), hoisted from the cells to the k-mer
            const float r0 = __builtin_amdgcn_rcpf(gs);
            const float r1 = __builtin_fmaf(__builtin_fmaf(-gs, r0, 1.0f), r0, r0);
            // log_inv_sqrt_2pi - gp_log_stdv is the first operation of :147
            kp[k] = make_float4(gm, gs, -0.918938f - m.level_log_stdv, r1);
            tame = tame && mid_range(gm) && mid_range(gs) && gs > 0.f;
        }
        {
            int i = lane;
            for (; i + 192 < n_events; i += 256) {              // four independent loads per trip
                const float a = evm[i], b = evm[i + 64], c = evm[i + 128], d = evm[i + 192];
                tame = (int)tame & (int)mid_range(a) & (int)mid_range(b) & (int)mid_range(c) & (int)mid_range(d);      // (no short circuit: the four loads stay in flight together)
            }
            for (; i < n_events; i += 64) tame = tame && mid_range(evm[i]);
        }
        // every event, mean and stdv of the read is 0 or within 2^+-40 (stdv > 0): then no division of the read needs
        // v_div_scale's rescaling or v_div_fixup's special cases, and the five operations left of it are the division
        const bool fdiv = __builtin_amdgcn_ballot_w64(!tame) == 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        ABEA_STAMP(t1);
        // ---- bands 0 and 1 (:264-279)
        int e_b = BW / 2 - 1, k_b = -1 - BW / 2;                 // lower-left corner of band 0
        const int oA = 2 * lane, oB = 2 * lane + 1;
        const float p0a = oA == -1 - k_b ? 0.0f : NINF, p0b = oB == -1 - k_b ? 0.0f : NINF;      // band 0: the start cell
        tring[lane * 16] = 0;
        e_b += 1;                                                // band 1 = move_down(band 0)
        float p1a = oA == e_b ? (float)lp_trim : NINF, p1b = oB == e_b ? (float)lp_trim : NINF;   // first event trimmed
        tring[lane * 16 + 1] = (uint8_t)((oA == e_b ? 1 : 0) | (oB == e_b ? 4 : 0));    // FROM_U
        // "up" and "left" of band 1 (a move down: up = band 0 at the same offset, left = one offset lower).  The
        // diagonal of a band is the previous band's "up" after a move to the right and its "left" after a move down:
        // (e-1,k-1) is the up-neighbour of (e,k-1) and the left-neighbour of (e-1,k) - no shift, no third band kept.
        float upA = p0a, upB = p0b, lfA = shr1(p0b, NINF), lfB = p0a;
        // registers that ride along: event means and k-mer parameters of the lane's two cells (all 128 offsets).
        // Indices outside the read are clamped: such values only ever reach cells that are not filled.
        auto eload = [&](int idx) -> float { return evm[min(max(idx, 0), n_events - 1)]; };
        auto kload = [&](int idx) -> float4 { return kp[min(max(idx, 0), n_kmers - 1)]; };
        // (x = the lane's cell A, y = cell B: pairs, because both cells go through the packed-float pipe together)
        v2f ev = {eload(e_b - oA), eload(e_b - oB)}, km, ks, kc, kr;
        { const float4 t = kload(k_b + oA); km.x = t.x; ks.x = t.y; kc.x = t.z; kr.x = t.w; }
        { const float4 t = kload(k_b + oB); km.y = t.x; ks.y = t.y; kc.y = t.z; kr.y = t.w; }
        // look-ahead windows: lane j holds the value that enters j moves after the window's base.  A window serves one
        // block of 16 bands and is loaded two blocks ahead, at the top of a block (a block moves at most 16 times in
        // either direction, the window holds 64).  The band loop contains no load and no wait on the vector-memory
        // counter; the one wait per block, at its end, is for loads issued 16 bands earlier and leaves the block's own
        // trace store in flight (gfx9 counts loads and stores in one in-order counter, so a wait for a fresh load
        // would also wait for every store before it).
        int ew_base = e_b + 1, kw_base = k_b + 128;              // first event / k-mer index not yet in the registers
        float ewc = eload(ew_base + lane);
        float4 kwc = kload(kw_base + lane);
        int en_base = ew_base, kn_base = kw_base;                // the windows of the next block
        // everything loaded so far is waited for here, once, so that no wait is left inside the band loop
        asm volatile("" : "+v"(ewc), "+v"(kwc.x), "+v"(kwc.y), "+v"(kwc.z), "+v"(kwc.w), "+v"(ev), "+v"(km), "+v"(ks), "+v"(kc), "+v"(kr));
        float ewn = ewc;
        float4 kwn = kwc;
        AbeaBest best = {NINF, 0, -1, 0};                        // :416-432, tracked on the fly
        bool best_found = false;
        unsigned fills_r = 0, n_fast = 0;
        bool prev_right = false;                                 // band 1 moved down

        auto step = [&](auto tag, auto divtag, const int b) __attribute__((always_inline)) {
            constexpr bool FAST = decltype(tag)::value, FDIV = decltype(divtag)::value;
            // ---- placement of the band (:289-307)
            const int lli = __builtin_amdgcn_readlane(__builtin_bit_cast(int, p1a), 0);
            const int uri = __builtin_amdgcn_readlane(__builtin_bit_cast(int, p1b), BW / 2 - 1);
            // both corners out of band: alternate; else towards the better corner (-inf < -inf is false, so one "or")
            const bool both_ob = (lli == NINF_BITS) & (uri == NINF_BITS);
            const bool right = (__builtin_bit_cast(float, lli) < __builtin_bit_cast(float, uri)) | (both_ob & ((b & 1) == 1));
            float dgA, dgB;
            if (right) {
                k_b += 1;
                // k-mer parameters move one offset down; offset 127 takes the next one of the look-ahead block
                const int j = k_b + 127 - kw_base;
                const float im = rlf(kwc.x, j), is = rlf(kwc.y, j), ic = rlf(kwc.z, j);
                km = (v2f){km.y, shl1(km.x, im)}; ks = (v2f){ks.y, shl1(ks.x, is)}; kc = (v2f){kc.y, shl1(kc.x, ic)};
                if constexpr (FDIV) kr = (v2f){kr.y, shl1(kr.x, rlf(kwc.w, j))};
                dgA = upA; dgB = upB;                            // up = band[b-1][o+1], left = band[b-1][o]
                upA = p1b; upB = shl1(p1a, NINF);
                lfA = p1a; lfB = p1b;
            } else {
                e_b += 1;
                const float ie = rlf(ewc, e_b - ew_base);
                ev = (v2f){shr1(ev.y, ie), ev.x};
                dgA = lfA; dgB = lfB;                            // up = band[b-1][o], left = band[b-1][o-1]
                upA = p1a; upB = p1b;
                lfB = p1a; lfA = shr1(p1b, NINF);
            }

            // ---- the cells this band may fill (:323-332) and the trim cell (:310-319)
            int min_off = 0, max_off = BW, trim_off = -1;
            bool trim_ok = false;
            float trim_val = NINF;
            if constexpr (!FAST) {
                min_off = max(max(0 - k_b, e_b - (n_events - 1)), 0);
                max_off = min(min(n_kmers - k_b, e_b + 1), BW);
                trim_off = -1 - k_b;
                const int trim_ev = e_b - trim_off;
                trim_ok = trim_off >= 0 && trim_off < BW && trim_ev >= 0 && trim_ev < n_events;
                if (trim_ok) trim_val = (float)(lp_trim * (double)((int64_t)trim_ev + 1));
                fills_r += (unsigned)max(max_off - min_off, 0);
            }
            // lp_match (:143-147) of the lane's two cells
            float lpeA, lpeB;
            // the candidates' first sums, (double)diag + lp_step and (double)up + lp_stay (:371-372), do not depend on the emission
            double pdA = 0, puA = 0, pdB = 0, puB = 0;
            bool pre_sums = false;
            if constexpr (FDIV && GBX_ABEA_ASM_DIV) {
                // The packed division chain (see the branch below) is eight dependent v_pk_* operations, and a dependent packed
                // operation needs a wait state: the compiler filled seven of them with s_nop - issue slots of their own for a
                // lone wavefront (scripts/issue_cost.hip), a tenth of the longest read's band.  Written out here with the
                // conversions and first sums of the candidates in those slots.
                v2f a, q0, t, q1, q, h, l;
                asm("v_pk_add_f32 %[a], %[ev], %[km] neg_lo:[0,1] neg_hi:[0,1]\n"
                    "v_cvt_f64_f32 %[pdA], %[dgA]\n"
                    "v_pk_mul_f32 %[q0], %[a], %[kr]\n"
                    "v_cvt_f64_f32 %[puA], %[upA]\n"
                    "v_pk_fma_f32 %[t], %[ks], %[q0], %[a] neg_lo:[1,0,0] neg_hi:[1,0,0]\n"
                    "v_add_f64 %[pdA], %[pdA], %[step]\n"
                    "v_pk_fma_f32 %[q1], %[t], %[kr], %[q0]\n"
                    "v_add_f64 %[puA], %[puA], %[stay]\n"
                    "v_pk_fma_f32 %[t], %[ks], %[q1], %[a] neg_lo:[1,0,0] neg_hi:[1,0,0]\n"
                    "v_cvt_f64_f32 %[pdB], %[dgB]\n"
                    "v_pk_fma_f32 %[q], %[t], %[kr], %[q1]\n"
                    "v_cvt_f64_f32 %[puB], %[upB]\n"
                    "v_pk_mul_f32 %[h], %[q], -0.5 op_sel_hi:[1,0]\n"
                    "v_add_f64 %[pdB], %[pdB], %[step]\n"
                    "v_pk_mul_f32 %[h], %[h], %[q]\n"
                    "v_add_f64 %[puB], %[puB], %[stay]\n"
                    "v_pk_add_f32 %[l], %[kc], %[h]\n"
                    : [a] "=&v"(a), [q0] "=&v"(q0), [t] "=&v"(t), [q1] "=&v"(q1), [q] "=&v"(q), [h] "=&v"(h), [l] "=&v"(l),
                      [pdA] "=&v"(pdA), [puA] "=&v"(puA), [pdB] "=&v"(pdB), [puB] "=&v"(puB)
                    : [ev] "v"(ev), [km] "v"(km), [kr] "v"(kr), [ks] "v"(ks), [kc] "v"(kc), [dgA] "v"(dgA), [upA] "v"(upA), [dgB] "v"(dgB), [upB] "v"(upB),
                      [step] "v"(lp_step), [stay] "v"(lp_stay));
                lpeA = l.x; lpeB = l.y;
                pre_sums = true;
            } else if constexpr (FDIV) {
                // both cells at once on the packed-float pipe: q = a / gs as the compiler's division computes it once
                // its rescaling is the identity - a first quotient and two residual corrections against gs
                const v2f a = ev - km;
                const v2f q0 = a * kr;
                const v2f q1 = __builtin_elementwise_fma(__builtin_elementwise_fma(-ks, q0, a), kr, q0);
                const v2f q = __builtin_elementwise_fma(__builtin_elementwise_fma(-ks, q1, a), kr, q1);
                const v2f l = kc + (-0.5f * q) * q;
                lpeA = l.x; lpeB = l.y;
            } else {
                const float aA = (ev.x - km.x) / ks.x, aB = (ev.y - km.y) / ks.y;
                lpeA = kc.x + (-0.5f * aA * aA);
                lpeB = kc.y + (-0.5f * aB * aB);
            }
            auto cell = [&](int o, float diag, float up, float left, float lpe, double pd, double pu, float &val, int &from) {
                const float score_d = (float)((pre_sums ? pd : (double)diag + lp_step) + (double)lpe);       // :371-373
                const float score_u = (float)((pre_sums ? pu : (double)up + lp_stay) + (double)lpe);
                const float score_l = (float)((double)left + lp_skip);
                float max_score = score_d;
                int f = 0;                                                                    // FROM_D
                if constexpr (FDIV) {
                    // a read that passed the range check has no NaN anywhere (finite emissions, scores finite or -inf),
                    // and without NaNs `u > m ? u : m` is the hardware maximum
                    max_score = __builtin_fmaxf(score_u, max_score);
                    f = max_score == score_u ? 1 : f;                                         // FROM_U
                    max_score = __builtin_fmaxf(score_l, max_score);
                    f = max_score == score_l ? 2 : f;                                         // FROM_L
                } else {
                    max_score = score_u > max_score ? score_u : max_score;
                    f = max_score == score_u ? 1 : f;                                         // FROM_U
                    max_score = score_l > max_score ? score_l : max_score;
                    f = max_score == score_l ? 2 : f;                                         // FROM_L
                }
                if constexpr (FAST) {
                    val = lane_in ? max_score : NINF;
                    from = f;                                    // the bytes of offsets >= BW are never read
                } else {
                    const bool fill = o >= min_off && o < max_off;
                    const bool trim = trim_ok && o == trim_off;
                    val = fill ? max_score : trim ? trim_val : NINF;
                    from = fill ? f : trim ? 1 : 0;
                }
            };
            float na, nb;
            int fa, fb;
            cell(oA, dgA, upA, lfA, lpeA, pdA, puA, na, fa);
            cell(oB, dgB, upB, lfB, lpeB, pdB, puB, nb, fb);
            // two back-pointers per lane; bits 4 and 5 of every byte of the row say "this band / the band before it
            // moved to the right", which is how the traceback follows the band corners without a per-band corner array
            tring[lane * 16 + (b & 15)] = (uint8_t)(fa | (fb << 2) | (right ? 16 : 0) | (prev_right ? 32 : 0));
            prev_right = right;

            // ---- best end on the last k-mer's column (:416-432): the cell (event, n_kmers-1) of this band, if any
            if constexpr (!FAST) {
                const int ev = b - n_kmers - 1, ol = n_kmers - 1 - k_b;
                if (ev == 0) best.e_default = e_b;                                      // band n_kmers + 1
                if (ev >= 0 && ev < n_events && ol >= 0 && ol < BW) {                   // e_b - ev == ol on this band
                    const float v = (ol & 1) ? rlf(nb, ol >> 1) : rlf(na, ol >> 1);
                    const float s = (float)((double)v + (double)(size_t)(n_events - ev) * lp_trim);
                    if (s > best.s) { best.s = s; best.ev = ev; best.off = ol; best_found = true; }
                }
            }
            p1a = na; p1b = nb;
        };
        // the ring -> trace block row0/16 in HBM: 1 KB, lane l's 16 bytes are column l (the lane's two cells) of the
        // 16 bands - the layout the traceback reads back.  Always whole: a read's trace has 16 rows of slack.
        auto flush = [&](int row0) {
            const v4u v = *(volatile lds_v4u *)((lds_u8 *)tring + lane * 16);
            *(v4u *)(trace + (int64_t)row0 * ROW + lane * 16) = v;
        };

        auto run_bands = [&](auto divtag) __attribute__((always_inline)) {
        for (int b0 = 0; b0 < n_bands; b0 += 16) {
            const int el_base = e_b + 1, kl_base = k_b + 128;    // the windows of the block after the next
            const float ewl = eload(el_base + lane);
            const float4 kwl = kload(kl_base + lane);
            const int bend = min(b0 + 16, n_bands);
            int b = max(b0, 2);
            while (b < bend) {
                // m: how many bands from here on are interior whatever way they move (both corners only ever grow)
                int m = 0;
                if (k_b >= 0 && e_b >= BW - 1) m = min(min(n_kmers - BW - 1 - k_b, n_events - 1 - e_b), bend - b);
                if (b <= n_kmers + 1) m = min(m, n_kmers + 1 - b);   // band n_kmers + 1 takes the general path (e_default)
                if (m > 0) {
                    for (int i = 0; i < m; ++i, ++b) step(std::true_type{}, divtag, b);
                    n_fast += (unsigned)m;
                } else {
                    step(std::false_type{}, divtag, b);
                    ++b;
                }
            }
            flush(b0);
            ewc = ewn; kwc = kwn; ew_base = en_base; kw_base = kn_base;
            ewn = ewl; kwn = kwl; en_base = el_base; kn_base = kl_base;
        }
        };
        if (fdiv) run_bands(std::true_type{});
        else run_bands(std::false_type{});
        fills += (unsigned long long)fills_r + (unsigned long long)n_fast * BW;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        ABEA_STAMP(t2);
        // ---- traceback (:409-500), in two passes.
        // (1) The walk: serial, wave-uniform, on the scalar unit but for one v_readlane per step.  A 16-band trace block
        // is four registers per lane (lane = column, register = 4 bands, byte = band); four blocks are resident and a
        // slot is reloaded with the block four further down the moment the walk leaves it.  All the walk keeps of a step
        // is its 2-bit move; the (at most 16) moves made inside a block leave as one 8-byte record, stored to the top of
        // the read's own trace area, which the walk has left behind.  One load and one store per block, both
        // unconditional, so that the vector-memory counter stays countable and the walk never waits for a fresh load.
        const int cap = 2 * n_events;
        const int ce0 = best.ev, ck0 = n_kmers - 1;
        int ce = ce0, ck = ck0, n_out = 0, curr_gap = 0, max_gap = 0, last_ck = -1, n_blk = 0;
        int off = best_found ? best.off : best.e_default - ce;  // offset of (ce, ck) in its band
        int bi = (ce + 1) + (ck + 1);
        uint8_t *const tmp_top = trace + ((((int64_t)n_bands + 15) & ~(int64_t)15) + 16) * ROW;
        auto load_block = [&](int kb) -> v4u { return *(const v4u *)(trace + (int64_t)max(kb, 0) * (16 * ROW) + lane * 16); };
        unsigned codes = 0;
        int cnt = 0;
        GBX_GUARD(gd_tb, (long long)n_bands + 8);              // a step leaves a band or two: the walk is at most n_bands long
        auto walk_group = [&](const unsigned comp, const int grp) __attribute__((always_inline)) {   // bands 4 grp .. 4 grp + 3
            while ((ck | ce) >= 0 && (bi >> 2) == grp) {
                const unsigned byte = (unsigned)__builtin_amdgcn_readlane((int)comp, off >> 1) >> ((bi & 3) << 3);
                const unsigned from = (byte >> ((off & 1) << 1)) & 3u;      // 0 diagonal, 1 up, 2 left (:455-470)
#ifdef GBX_LOOP_GUARD
                // (3 is no move: the band loop never writes it; a walk that met one would stand still for ever.  Guard build only: in
                // the product build this one test cost the kernel two spilled vector registers and a fifth of its speed - 48.5 against
                // 39.6 ms on 'large' - although the walk is a hundredth of a read's time: the band loop's allocation shifted)
                if (from == 3u || GBX_GUARD_TRIP(gd_tb, GBX_GK_ABEA, 1, r)) { ck = -1; break; }
#endif
                codes |= from << (cnt << 1);
                cnt += 1;
                last_ck = ck;
                // branch-free: the k-mer moves unless "up", the event unless "left"; the offset in the band is (event
                // index of the band's lower-left corner) - event, and a band that moved down raised its corner
                const int dk = (int)(~from & 1u), de = (int)(~(from >> 1) & 1u);
                const int rflag = (int)((byte >> 4) & 1u), rprev = (int)((byte >> 5) & 1u);
                ck -= dk; ce -= de; bi -= dk + de;
                off += rflag + (dk & de & rprev) - dk;
                curr_gap = (curr_gap + 1) & -(int)(from >> 1);
                max_gap = max(curr_gap, max_gap);
            }
        };
        {
            int kb = bi >> 4;
            v4u ring0 = load_block(kb), ring1 = load_block(kb - 1), ring2 = load_block(kb - 2), ring3 = load_block(kb - 3);
            auto walk_block = [&](v4u &slot) __attribute__((always_inline)) -> bool {
                codes = 0; cnt = 0;
                walk_group(slot.w, kb * 4 + 3); walk_group(slot.z, kb * 4 + 2);
                walk_group(slot.y, kb * 4 + 1); walk_group(slot.x, kb * 4 + 0);
                *(uint2 *)(tmp_top - 8 * (int64_t)(n_blk + 1)) = make_uint2(codes, (unsigned)cnt);   // every lane, same 8 bytes
                n_blk += 1; n_out += cnt;
                slot = load_block(kb - 4);
                kb -= 1;
                return (ck | ce) >= 0;
            };
            while (walk_block(ring0) && walk_block(ring1) && walk_block(ring2) && walk_block(ring3)) {}
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        ABEA_STAMP(t2b);
        // (2) 64 records at a time, a lane per record: where each block's walk started (a scan of the moves), its pairs
        // into the caller's array in reverse (std::reverse, :502), the emissions of its steps - and then their sum
        // in the order of the walk (:448-449; a double sum of floats, so the order is part of the result).  The
        // reference's array holds 2 x n_events pairs; a walk longer than that (only possible with far more k-mers than
        // events) is reported as failed instead of overrunning it.
        gbx_abea_pair *out = A.out + 2 * ev0;
        const bool fits = n_out <= cap;
        double sum_emission = 0;
        int base_e = ce0, base_k = ck0, base_i = 0;
        for (int g0 = 0; g0 < n_blk; g0 += 64) {
            const bool have = g0 + lane < n_blk;
            uint2 rec = make_uint2(0u, 0u);
            if (have) rec = *(const uint2 *)(tmp_top - 8 * (int64_t)(g0 + lane + 1));
            const unsigned cd = rec.x;
            const int n = (int)rec.y;
            const unsigned m = n >= 16 ? 0x55555555u : ((1u << (2 * n)) - 1u) & 0x55555555u;
            const int de = __builtin_popcount(m & ~(cd >> 1)), dk = __builtin_popcount(m & ~cd);   // moves 0,1 / 0,2
            int sn = n, se = de, sk = dk;                                                            // inclusive scans
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int tn = __shfl_up(sn, d), te = __shfl_up(se, d), tk = __shfl_up(sk, d);
                if (lane >= d) { sn += tn; se += te; sk += tk; }
            }
            int e = base_e - (se - de), k = base_k - (sk - dk);
            const int i0 = base_i + (sn - n);
            float lpe[16];
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                lpe[t] = 0.f;
                if (t < n) {
                    const float x = evm[e];
                    const float4 kq = kp[k];
                    const float a = (x - kq.x) / kq.y;                                        // lp_match, :143-147
                    lpe[t] = kq.z + (-0.5f * a * a);
                    if (fits) { gbx_abea_pair o; o.ref_pos = k; o.read_pos = e; out[n_out - 1 - (i0 + t)] = o; }
                    const unsigned code = (cd >> (2 * t)) & 3u;
                    e -= code != 2u; k -= code != 1u;
                }
            }
            const int nj = min(64, n_blk - g0);
            for (int j = 0; j < nj; ++j) {
#pragma unroll
                for (int t = 0; t < 16; ++t) sum_emission += (double)rlf(lpe[t], j);         // + 0.0f past a block's end
            }
            base_i += __builtin_amdgcn_readlane(sn, 63); base_e -= __builtin_amdgcn_readlane(se, 63); base_k -= __builtin_amdgcn_readlane(sk, 63);
        }
        const double avg_log_emission = sum_emission / (double)n_out;
        // spanned (:530-541): out.front() is the walk's last pair and must sit on k-mer 0, out.back() its first, which
        // sits on the last k-mer by construction
        bool ok = n_out >= 1 && fits && last_ck == 0;
        ok = ok && !(avg_log_emission < -5.0) && !(max_gap > 50);
        if (lane == 0) A.n_pairs[r] = ok ? n_out : 0;
        ABEA_STAMP(t3);
#ifdef GBX_ABEA_PHASE_STATS
        cyc_pre += t1 - t0; cyc_dp += t2 - t1; cyc_tb += t2b - t2; cyc_p2 += t3 - t2b; steps += (unsigned long long)n_out;
#endif
    }
    if (lane == 0) {
        atomicAdd(A.cells, fills);
#ifdef GBX_ABEA_PHASE_STATS
        atomicAdd(A.cells + 1, cyc_pre); atomicAdd(A.cells + 2, cyc_dp); atomicAdd(A.cells + 3, cyc_tb);   // s_memtime ticks, summed over wavefronts
        atomicAdd(A.cells + 4, cyc_p2); atomicAdd(A.cells + 5, steps);
#endif
    }
}

}  // namespace

// host-buffer entry: read r's pairs out + 2 * event_off[r] .. + n_pairs[r] -> packed + prefix[r] (a block per read)
__global__ void __launch_bounds__(256) abea_pack_kernel(int n_reads, const int64_t *__restrict__ event_off, const gbx_abea_pair *__restrict__ out,
                                                        const int32_t *__restrict__ n_pairs, const int64_t *__restrict__ prefix,
                                                        gbx_abea_pair *__restrict__ packed)
{
    for (int r = blockIdx.x; r < n_reads; r += gridDim.x) {
        const gbx_abea_pair *src = out + 2 * event_off[r];              // the same absolute indexing as abea_kernel
        gbx_abea_pair *dst = packed + prefix[r];
        for (int k = threadIdx.x; k < n_pairs[r]; k += 256) dst[k] = src[k];
    }
}

// workspace: [256 B header: cursor, cells] [16 B per k-mer] [trace rows]
static size_t abea_align_up(size_t x) { return (x + 255) & ~(size_t)255; }
size_t abea_workspace_bytes(int64_t n_reads, int64_t n_kmers_total, int64_t n_bands_total)
{
    const size_t nk = (size_t)(n_kmers_total > 0 ? n_kmers_total : 0) + 64;
    const size_t nb = (size_t)(n_bands_total > 0 ? n_bands_total : 0) + 32 * (size_t)(n_reads > 0 ? n_reads : 0) + 32;
    return 256 + abea_align_up(nk * 16) + abea_align_up(nb * ROW);
}

int abea_read_cells(const void *d_work, int64_t *cells, hipStream_t s)
{
    unsigned long long v = 0;
    GBX_HIP(hipMemcpyAsync(&v, (const char *)d_work + 8, sizeof(v), hipMemcpyDeviceToHost, s));
    GBX_HIP(hipStreamSynchronize(s));
    *cells = (int64_t)v;
    return GBX_OK;
}

int abea_pack_pairs(int64_t n_reads, const int64_t *d_event_off, const gbx_abea_pair *d_out, const int32_t *d_n_pairs,
                    const int64_t *d_prefix, void *d_work, int64_t n_kmers_total, gbx_abea_pair **d_packed, hipStream_t s)
{
    const size_t nk = (size_t)n_kmers_total + 64;
    gbx_abea_pair *packed = (gbx_abea_pair *)((char *)d_work + 256 + abea_align_up(nk * 16));      // the trace area (64 B per band >> 8 B per pair)
    *d_packed = packed;
    if (n_reads == 0) return GBX_OK;
    {
        Stage st("abea_pack", s);
        hipLaunchKernelGGL(abea_pack_kernel, dim3((unsigned)(n_reads < 16384 ? n_reads : 16384)), dim3(256), 0, s, (int)n_reads, d_event_off, d_out, d_n_pairs, d_prefix, packed);
    }
    GBX_HIP(hipGetLastError());
    return GBX_OK;
}

int abea_launch(int64_t n_reads, const int64_t *d_seq_off, const int32_t *d_seq_len, const char *d_seq,
                const int64_t *d_event_off, const float *d_event_mean, const gbx_abea_model *d_models,
                const float *d_scale, const float *d_shift, const int64_t *d_band_off, const int32_t *d_order,
                const double *d_lp, int64_t n_kmers_total, int64_t n_bands_total,
                gbx_abea_pair *d_out, int32_t *d_n_pairs, void *d_work, size_t work_bytes, hipStream_t s)
{
    if (n_reads == 0) return GBX_OK;
    if (n_reads > 0x7fffffffLL - 1024) { set_error("abea: more than 2^31 reads in one call"); return GBX_ERR_UNSUPPORTED; }
    if (work_bytes < abea_workspace_bytes(n_reads, n_kmers_total, n_bands_total)) { set_error("abea: workspace too small"); return GBX_ERR_ARG; }
    const size_t nk = (size_t)n_kmers_total + 64;
    char *w = (char *)d_work;
    AbeaArgs A;
    A.n_reads = (int)n_reads;
    A.seq_off = d_seq_off; A.seq_len = d_seq_len; A.seq = d_seq; A.event_off = d_event_off; A.event_mean = d_event_mean;
    A.models = d_models; A.scale = d_scale; A.shift = d_shift; A.band_off = d_band_off; A.order = d_order; A.lp = d_lp;
    A.lp_skip = log(1e-10); A.lp_trim = log(0.01);                            // align.c:201-205, the C library's log
    A.out = d_out; A.n_pairs = d_n_pairs;
    A.cursor = (unsigned *)w; A.cells = (unsigned long long *)(w + 8);
    size_t o = 256;
    A.kp = (float4 *)(w + o); o += abea_align_up(nk * 16);
    A.trace = (uint8_t *)(w + o);
    GBX_HIP(hipMemsetAsync(d_work, 0, 256, s));
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, abea_kernel, 64, 0) != hipSuccess || per_cu < 1) { (void)hipGetLastError(); per_cu = 8; }
    if (per_cu > 16) per_cu = 16;
    const int64_t slots = (int64_t)cus * per_cu;
    A.prio_cut = (unsigned)(cus * 4);
    const int grid = (int)(n_reads < slots ? n_reads : slots);
    {
        Stage st("abea_align", s);
        hipLaunchKernelGGL(abea_kernel, dim3(grid), dim3(64), 0, s, A);
    }
    GBX_HIP(hipGetLastError());
    GBX_GUARD_CHECK("abea");
    return GBX_OK;
}

}  // namespace gbx

// development aid (scripts/dbg_abea_phases.py; build with EXTRA=-DGBX_ABEA_PHASE_STATS, zeros otherwise): s_memtime ticks the wavefronts of the last launch spent in the per-k-mer
// prologue, the band loop, the traceback walk and its second pass, and the number of traceback steps
extern "C" int gbx_debug_abea_ticks(const void *d_work, unsigned long long *out5)
{
    return hipMemcpy(out5, (const char *)d_work + 16, 40, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}
