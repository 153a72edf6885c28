// bsw_kernels.hip — banded Smith-Waterman seed extension for gfx950 (MI355X).
//
// Semantics: BandedPairWiseSW::scalarBandedSWA, R/benchmarks/bsw/bandedSWA.cpp:128-249
// (bwa ksw_extend2), bit-exact on all six outputs.  The driver's entry point
// getScores16 (bandedSWA.cpp:1124-1148) is what gbx_bsw_extend_* replaces.
//
// Design: the DP is a row-sequential state machine (band clip,
// zero-row exit, z-drop on the LAST arg-max, window narrowing) so the row is
// the unit of parallelism.  A *group* of LPP lanes owns one pair; each lane
// keeps CPL consecutive columns of the reference's eh[] array (h and e planes)
// in registers for the whole extension.  Within a row
//     M(j), E'(j)          depend only on the previous row    -> element-wise
//     F(j+1)=max(F(j)-e, max(M(j)-oe,0))                      -> max-plus prefix scan
// The in-lane part of the scan is serial over CPL columns; the cross-lane part
// is 4 DPP row_shr steps (16-lane DPP rows are exactly one group for LPP=16,
// so four pairs run per wavefront with no cross-talk).  Row max / last arg-max,
// first / last live column and the global-score cell are DPP all-reduces.
// No LDS, no MFMA.  Long queries (qlen > 1024) use the one-pair-per-wavefront
// LDS-staged kernel at the bottom of this file.
//
// State invariants relied on (proved in DESIGN.md, checked by the adversarial
// tests): cells right of the live window are either never written (first-row
// initial values, E=0) or hold zeros; cells left of it are never read again.
#include "gbx_internal.h"
#include <atomic>
#include <cstdlib>

namespace gbx {
namespace {

constexpr int NEG = -(1 << 29);
constexpr int BIGJ = 1 << 20;

// qlen classes: 0..15 = qlen <= 8,16,..,128; 16..18 = <= 160,192,256; 19 = <= 1024; 20 = LDS kernel (shapes: class_shapes())
constexpr int NCLS = 21;
constexpr int CLS_LDS = NCLS - 1;
constexpr int NTB = 32;                     // target-length buckets per class (width 16), longest first
constexpr int NBIN = NCLS * NTB;            // 672 bins: pairs are binned by (class, descending target length)
constexpr int HDR = 1024;                   // ints per header array (>= NBIN + 1)
constexpr int REG_SCORE_LIMIT = 1 << 20;   // register kernels pack (h<<10|j) in 31 bits

struct BswDev {
    int o_del, e_del, o_ins, e_ins, oe_del, oe_ins, zdrop, end_bonus, w, max_mat;
    uint32_t colword[5];   // colword[q] = bytes mat[0][q], mat[1][q], mat[2][q], mat[3][q]
    int32_t col4[5];       // col4[q]    = mat[4][q]
    uint32_t row0[5];      // row0[t]    = bytes mat[t][0], mat[t][1], mat[t][2], mat[t][3]
    uint32_t row1[5];      // row1[t]    = byte  mat[t][4]
    uint8_t remap[24];     // query class -> the class whose kernel runs it (identity, or a wider class for small jobs)
    uint32_t lrow[5];      // lane path: lrow[t] = mat[t][q] as five signed 6-bit fields at bit 6q (read with one v_bfe_i32)
    int lane_on;           // 1: pairs that qualify (lane_ok) run on bsw_lane_kernel, one pair per lane
};

struct BswPairs {
    const uint8_t *ref, *qer;
    const int64_t *idr, *idq;
    const int32_t *len1, *len2, *h0;
    gbx_bsw_result *out;
    int packed = 0;        // 1: ref / qer are the packed images (two codes per byte), for launches of bsw_lane_kernel<.., PACKED> only
};

// workspace layout (ints): counts[HDR] | cursors[HDR] | base[HDR] (exclusive prefix of counts) | misc[HDR] | order[n] | wband[n]
//   misc[0] = bad-pair count; order[] = pair indices binned by (class, target length), wband[] = their clamped band width
constexpr int WS_HDR = 4 * HDR;
struct BswWork {
    int32_t *counts, *cursors, *base, *bad, *order, *wband;
    // lane path (bsw_lane_kernel): pairs sorted by (query length, seed score): lbase[] = exclusive prefix of the key
    // histogram (LANE_BINS + 1 entries), lrank[] = a pair's rank in its bin, lorder[] = sorted pair indices, lchunk[] = per-launch
    // chunk cursors
    int32_t *lbase, *lrank, *lorder, *lchunk;
};

__host__ __device__ inline int cls_of(int qlen, int bound)
{
    if (qlen > 1024 || bound >= REG_SCORE_LIMIT) return CLS_LDS;
    if (qlen > 256) return 19;
    if (qlen <= 128) return (qlen - 1) >> 3;          // 8 columns per class: a pair wastes at most 7
    return qlen <= 160 ? 16 : qlen <= 192 ? 17 : 18;
}
// ---- lane path: one pair per LANE ------------------------------------------------------------------------------
// Sort key of a pair = (qlen, min(h0, 255)): the live window of a row is a function of the row, the query length and
// the seed score far more than of the bases, so the 64 pairs of a wavefront taken from consecutive keys sweep nearly
// the same columns in every row (measured on 'large': 97 % of the lane-iterations do live work).
constexpr int LANE_QMAX = 159;                       // longest query on the lane path
constexpr int LANE_ROWS = 2 * (LANE_QMAX + 1);       // key rows: format (0 = compact cells, 1 = wide) x query length
constexpr int LANE_BINS = LANE_ROWS * 256;
constexpr int LANE_SCORE_LIMIT = 8192;               // wide cell word = h:14 | e:13 | query code x 6 : 5
constexpr int LANE_COMPACT_LIMIT = 256;              // compact cell = h:8 | e:8, query code in a byte plane
constexpr int LANE_NRANGE = 5;
// query-length ranges of the launches = LDS classes: compact 16, 10, 8, 6, 5 wavefronts per CU, wide 15, 7, 5, 4, 3.  (Eight
// compact ranges, one per step of the LDS occupancy between 16 and 5, measured no faster on 'large': 6.19 vs 6.03 ms.)
constexpr int LANE_RANGE_HI[2][LANE_NRANGE] = {{47, 79, 99, 135, LANE_QMAX}, {39, 79, 103, 127, LANE_QMAX}};
__host__ __device__ inline bool lane_ok(int lane_on, int qlen, int tlen, int h0, int max_mat)
{
    return lane_on && qlen >= 1 && qlen <= LANE_QMAX && tlen >= 1 && h0 >= 0 && h0 + qlen * (max_mat > 0 ? max_mat : 0) < LANE_SCORE_LIMIT;
}
__host__ __device__ inline int lane_key(int qlen, int h0, int max_mat)
{
    const int wide = h0 + qlen * (max_mat > 0 ? max_mat : 0) < LANE_COMPACT_LIMIT ? 0 : 1;
    return ((wide * (LANE_QMAX + 1) + qlen) << 8) | (h0 < 255 ? h0 : 255);
}

__host__ __device__ inline int bin_of(int cls, int tlen)
{
    const int tb = tlen >> 4;
    return cls * NTB + (NTB - 1 - (tb < NTB - 1 ? tb : NTB - 1));
}

__device__ inline int imax3(int a, int b, int c) { return max(max(a, b), c); }

template <int CTRL, int ROWMASK = 0xf, int BANKMASK = 0xf>
__device__ inline int dpp(int old, int x)
{
    return __builtin_amdgcn_update_dpp(old, x, CTRL, ROWMASK, BANKMASK, false);
}

// ---- group primitives -----------------------------------------------------
// A group is LPP consecutive lanes (2, 4, 8, 16 or 64) that own one pair.
// mov_dpp with an undefined `old` and bound_ctrl:1 (out-of-row lanes read 0) lets the compiler fold the
// move into the consumer (v_max_i32_dpp): one VALU op per scan / reduce step.  max is idempotent, so a
// lane that reads itself (quad_perm patterns below) is harmless.
template <int CTRL, int ROWMASK = 0xf>
__device__ inline int dppz(int x)
{
    return __builtin_amdgcn_mov_dpp(x, CTRL, ROWMASK, 0xf, true);
}

// inclusive prefix max over the LPP lanes of a group, for values >= 0 (0 is the fill)
template <int LPP>
__device__ inline int group_scan_max(int x)
{
    if (LPP == 2) return max(x, dppz<0xa0>(x));           // quad_perm [0,0,2,2]
    if (LPP <= 8) {
        x = max(x, dppz<0x90>(x));                        // quad_perm [0,0,1,2]
        x = max(x, dppz<0x40>(x));                        // quad_perm [0,0,0,1]: prefix inside each quad
        if (LPP == 8) {
            const int tot = dppz<0xff>(x);                // quad total in every lane of the quad
            x = max(x, dpp<0x114, 0xf, 0xa>(0, tot));     // row_shr:4 into the odd quads only
        }
        return x;
    }
    x = max(x, dppz<0x111>(x));
    x = max(x, dppz<0x112>(x));
    x = max(x, dppz<0x114>(x));
    x = max(x, dppz<0x118>(x));
    if (LPP >= 32) x = max(x, dpp<0x142, 0xa>(NEG, x));   // row_bcast15 -> rows 1,3
    if (LPP >= 64) x = max(x, dpp<0x143, 0xc>(NEG, x));   // row_bcast31 -> rows 2,3
    return x;
}

// value of the previous lane of the group; lane 0 of the group gets `fill`
template <int LPP>
__device__ inline int group_shift_up(int x, int fill, int gl)
{
    if (LPP == 16) return dpp<0x111>(fill, x);            // row_shr:1, old = fill
    if (LPP < 16) { const int y = dpp<0x111>(fill, x); return gl == 0 ? fill : y; }
    int y = dpp<0x138>(fill, x);                          // wave_shr:1
    return gl == 0 ? fill : y;
}

// all-reduce max over the group; every lane gets the result
template <int LPP>
__device__ inline int group_allmax(int x)
{
    if (LPP == 2) return max(x, dppz<0xb1>(x));           // quad_perm [1,0,3,2]
    if (LPP <= 8) {
        x = max(x, dppz<0xb1>(x));                        // quad_perm [1,0,3,2]
        x = max(x, dppz<0x4e>(x));                        // quad_perm [2,3,0,1]
        if (LPP == 8) x = max(x, dppz<0x141>(x));         // row_half_mirror
        return x;
    }
    if (LPP == 16) {
        x = max(x, dppz<0x121>(x));                       // row_ror:1,2,4,8 (every lane has a source)
        x = max(x, dppz<0x122>(x));
        x = max(x, dppz<0x124>(x));
        x = max(x, dppz<0x128>(x));
        return x;
    }
    x = max(x, dpp<0x111>(x, x));
    x = max(x, dpp<0x112>(x, x));
    x = max(x, dpp<0x114>(x, x));
    x = max(x, dpp<0x118>(x, x));
    x = max(x, dpp<0x142, 0xa>(x, x));
    x = max(x, dpp<0x143, 0xc>(x, x));
    return __builtin_amdgcn_readlane(x, 63);
}

// band clamp of scalarBandedSWA (:159-168) in integers: (int)((double)n / e + 1.) clipped below at 1
__device__ inline int band_width(const BswDev &prm, int qlen)
{
    const int n_ins = qlen * prm.max_mat + prm.end_bonus - prm.o_ins;
    const int n_del = qlen * prm.max_mat + prm.end_bonus - prm.o_del;
    const int l_ins = n_ins >= 0 ? n_ins / prm.e_ins + 1 : 1;
    const int l_del = n_del >= 0 ? n_del / prm.e_del + 1 : 1;
    return min(prm.w, min(l_ins, l_del));
}

// ---- classify / bin pairs by (query-length class, descending target length) ----------------
// Degenerate pairs (len 0) are answered here; bad pairs (negative or too long) get -1 outputs
// and bump work.bad.  pass 0: histogram; bsw_scan_kernel; pass 1: scatter into order[].
// Handing pairs out longest-first makes the static round-robin over groups an LPT schedule.
constexpr int CLS_THREADS = 1024;
__global__ void __launch_bounds__(CLS_THREADS) bsw_classify_kernel(BswDev prm, BswPairs P, int64_t n, BswWork W, int pass)
{
    __shared__ int lcount[NBIN];
    __shared__ int lbase[NBIN];
    const int tid = threadIdx.x;
    for (int b = tid; b < NBIN; b += CLS_THREADS) lcount[b] = 0;
    __syncthreads();
    const int64_t k = (int64_t)blockIdx.x * CLS_THREADS + tid;
    int bin = -1, slot = 0;
    if (k < n) {
        const int qlen = P.len2[k], tlen = P.len1[k], h0 = P.h0[k];
        if (qlen < 0 || tlen < 0 || qlen > GBX_BSW_MAX_QLEN || tlen > GBX_BSW_MAX_TLEN) {
            if (pass == 0) {
                gbx_bsw_result r = {-1, -1, -1, -1, -1, -1};
                P.out[k] = r;
                atomicAdd(W.bad, 1);
            }
        } else if (tlen == 0 || qlen == 0) {
            if (pass == 0) {
                // scalarBandedSWA with an empty matrix: no row at all (tlen==0), or one row
                // with an empty window that only updates gscore (qlen==0), bandedSWA.cpp:174-218
                gbx_bsw_result r;
                r.score = h0; r.tle = 0; r.qle = 0; r.max_off = 0;
                if (tlen == 0) { r.gtle = 0; r.gscore = -1; }
                else { int left = max(h0 - (prm.o_del + prm.e_del), 0); r.gtle = 1; r.gscore = max(-1, left); }
                P.out[k] = r;
            }
        } else if (lane_ok(prm.lane_on, qlen, tlen, h0, prm.max_mat)) {
            // bsw_lane_kernel's pair (sorted by bsw_lane_sort_kernel)
        } else {
            const int bound = max(h0, 0) + qlen * max(prm.max_mat, 0);
            bin = bin_of(prm.remap[cls_of(qlen, bound)], tlen);
            slot = atomicAdd(&lcount[bin], 1);
        }
    }
    __syncthreads();
    if (pass == 0) {
        for (int b = tid; b < NBIN; b += CLS_THREADS)
            if (lcount[b]) atomicAdd(&W.counts[b], lcount[b]);
        return;
    }
    for (int b = tid; b < NBIN; b += CLS_THREADS)
        lbase[b] = lcount[b] ? W.base[b] + atomicAdd(&W.cursors[b], lcount[b]) : 0;
    __syncthreads();
    if (bin >= 0) {
        W.order[lbase[bin] + slot] = (int)k;
        W.wband[lbase[bin] + slot] = band_width(prm, P.len2[k]);
    }
}

// exclusive prefix of the bin counts (one block)
__global__ void __launch_bounds__(HDR) bsw_scan_kernel(BswWork W)
{
    __shared__ int tmp[HDR];
    const int tid = threadIdx.x;
    const int v = tid < NBIN ? W.counts[tid] : 0;
    tmp[tid] = v;
    __syncthreads();
    for (int d = 1; d < HDR; d <<= 1) {
        const int add = tid >= d ? tmp[tid - d] : 0;
        __syncthreads();
        tmp[tid] += add;
        __syncthreads();
    }
    if (tid <= NBIN) W.base[tid] = tmp[tid] - v;       // base[NBIN] = total
}

// ---- register-resident row kernel ------------------------------------------
// SYM: o_ins+e_ins == o_del+e_del, the gap-open term of E and F is shared.
// the register budget is pinned per CPL (second launch-bound = wavefronts per SIMD) so that the wide
// shapes keep 4 wavefronts per SIMD resident
constexpr int rows_min_waves(int cpl) { return cpl >= 17 ? 3 : cpl >= 12 ? 4 : cpl >= 9 ? 5 : cpl >= 7 ? 6 : cpl >= 5 ? 7 : 1; }

template <int LPP, int CPL, bool SYM>
__global__ void __launch_bounds__(256, rows_min_waves(CPL)) bsw_rows_kernel(BswDev prm, BswPairs P, BswWork W, int cls)
{
    constexpr int KB = 10;                       // bits for the column index in the row key
    constexpr int GROUPS_PER_BLOCK = 256 / LPP;
    constexpr int NQ = (CPL + 3) / 4;            // query codes, one byte per column: selectors of v_perm_b32
    static_assert(LPP * CPL <= (1 << KB), "column index must fit the key");
    static_assert(CPL <= 32, "in-lane column index uses 5 bits");

    // scoring-matrix row of the current target base: 5 signed bytes {row0 = q 0..3, row1 = q 4}
    __shared__ uint2 s_row[8];
    const int tid = threadIdx.x;
    if (tid < 8) s_row[tid] = make_uint2(prm.row0[min(tid, 4)], prm.row1[min(tid, 4)]);
    __syncthreads();

    const int gl = tid & (LPP - 1);
    const int j0 = gl * CPL;
    const int ngroups = gridDim.x * GROUPS_PER_BLOCK;
    int next = blockIdx.x * GROUPS_PER_BLOCK + tid / LPP;

    // cls < 0: direct mode (bsw_launch_direct) - the job's pairs 0..-cls-1 in input order, no binning pass
    const bool direct = cls < 0;
    const int first = direct ? 0 : W.base[cls * NTB];
    const int cnt = direct ? -cls : W.base[(cls + 1) * NTB] - first;
    const int32_t *order = W.order + first;
    const int32_t *wband = W.wband + first;

    const int e_ins = prm.e_ins, e_del = prm.e_del, oe_ins = prm.oe_ins, oe_del = prm.oe_del;
    const int lane_tilt = gl * (CPL * e_ins);

    int Hs[CPL + 1], Ev[CPL];
    uint32_t Qp[NQ];
#pragma unroll
    for (int c = 0; c < CPL; ++c) { Hs[c] = 0; Ev[c] = 0; }
#pragma unroll
    for (int c = 0; c < NQ; ++c) Qp[c] = 0;

    // group-uniform state (replicated in every lane of the group); g_i / g_score are only
    // meaningful in lane lq, the lane that owns column qlen-1, which also writes the result
    int qlen = 1, tlen = 1, w = 0, beg = 0, end = 0, i = 0, pair = 0, lq = 0;
    int best = 0, best_i = -1, best_j = -1, g_i = -1, g_score = -1, off = 0;
    int leftv = 0;                               // h0 - (o_del + e_del*(i+1)), :183-186
    const uint8_t *tptr = P.ref;
    uint2 rcur = s_row[0];                       // matrix row of target base i
    int tnext = 0;                               // target base i+1
    bool active = false, done = false;

    for (;;) {
        if (done) {                                                   // retire
            if (gl == lq) {
                gbx_bsw_result r;
                r.score = best; r.tle = best_i + 1; r.gtle = g_i + 1; r.qle = best_j + 1;
                r.gscore = g_score; r.max_off = off;
                P.out[pair] = r;
            }
            done = false; active = false;
            tptr = P.ref; tlen = 1; i = 0;
        }
        if (!active && next < cnt) {                                  // fetch + first row (:155-168)
            pair = direct ? next : order[next];
            w = direct ? band_width(prm, P.len2[pair]) : wband[next];
            next += ngroups;
            qlen = P.len2[pair]; tlen = P.len1[pair];
            const int h0 = P.h0[pair];
            const uint8_t *q = P.qer + P.idq[pair];
            tptr = P.ref + P.idr[pair];
#pragma unroll
            for (int c = 0; c < NQ; ++c) Qp[c] = 0;
#pragma unroll
            for (int c = 0; c < CPL; ++c) {
                const int j = j0 + c;
                int qc = j < qlen ? q[j] : 0;
                qc = min(qc, 4);
                Qp[c >> 2] |= (uint32_t)qc << ((c & 3) * 8);
                Hs[c] = j == 0 ? h0 : max(h0 - oe_ins - (j - 1) * e_ins, 0);
                Ev[c] = 0;
            }
            lq = (qlen - 1) / CPL;
            leftv = h0 - prm.o_del - e_del;
            best = h0; best_i = -1; best_j = -1; g_i = -1; g_score = -1; off = 0;
            beg = 0; end = qlen; i = 0;
            rcur = s_row[min((int)tptr[0], 4)];
            tnext = tptr[min(1, tlen - 1)];
            active = true;
        }
        if (!__any(active)) break;

        // ------------------------------------------------------------ one row
        const int b = max(beg, i - w);                                 // :179-181
        const int e = min(min(end, i + w + 1), qlen);
        const int left0 = b == 0 ? max(leftv, 0) : 0;                  // :183-186
        leftv -= e_del;
        const int lo = b - j0, hi = e - j0;
        // scores of this row: one byte permute per 4 columns picks mat[t][q] by the query code
        uint32_t sw[NQ];
#pragma unroll
        for (int k = 0; k < NQ; ++k) sw[k] = __builtin_amdgcn_perm(rcur.y, rcur.x, Qp[k]);
        rcur = s_row[min(tnext, 4)];
        tnext = tptr[min(i + 2, tlen - 1)];

        int G[CPL];
        // previous-row terms.  G = max(M, E, in-lane part of F); the cross-lane F is folded in below.
        int lf = NEG;
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            const bool p = (c >= lo) & (c < hi);
            const int hsv = Hs[c];
            const int sc = (int)(int8_t)(sw[c >> 2] >> ((c & 3) * 8));
            const int m = (p & (hsv != 0)) ? hsv + sc : 0;             // :196
            const int ein = p ? Ev[c] : 0;
            G[c] = imax3(m, ein, lf);                                  // :197-198 without the carried F
            const int mo = m - oe_ins;
            lf = imax3(lf - e_ins, mo, 0);                             // F(i,j+1), :207-210
            Ev[c] = imax3(ein - e_del, SYM ? mo : m - oe_del, 0);      // E(i+1,j), :202-206
        }
        // cross-lane part of the F scan: carry-out of lane l, tilted by l*CPL*e_ins
        const int carry = group_scan_max<LPP>(lf + lane_tilt) - lane_tilt;
        int X = group_shift_up<LPP>(carry, NEG, gl);                   // F entering this lane's column 0

        // h, the row key (last arg-max) and the shifted store eh[j+1].h <- H(i,j) for j < e
        int kl = 0, hl = 0;
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            int h = max(G[c], X);
            X -= e_ins;
            const bool pr = c < hi;
            Hs[c + 1] = pr ? h : Hs[c + 1];
            hl = pr ? h : hl;                                          // h of the last live column in this lane
            h = pr ? h : 0;
            G[c] = h;
            kl = max(kl, (h << 5) | c);
        }
        // column 0 takes the previous lane's last h; the pair's column 0 takes the first-column value
        const int hin = group_shift_up<LPP>(G[CPL - 1], left0, gl);
        Hs[0] = hi >= 0 ? hin : Hs[0];
        unsigned zm = 0;
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            const int hs = c == 0 ? hin : G[c - 1];
            unsigned nz;                                               // min(x,1); asm keeps it one op (no cmp+select)
            asm("v_min_u32_e32 %0, 1, %1" : "=v"(nz) : "v"(hs | Ev[c]));
            zm |= nz << c;
        }
        Hs[CPL] = 0;                                                   // column (gl+1)*CPL lives in the next lane

        // group reductions
        const int key = group_allmax<LPP>(((kl >> 5) << KB) | (j0 + (kl & 31)));
        const int m = key >> KB, mj = key & ((1 << KB) - 1);
        const int zf_l = zm ? j0 + (__builtin_ffs((int)zm) - 1) : BIGJ;
        const int zl_l = zm ? j0 + (31 - __builtin_clz(zm)) : -1;
        const int zfirst = BIGJ - group_allmax<LPP>(BIGJ - zf_l);
        const int zlast = group_allmax<LPP>(zl_l);

        // row epilogue, :213-233, branch-free
        const bool live = b < e;
        const int jfin = live ? e : b;
        const int hend = live ? hl : left0;                            // = H(i,e-1) in lane lq when e == qlen
        const bool gq = jfin == qlen;                                  // :214-217
        g_i = (gq & (hend >= g_score)) ? i : g_i;
        g_score = gq ? max(g_score, hend) : g_score;
        const bool upd = (m > best) & (m != 0);                        // :218-221
        const int dd = (i - best_i) - (mj - best_j);                   // :222-228
        const int pen = max(__mul24(dd, e_del), __mul24(-dd, e_ins));
        const bool zbrk = (prm.zdrop > 0) & !upd & (best - m - pen > prm.zdrop);
        const bool brk = (m == 0) | zbrk;
        const int dmi = mj - i;
        off = upd ? imax3(off, dmi, -dmi) : off;
        best_i = upd ? i : best_i;
        best_j = upd ? mj : best_j;
        best = upd ? m : best;
        beg = min(zfirst, e);                                          // :230-233
        end = min(zlast + 2, qlen);
        ++i;
        done = active & (brk | (i >= tlen));
    }
}

// ---- one pair per wavefront, row staged in LDS (long queries) ---------------
// hd[] / ev[] planes of eh[] live in LDS (int32), columns are swept in 64-wide
// chunks with the F carry and the running reductions handed from chunk to chunk.
__global__ void __launch_bounds__(64) bsw_lds_kernel(BswDev prm, BswPairs P, BswWork W, int cls)
{
    extern __shared__ int lds[];
    const int lane = threadIdx.x;
    const int first = W.base[cls * NTB];
    const int cnt = W.base[(cls + 1) * NTB] - first;
    const int32_t *order = W.order + first;
    const int e_ins = prm.e_ins, e_del = prm.e_del, oe_ins = prm.oe_ins, oe_del = prm.oe_del;

    for (int slot = blockIdx.x; slot < cnt; slot += gridDim.x) {
        const int pair = order[slot];
        const int qlen = P.len2[pair], tlen = P.len1[pair], h0 = P.h0[pair];
        const uint8_t *q = P.qer + P.idq[pair];
        const uint8_t *tp = P.ref + P.idr[pair];
        int *hd = lds, *ev = lds + (qlen + 1);
        for (int j = lane; j <= qlen; j += 64) {
            hd[j] = j == 0 ? h0 : max(h0 - oe_ins - (j - 1) * e_ins, 0);
            ev[j] = 0;
        }
        __syncthreads();
        int w;
        {
            int lim = (int)((double)(qlen * prm.max_mat + prm.end_bonus - prm.o_ins) / (double)e_ins + 1.);
            lim = max(lim, 1); w = min(prm.w, lim);
            lim = (int)((double)(qlen * prm.max_mat + prm.end_bonus - prm.o_del) / (double)e_del + 1.);
            lim = max(lim, 1); w = min(w, lim);
        }
        int best = h0, best_i = -1, best_j = -1, g_i = -1, g_score = -1, off = 0;
        int beg = 0, end = qlen;
        for (int i = 0; i < tlen; ++i) {
            const int b = max(beg, i - w);
            const int e = min(min(end, i + w + 1), qlen);
            const int left0 = b == 0 ? max(h0 - (prm.o_del + e_del * (i + 1)), 0) : 0;
            const int t = min((int)tp[i], 4);
            int fcarry = NEG;           // F entering the chunk's first column
            int left = left0;           // H(i, j-1) entering the chunk's first column
            int m = 0, mj = -1;
            int zfirst = BIGJ, zlast = -1;
            if (left0 != 0) { zfirst = b; zlast = b; }       // eh[beg].h = left0
            for (int cb = b; cb < e; cb += 64) {
                const int j = cb + lane;
                const bool p = j < e;
                const int qc = p ? min((int)q[j], 4) : 0;
                const int s = t == 4 ? prm.col4[qc] : (int)(int8_t)(prm.colword[qc] >> (t << 3));
                const int hsv = p ? hd[j] : 0;
                const int ein = p ? ev[j] : 0;
                const int mm = (p & (hsv != 0)) ? hsv + s : 0;
                // F via tilted prefix max over the chunk
                const int src = p ? max(mm - oe_ins, 0) : NEG;          // contributes to F at j+1
                int pre = group_scan_max<64>(src + lane * e_ins);       // inclusive, tilted
                int fincl = pre - lane * e_ins;                         // F(j+1) from in-chunk sources
                int fexcl = group_shift_up<64>(fincl, NEG, lane);       // F(j) from in-chunk sources
                fexcl = lane == 0 ? NEG : fexcl - 0;                    // (shift fills NEG)
                // decay the in-chunk value by one column: fincl is F at j+1, shift gives F at j for this lane
                const int f = max(fexcl, fcarry - lane * e_ins);
                int h = imax3(mm, ein, f);
                h = p ? h : 0;
                const int enew = imax3(ein - e_del, mm - oe_del, 0);
                // carry to the next chunk: F at column cb+64
                {
                    const int tot = __builtin_amdgcn_readlane(fincl, 63);
                    fcarry = max(tot, fcarry - 64 * e_ins);
                }
                // shifted h store
                int hprev = group_shift_up<64>(h, left, lane);
                hprev = lane == 0 ? left : hprev;
                left = __builtin_amdgcn_readlane(h, 63);                // may be masked 0 beyond e; fixed below
                if (p) { hd[j] = hprev; ev[j] = enew; }
                // reductions
                {
                    const int hm = group_allmax<64>(h);
                    if (hm >= m && (hm > 0 || m == 0)) {
                        // last column with h == hm inside this chunk
                        const unsigned long long eq = __ballot(p && h == hm);
                        if (eq) { mj = cb + 63 - __builtin_clzll(eq); m = hm; }
                    }
                    const unsigned long long nzh = __ballot(p && h != 0);      // hd[j+1] != 0
                    const unsigned long long nze = __ballot(p && enew != 0);   // ev[j] != 0
                    if (nzh) {
                        zfirst = min(zfirst, cb + (int)__builtin_ctzll(nzh) + 1);
                        zlast = max(zlast, cb + 63 - (int)__builtin_clzll(nzh) + 1);
                    }
                    if (nze) {
                        zfirst = min(zfirst, cb + (int)__builtin_ctzll(nze));
                        zlast = max(zlast, cb + 63 - (int)__builtin_clzll(nze));
                    }
                }
                // h of the last live column (column e-1) for eh[end] / gscore
                if (cb + 64 >= e) left = __builtin_amdgcn_readlane(h, (e - 1 - cb) & 63);
            }
            // eh[end] = {left, 0}
            if (lane == 0) { hd[e] = b < e ? left : left0; ev[e] = 0; }
            __syncthreads();
            const int hend = b < e ? left : left0;
            const int jfin = b < e ? e : b;
            if (jfin == qlen) { g_i = g_score > hend ? g_i : i; g_score = max(g_score, hend); }
            if (m == 0) break;
            if (m > best) {
                best = m; best_i = i; best_j = mj; off = max(off, abs(mj - i));
            } else if (prm.zdrop > 0) {
                const int di = i - best_i, dj = mj - best_j;
                const int pen = di > dj ? (di - dj) * e_del : (dj - di) * e_ins;
                if (best - m - pen > prm.zdrop) break;
            }
            beg = min(zfirst, e);
            end = min(zlast + 2, qlen);
        }
        if (lane == 0) {
            gbx_bsw_result r;
            r.score = best; r.tle = best_i + 1; r.gtle = g_i + 1; r.qle = best_j + 1;
            r.gscore = g_score; r.max_off = off;
            P.out[pair] = r;
        }
        __syncthreads();
    }
}

// ---- one pair per lane --------------------------------------------------------------------------------------
// The row kernels above give a pair 2-64 lanes and pay for it twice on the 151-bp workload: 45 % of the columns they
// compute lie outside the live window (a lane owns fixed columns), and a third of their instructions are window masks,
// scans and reductions.  Measured instruction rates (profiles/valu_peak.json) say the rest cannot be bought back by
// recoding: max / max3 / DPP / SDWA / compare / select all issue at one wave64 instruction per 4 cycles.
// Here a LANE owns a pair and walks its own window [beg, end) of every row, exactly as scalarBandedSWA
// (bandedSWA.cpp:128-249) does: no dead columns, no masks, no cross-lane traffic at all.  The eh[] array of the 64
// pairs of a wavefront lives in LDS, one 32-bit word per cell (h:14 | e:13 | query code x 6 : 5), column-major
// ([column][lane]: every access is bank-conflict free whatever column each lane is at).  The wavefront runs the rows
// in lock-step (a row lasts as long as its widest window), which costs nothing when the lanes' windows agree: the
// pairs are sorted by (query length, seed score) first (bsw_lane_sort_kernel).
__global__ void __launch_bounds__(256) bsw_lane_sort_kernel(BswDev prm, BswPairs P, int64_t n, BswWork W, int pass)
{
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    const int qlen = P.len2[k], tlen = P.len1[k], h0 = P.h0[k];
    if (!lane_ok(prm.lane_on, qlen, tlen, h0, prm.max_mat) || tlen > GBX_BSW_MAX_TLEN) return;
    const int key = lane_key(qlen, h0, prm.max_mat);
    // one pass of atomics: the count's old value is the pair's rank in its bin, kept for the placement (2 M scattered atomics
    // cost 0.13 ms whether they return a value or not: the second pass used to pay that again for its cursors)
    if (pass == 0) W.lrank[k] = atomicAdd(&W.lbase[key + 1], 1);
    else W.lorder[W.lbase[key] + W.lrank[k]] = (int)k;
}
// exclusive prefix of the key histogram in place: lbase[k + 1] holds count(k) on entry, lbase[k] = pairs with a key below k
// on exit (one block)
__global__ void __launch_bounds__(1024) bsw_lane_scan_kernel(BswWork W)
{
    constexpr int PER = (LANE_BINS + 1023) / 1024;
    __shared__ int part[1024];
    const int tid = threadIdx.x;
    int sum = 0;
    for (int k = 0; k < PER; ++k) { const int b = tid * PER + k; if (b < LANE_BINS) sum += W.lbase[b + 1]; }
    part[tid] = sum;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const int add = tid >= d ? part[tid - d] : 0;
        __syncthreads();
        part[tid] += add;
        __syncthreads();
    }
    int run = part[tid] - sum;
    for (int k = 0; k < PER; ++k) {
        const int b = tid * PER + k;
        if (b < LANE_BINS) { const int c = W.lbase[b + 1]; run += c; W.lbase[b + 1] = run; }
    }
}

// Two columns of the compact lane kernel, scheduled by hand (28 VALU instructions; the compiler's version of the same
// C++ needs 43).  w = the lane's dword of the column pair, bytes {e(2p), h(2p), e(2p+1), h(2p+1)}; qq = the pair's query
// codes (0..4) in bytes 0 and 1, zero above: they are the selector of one v_perm_b32 over the matrix row of the target
// base (rw = its four bytes against A C G T, rwn = the byte against N), which leaves the two scores in bytes 0 and 1;
// scores and cell fields are byte operands (SDWA), never unpacked.  Returns the new dword; updates f, left (the previous
// column's h), key = max(kin, the two columns' (h << 18 | pa), (h << 18 | pa1)): pa, pa1 are absolute cell addresses, or
// offsets within a trip of the main loop whose base is added once per trip.  vcc is written two instructions before it
// is read.
template <bool SYM>
__device__ __forceinline__ uint32_t lane_pair_step(uint32_t w, uint32_t qq, uint32_t rw, uint32_t rwn, int &f, int &left, uint32_t kin, uint32_t &key, int pa, int pa1,
                                                   int zero, int oe_del, int oe_ins, int e_del, int e_ins)
{
    uint32_t wn, kout;
    int hb;
    if (SYM) {
        int sa, ta, tb, ma, mb, x, ha, td, ed, ena, enb, fd, ka, kb, u, v;
        asm volatile(
            "v_perm_b32 %[sa], %[rwn], %[rw], %[qq]\n"
            "v_cmp_ne_u32_sdwa vcc, %[w], %[zero] src0_sel:BYTE_1 src1_sel:DWORD\n"
            "v_subrev_u32 %[fd], %[eins], %[f]\n"
            "v_add_u32_sdwa %[ta], sext(%[sa]), %[w] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1\n"
            "v_cndmask_b32 %[ma], 0, %[ta], vcc\n"
            "v_cmp_ne_u32_sdwa vcc, %[w], %[zero] src0_sel:BYTE_3 src1_sel:DWORD\n"
            "v_max_i32_sdwa %[x], %[ma], %[w] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n"
            "v_add_u32_sdwa %[tb], sext(%[sa]), %[w] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_3\n"
            "v_max_i32 %[ha], %[x], %[f]\n"
            "v_subrev_u32 %[td], %[oed], %[ma]\n"
            "v_cndmask_b32 %[mb], 0, %[tb], vcc\n"
            "v_sub_u32_sdwa %[ed], %[w], %[edel] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD\n"
            "v_max3_i32 %[f], %[fd], %[td], 0\n"
            "v_max3_i32 %[ena], %[ed], %[td], 0\n"
            "v_max_i32_sdwa %[x], %[mb], %[w] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n"
            "v_lshl_or_b32 %[ka], %[ha], 18, %[pa]\n"
            "v_max_i32 %[hb], %[x], %[f]\n"
            "v_subrev_u32 %[td], %[oed], %[mb]\n"
            "v_subrev_u32 %[fd], %[eins], %[f]\n"
            "v_sub_u32_sdwa %[ed], %[w], %[edel] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD\n"
            "v_max3_i32 %[f], %[fd], %[td], 0\n"
            "v_max3_i32 %[enb], %[ed], %[td], 0\n"
            "v_lshl_or_b32 %[kb], %[hb], 18, %[pa1]\n"
            "v_lshl_or_b32 %[u], %[left], 8, %[ena]\n"
            "v_lshl_or_b32 %[v], %[ha], 8, %[enb]\n"
            "v_max3_u32 %[key], %[kin], %[ka], %[kb]\n"
            "v_lshl_or_b32 %[wn], %[v], 16, %[u]\n"
            : [sa] "=&v"(sa), [ta] "=&v"(ta), [tb] "=&v"(tb), [ma] "=&v"(ma), [mb] "=&v"(mb), [x] "=&v"(x), [ha] "=&v"(ha),
              [hb] "=&v"(hb), [td] "=&v"(td), [ed] "=&v"(ed), [ena] "=&v"(ena), [enb] "=&v"(enb), [fd] "=&v"(fd), [ka] "=&v"(ka),
              [kb] "=&v"(kb), [u] "=&v"(u), [v] "=&v"(v), [wn] "=&v"(wn), [f] "+v"(f), [key] "=v"(kout)
            : [w] "v"(w), [qq] "v"(qq), [rw] "v"(rw), [rwn] "v"(rwn), [left] "v"(left), [kin] "v"(kin), [pa] "v"(pa), [pa1] "v"(pa1), [zero] "v"(zero), [oed] "s"(oe_del),
              [edel] "s"(e_del), [eins] "s"(e_ins)
            : "vcc");
    } else {
        int sa, ta, tb, ma, mb, x, ha, td, ti, ed, ena, enb, fd, ka, kb, u, v;
        asm volatile(
            "v_perm_b32 %[sa], %[rwn], %[rw], %[qq]\n"
            "v_cmp_ne_u32_sdwa vcc, %[w], %[zero] src0_sel:BYTE_1 src1_sel:DWORD\n"
            "v_subrev_u32 %[fd], %[eins], %[f]\n"
            "v_add_u32_sdwa %[ta], sext(%[sa]), %[w] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1\n"
            "v_cndmask_b32 %[ma], 0, %[ta], vcc\n"
            "v_cmp_ne_u32_sdwa vcc, %[w], %[zero] src0_sel:BYTE_3 src1_sel:DWORD\n"
            "v_max_i32_sdwa %[x], %[ma], %[w] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n"
            "v_add_u32_sdwa %[tb], sext(%[sa]), %[w] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_3\n"
            "v_max_i32 %[ha], %[x], %[f]\n"
            "v_subrev_u32 %[td], %[oed], %[ma]\n"
            "v_subrev_u32 %[ti], %[oei], %[ma]\n"
            "v_cndmask_b32 %[mb], 0, %[tb], vcc\n"
            "v_sub_u32_sdwa %[ed], %[w], %[edel] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD\n"
            "v_max3_i32 %[f], %[fd], %[ti], 0\n"
            "v_max3_i32 %[ena], %[ed], %[td], 0\n"
            "v_max_i32_sdwa %[x], %[mb], %[w] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n"
            "v_lshl_or_b32 %[ka], %[ha], 18, %[pa]\n"
            "v_max_i32 %[hb], %[x], %[f]\n"
            "v_subrev_u32 %[td], %[oed], %[mb]\n"
            "v_subrev_u32 %[ti], %[oei], %[mb]\n"
            "v_subrev_u32 %[fd], %[eins], %[f]\n"
            "v_sub_u32_sdwa %[ed], %[w], %[edel] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD\n"
            "v_max3_i32 %[f], %[fd], %[ti], 0\n"
            "v_max3_i32 %[enb], %[ed], %[td], 0\n"
            "v_lshl_or_b32 %[kb], %[hb], 18, %[pa1]\n"
            "v_lshl_or_b32 %[u], %[left], 8, %[ena]\n"
            "v_lshl_or_b32 %[v], %[ha], 8, %[enb]\n"
            "v_max3_u32 %[key], %[kin], %[ka], %[kb]\n"
            "v_lshl_or_b32 %[wn], %[v], 16, %[u]\n"
            : [sa] "=&v"(sa), [ta] "=&v"(ta), [tb] "=&v"(tb), [ma] "=&v"(ma), [mb] "=&v"(mb), [x] "=&v"(x), [ha] "=&v"(ha),
              [hb] "=&v"(hb), [td] "=&v"(td), [ti] "=&v"(ti), [ed] "=&v"(ed), [ena] "=&v"(ena), [enb] "=&v"(enb), [fd] "=&v"(fd), [ka] "=&v"(ka),
              [kb] "=&v"(kb), [u] "=&v"(u), [v] "=&v"(v), [wn] "=&v"(wn), [f] "+v"(f), [key] "=v"(kout)
            : [w] "v"(w), [qq] "v"(qq), [rw] "v"(rw), [rwn] "v"(rwn), [left] "v"(left), [kin] "v"(kin), [pa] "v"(pa), [pa1] "v"(pa1), [zero] "v"(zero), [oed] "s"(oe_del),
              [oei] "s"(oe_ins), [edel] "s"(e_del), [eins] "s"(e_ins)
            : "vcc");
    }
    left = hb;
    key = kout;
    return wn;
}

// COMPACT: every score of the pair stays below 256 (a 151-bp read's extension always does: seed score + query length <=
// read length) - the cell is 16 bits (h:8 | e:8), the cells of columns 2p and 2p+1 share the lane's dword of row p of the
// cell plane ([column pair][lane] dwords: bank = lane for any column), and the query codes of a column pair are a
// halfword of row p of a second plane ([pair][lane & 31][lane >> 5]: lanes l and l + 32 are served in different LDS
// cycles, so again no two lanes of a group meet in a bank).  3 bytes per column and lane instead of 4: queries up to
// ~100 long run at two wavefronts per SIMD and more; a column pair is one dword read, one halfword read and one dword
// write, and its fields are byte operands of the arithmetic (SDWA), never unpacked.
#ifdef GBX_BSW_LANE_STATS
// development aid (scripts/dbg_bsw_lanes.py): how full the lock-step rows are.  Per launch slot: [0] cells the lanes
// computed, [1] 64 x the widest window of every row (what the wavefront paid), [2] rows the lanes ran, [3] 64 x rows the
// wavefront ran; units of 256.
__device__ unsigned long long g_bsw_lane_stats[16][4];
#endif
// PACKED: P.ref / P.qer are the packed images of the arenas (two base codes per byte, host_pipeline.h: pack4; code k of an
// arena is nibble k & 1 of byte k >> 1) - the pipelined host call's chunks whose pairs all run here are never expanded.
template <bool SYM, bool COMPACT, bool CODE4 = false, bool PACKED = false>
__global__ void __launch_bounds__(64) bsw_lane_kernel(BswDev prm, BswPairs P, BswWork W, int rlo, int rhi, int cols, int slot)
{
    extern __shared__ uint32_t lcell[];
#ifdef GBX_BSW_LANE_STATS
    __shared__ int st_rowmax[2048];
    for (int k = threadIdx.x; k < 2048; k += 64) st_rowmax[k] = 0;
    unsigned long long st_cells = 0, st_rows = 0, st_wcells = 0, st_wrows = 0;
#endif
#define LCELL(byte_addr) (*(uint32_t *)((char *)lcell + (byte_addr)))
#define LCELL16(byte_addr) (*(uint16_t *)((char *)lcell + (byte_addr)))
#define LQ8(byte_addr) (*((uint8_t *)lcell + (byte_addr)))
    const int lane = threadIdx.x;
    const int first = W.lbase[rlo << 8], count = W.lbase[(rhi + 1) << 8] - first;
    const int nchunks = (count + 63) >> 6;
    const int32_t *order = W.lorder + first;
    const int e_ins = prm.e_ins, e_del = prm.e_del, oe_ins = prm.oe_ins, oe_del = prm.oe_del;
    // wide: cell of column j = dword at j * 256 + lane * 4
    // compact: cell of column j = halfword at (j >> 1) * 256 + lane * 4 + (j & 1) * 2; query code of column j = byte at
    //          qb + (j >> 1) * 128 + (j & 1)
    const int cb = lane * 4;
    // CODE4 (round 5, the two longest compact classes): the query codes of a column pair FOUR bits each, a byte per pair and lane - 2.5
    // instead of 3 bytes of LDS per column and lane: 7 instead of 6 wavefronts of the 100..135 class per CU, 6 instead of 5 of the
    // class above - for three more instructions per column pair that expand the byte into the v_perm_b32 selector.  Measured
    // (profiles/r05p_bsw_query_codes_ab.txt): the step's loop alone 1 094 -> 1 243 G cells/s at those occupancies; in the job the
    // 100..135 launch 16 % shorter in isolation, the 80..99 and 48..79 launches 7 % and 2 % LONGER (they have the wavefronts already,
    // and 64-byte rows let lanes of a quad that stand two pairs apart meet in an LDS bank): so only where it pays.  A conflict-free
    // form (the codes of four pairs as one dword per lane, lined up by v_alignbit_b32) was slower than either: the job as a whole
    // runs at 80 % of its VALU issue bound, and that form's extra instructions are paid by every class.
    constexpr int QROW = CODE4 ? 64 : 128;                      // bytes of the code plane per column pair
    const int qb = (cols >> 1) * 256 + (CODE4 ? lane : (lane & 31) * 4 + (lane >> 5) * 2);       // cols is even; halfwords: bank = lane & 31 for any column
    auto LQREAD = [&](int a) -> uint32_t { return CODE4 ? (uint32_t)LQ8(a) : (uint32_t)LCELL16(a); };
    auto QSEL = [](uint32_t q) -> uint32_t { return CODE4 ? (((q >> 4) << 8) | (q & 15u)) : q; };
    auto QLOW = [](uint32_t q) -> uint32_t { return CODE4 ? (q & 15u) : (q & 0xffu); };
    auto cell_at = [&](int j) { return COMPACT ? (j >> 1) * 256 + cb + (j & 1) * 2 : j * 256 + cb; };
    // the scoring matrix by target base, behind the planes: {bytes against A C G T, byte against N, 6-bit fields (wide format), -}
    // (in the last 128 bytes of the look-ahead rows: nothing is written there, and what a look-ahead load reads is not used)
    const int tab = COMPACT ? (cols >> 1) * (256 + QROW) + 512 : (cols + 2) * 256 - 128;
    if (lane < 5) *(uint4 *)((char *)lcell + tab + lane * 16) = make_uint4(prm.row0[lane], prm.row1[lane], prm.lrow[lane], 0u);
#define LTAB(base_code) (*(const uint4 *)((const char *)lcell + tab + min((int)(base_code), 4) * 16))
    int rel0 = 0, rel1 = 2, rel2 = 256, rel3 = 258, rel4 = 512, rel5 = 514, rel6 = 768, rel7 = 770;      // cell offsets within a trip of the main loop
    asm volatile("" : "+v"(rel0), "+v"(rel1), "+v"(rel2), "+v"(rel3), "+v"(rel4), "+v"(rel5), "+v"(rel6), "+v"(rel7));
    for (;;) {
        int c = 0;
        if (lane == 0) c = atomicAdd(&W.lchunk[slot], 1);
        c = __builtin_amdgcn_readfirstlane(c);
        if (c >= nchunks) break;
        // the sorted list ascends in query length: chunks are taken from its end, so the kernel's tail is short pairs
        const int hi_ = count - (c << 6), idx = hi_ - 64 + lane;
        const bool have = idx >= 0;                     // only the last chunk of a launch is ragged: its spare lanes repeat entry 0
        const int pair = order[have ? idx : 0];
        const int qlen = P.len2[pair], tlen = P.len1[pair], h0 = P.h0[pair];
        const int64_t qo = P.idq[pair], to = P.idr[pair];
        const uint8_t *q = P.qer + (PACKED ? 0 : qo);
        const uint8_t *t = P.ref + (PACKED ? 0 : to);
        // sixteen query codes from column j on, as bytes; packed: nine bytes hold them, whichever half of the first one they start in
        auto qpiece = [&](int j) -> uint4 {
            uint4 r;
            if (!PACKED) { __builtin_memcpy(&r, q + j, 16); return r; }
            const int64_t o = qo + j;
            const uint8_t *b = q + (o >> 1);
            uint64_t v;
            __builtin_memcpy(&v, b, 8);
            if (o & 1) v = (v >> 4) | ((uint64_t)b[8] << 60);
            auto spread = [](unsigned h) -> unsigned { return (h & 0xfu) | ((h & 0xf0u) << 4) | ((h & 0xf00u) << 8) | ((h & 0xf000u) << 12); };
            const unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
            r.x = spread(lo & 0xffffu); r.y = spread(lo >> 16); r.z = spread(hi & 0xffffu); r.w = spread(hi >> 16);
            return r;
        };
        // the target's base of row i (the arenas, packed or not, are readable 16 codes past their last base)
        auto tbase = [&](int i) -> int {
            if (!PACKED) return t[i];
            const int64_t o = to + i;
            return (t[o >> 1] >> ((int)(o & 1) * 4)) & 15;
        };
        // first row, :155-157, and the query codes (x 6: the bit offset of the score field in the matrix row word).  The query
        // comes in 16-byte pieces, the next one requested before the current one is unpacked (round 4: a byte load per column,
        // each waited for on the spot, was ~130 serial memory round trips per chunk of pairs - a tenth of a long-query
        // chunk's time, at 1.5 wavefronts per SIMD with little to hide it behind); the arenas are readable 16 bytes past
        // their last base.  Compact format: a column pair is one dword of cells and one halfword of codes.
        {
            uint4 wq = qpiece(0);
            int hrun = h0 - oe_ins + e_ins;                 // h0 - oe_ins - (j - 1) e_ins at j = 0
            for (int j0 = 0; j0 <= qlen; j0 += 16) {
                const uint4 cur = wq;
                if (j0 + 16 <= qlen) wq = qpiece(j0 + 16);
                const uint32_t ww[4] = {cur.x, cur.y, cur.z, cur.w};
#pragma unroll
                for (int c = 0; c < 16; c += 2) {
                    const int j = j0 + c;
                    if (j <= qlen) {
                        const uint32_t two = (ww[c >> 2] >> ((c & 3) * 8)) & 0xffffu;
                        const int c0 = j < qlen ? min((int)(two & 0xffu), 4) : 0, c1 = j + 1 < qlen ? min((int)(two >> 8), 4) : 0;
                        const int hv0 = j == 0 ? h0 : max(hrun, 0), hv1 = max(hrun - e_ins, 0);
                        hrun -= 2 * e_ins;
                        if (COMPACT) {
                            // (cell j + 1 may lie one past qlen: it exists - look-ahead rows - and is never live)
                            LCELL((j >> 1) * 256 + cb) = ((uint32_t)hv0 << 8) | ((uint32_t)hv1 << 24);
                            if (CODE4) LQ8(qb + (j >> 1) * QROW) = (uint8_t)(c0 | (c1 << 4));
                            else LCELL16(qb + (j >> 1) * QROW) = (uint16_t)(c0 | (c1 << 8));
                        } else {
                            LCELL(cell_at(j)) = ((uint32_t)hv0 << 18) | (uint32_t)(c0 * 6);
                            if (j + 1 <= qlen) LCELL(cell_at(j + 1)) = ((uint32_t)hv1 << 18) | (uint32_t)(c1 * 6);
                        }
                    }
                }
            }
        }
        const int w = band_width(prm, qlen);
        int best = h0, best_i = -1, best_j = -1, g_i = -1, g_score = -1, off = 0;
        int beg = 0, end = qlen;
        // the matrix row of the target base two rows ahead of its use: row i's in registers, row i+1's on its way from the
        // table, the base of row i+2 on its way from memory (the arenas are readable 16 bytes past their last base)
        uint4 mrow = LTAB(tbase(0)), mnext = LTAB(tbase(1));
        int tb2 = tbase(2);
        const uint8_t *tp = t + 3;
        int hrow = h0 - prm.o_del - e_del;                 // h0 - (o_del + e_del * (i + 1))
        int imw = -w, ipw = w + 1;                         // i - w, i + w + 1
        for (int i = 0; i < tlen; ++i) {
            const uint32_t rw = COMPACT ? mrow.x : mrow.z, rwn = mrow.y;
            mrow = mnext;
            mnext = LTAB(tb2);
            if (PACKED) tb2 = tbase(i + 3);
            else tb2 = *tp++;
            beg = max(beg, imw);                           // :179-181
            end = min(min(end, ipw), qlen);
            ++imw; ++ipw;
#ifdef GBX_BSW_LANE_STATS
            if (have) { st_cells += (unsigned)max(end - beg, 0); st_rows += 1; atomicMax(&st_rowmax[i & 2047], max(end - beg, 0) + 1); }
#endif
            int left = beg == 0 ? max(hrow, 0) : 0;        // :183-186
            hrow -= e_del;
            int f = 0;
            uint32_t key = 0;                              // (row maximum << 18) | byte address of the cell of its last arg-max
            int vzero = 0;
            asm volatile("" : "+v"(vzero));                  // a zero in a vector register (SDWA compare operand)
            uint32_t sb0 = 0, sb1 = 0, se0 = 0, se1 = 0;    // compact: the cells the next window is decided on (below)
            const int sp0 = beg >> 1, sq = max((end >> 1) - 1, 0);
            if (COMPACT) {
                // one column, :187-212: h8 / e8 = the cell's fields (byte operands), qo = the query code's field offset; returns the new cell
                auto step = [&](int diag, int e, uint32_t qc, int at) -> uint32_t {
                    const int sc = (int)(int8_t)__builtin_amdgcn_perm(rwn, rw, qc);
                    const int m = diag ? diag + sc : 0;    // :196
                    const int h = imax3(m, e, f);
                    key = max(key, ((uint32_t)h << 18) | (uint32_t)at);          // ties: the larger address wins = last arg-max, :200-201
                    const int td = m - oe_del;
                    const int en = imax3(e - e_del, td, 0);                      // E(i+1,j), :202-206
                    f = imax3(f - e_ins, SYM ? td : m - oe_ins, 0);              // F(i,j+1), :207-210
                    const uint32_t cell = ((uint32_t)left << 8) | (uint32_t)en;  // eh[j] = {H(i,j-1), E(i+1,j)}
                    left = h;
                    return cell;
                };
                // An odd first column is the high half of a column pair whose low half lies before the window.  It used to be a step
                // of its own (25 instructions and two LDS round trips at the head of every row, nothing to overlap them with); now
                // the pair runs as a whole with the low half's input cell forced to zero (round 4): a zero cell yields m = 0, h = 0,
                // e = 0 and leaves f = 0, which is exactly the state the scalar loop enters column beg with (:183-186: h1 = 0 for
                // beg > 0), its key (h = 0) never wins the row maximum, and what the step writes into the cell before the window
                // is never read again: windows only move right (beg is non-decreasing, :230-233).
                const bool odd_first = (beg & 1) && beg < end;
                int j = odd_first ? beg - 1 : beg;
                // Whole column pairs, four per trip, the loads two to four pairs ahead of their use (an LDS round trip is 100+
                // cycles under load and the wavefront has nothing else to do meanwhile).  Two register sets take turns, so
                // nothing is copied.  Loads past the window read cells that exist (look-ahead rows are part of the allocation)
                // and are not used.
                int pa = (j >> 1) * 256 + cb, qa = qb + (j >> 1) * QROW;
                uint32_t w0 = LCELL(pa), q0 = LQREAD(qa), w1 = LCELL(pa + 256), q1 = LQREAD(qa + QROW);
                w0 &= odd_first ? 0xffff0000u : 0xffffffffu;
                for (; j + 7 < end; j += 8, pa += 1024, qa += 4 * QROW) {
                    const uint32_t x0 = LCELL(pa + 512), y0 = LQREAD(qa + 2 * QROW), x1 = LCELL(pa + 768), y1 = LQREAD(qa + 3 * QROW);
                    uint32_t kt;                            // the trip's maximum, positions relative to pa (eight constant registers)
                    LCELL(pa) = lane_pair_step<SYM>(w0, QSEL(q0), rw, rwn, f, left, (uint32_t)vzero, kt, rel0, rel1, vzero, oe_del, oe_ins, e_del, e_ins);
                    LCELL(pa + 256) = lane_pair_step<SYM>(w1, QSEL(q1), rw, rwn, f, left, kt, kt, rel2, rel3, vzero, oe_del, oe_ins, e_del, e_ins);
                    w0 = LCELL(pa + 1024); q0 = LQREAD(qa + 4 * QROW); w1 = LCELL(pa + 1280); q1 = LQREAD(qa + 5 * QROW);
                    LCELL(pa + 512) = lane_pair_step<SYM>(x0, QSEL(y0), rw, rwn, f, left, kt, kt, rel4, rel5, vzero, oe_del, oe_ins, e_del, e_ins);
                    LCELL(pa + 768) = lane_pair_step<SYM>(x1, QSEL(y1), rw, rwn, f, left, kt, kt, rel6, rel7, vzero, oe_del, oe_ins, e_del, e_ins);
                    key = max(key, kt + (uint32_t)pa);
                }
                if (j + 3 < end) {
                    const uint32_t x0 = LCELL(pa + 512), y0 = LQREAD(qa + 2 * QROW);
                    LCELL(pa) = lane_pair_step<SYM>(w0, QSEL(q0), rw, rwn, f, left, key, key, pa, pa + 2, vzero, oe_del, oe_ins, e_del, e_ins);
                    LCELL(pa + 256) = lane_pair_step<SYM>(w1, QSEL(q1), rw, rwn, f, left, key, key, pa + 256, pa + 258, vzero, oe_del, oe_ins, e_del, e_ins);
                    w0 = x0; q0 = y0;
                    w1 = LCELL(pa + 768); q1 = LQREAD(qa + 3 * QROW);
                    j += 4; pa += 512; qa += 2 * QROW;
                }
                if (j + 1 < end) {
                    LCELL(pa) = lane_pair_step<SYM>(w0, QSEL(q0), rw, rwn, f, left, key, key, pa, pa + 2, vzero, oe_del, oe_ins, e_del, e_ins);
                    w0 = w1; q0 = q1;
                    j += 2; pa += 256; qa += QROW;
                }
                if (j < end) {                             // even last column: the low half of its dword, alone
                    LCELL16(pa) = (uint16_t)step((int)((w0 >> 8) & 0xffu), (int)(w0 & 0xffu), QLOW(q0), pa);
                }
                LCELL16(cell_at(end)) = (uint16_t)(left << 8);                   // eh[end] = {h1, 0}, :213
                // The next window (:230-233) starts at the first non-zero cell from beg on and ends two past the last one up to
                // end: the four cells from beg's pair on and the four up to end's pair are requested now, behind the row's last
                // write, and arrive while the row's results are evaluated - no LDS round trip per probed cell.
                sb0 = LCELL(sp0 * 256 + cb); sb1 = LCELL(sp0 * 256 + 256 + cb);
                se0 = LCELL(sq * 256 + cb); se1 = LCELL(sq * 256 + 256 + cb);
            } else {
                uint32_t left18 = (uint32_t)left << 18;
                int ab = cell_at(beg);
                const int abend = cell_at(end);
                uint32_t cw = LCELL(ab);
#pragma unroll 2
                for (; ab < abend; ab += 256) {
                    const uint32_t nw = LCELL(ab + 256);
                    const int diag = (int)(cw >> 18), e = (int)((cw >> 5) & 0x1fffu);
                    const uint32_t qo = cw & 31u;
                    const int sc = __builtin_amdgcn_sbfe((int)rw, qo, 6u);
                    const int m = diag ? diag + sc : 0;
                    const int h = imax3(m, e, f);
                    const uint32_t h18 = (uint32_t)h << 18;
                    key = max(key, h18 | (uint32_t)ab);
                    const int td = m - oe_del;
                    const int en = imax3(e - e_del, td, 0);
                    f = imax3(f - e_ins, SYM ? td : m - oe_ins, 0);
                    LCELL(ab) = left18 | ((uint32_t)en << 5) | qo;
                    left18 = h18;
                    cw = nw;
                }
                left = (int)(left18 >> 18);
                LCELL(abend) = left18 | (LCELL(abend) & 31u);
            }
            const int jfin = beg < end ? end : beg;
            if (jfin == qlen) {                             // :214-217
                if (!(g_score > left)) g_i = i;
                g_score = max(g_score, left);
            }
            const int row_best = (int)(key >> 18);
            const int karg = (int)(key & 0x3ffffu) - cb;    // byte offset of the arg-max cell from the lane's first cell
            const int row_arg = COMPACT ? ((karg >> 8) << 1) | ((karg >> 1) & 1) : karg >> 8;
            if (row_best == 0) break;                       // :218
            if (row_best > best) {                          // :219-221
                best = row_best; best_i = i; best_j = row_arg;
                off = max(off, abs(row_arg - i));
            } else if (prm.zdrop > 0) {                     // :222-228
                const int di = i - best_i, dj = row_arg - best_j;
                if (di > dj) { if (best - row_best - (di - dj) * e_del > prm.zdrop) break; }
                else if (best - row_best - (dj - di) * e_ins > prm.zdrop) break;
            }
            // the next row's window, :230-233 (h == 0 and e == 0 <=> the cell is zero / the word is below 32)
            int j = beg;
            if (COMPACT) {
                // cells 2 sp0 .. 2 sp0 + 3: the first non-zero one at or after beg (beg is 2 sp0 or 2 sp0 + 1)
                int k = (sb1 >> 16) ? 3 : 4;
                k = (sb1 & 0xffffu) ? 2 : k;
                k = (sb0 >> 16) ? 1 : k;
                k = (sb0 & 0xffffu) && !(beg & 1) ? 0 : k;
                j = 2 * sp0 + k;
                if (k == 4) while (j < end && LCELL16(cell_at(j)) == 0) ++j;      // four zero cells in a row: rare
                j = min(j, end);                            // cells at and beyond end do not count (they hold older rows)
                const int nbeg = beg < end ? j : beg;
                // cells 2 sq .. 2 sq + 3 reach end (and past it): the last non-zero one up to end
                const int khi = end - 2 * sq;              // 2 or 3; 0 or 1 while end < 2
                k = (se0 & 0xffffu) ? 0 : -1;
                k = (se0 >> 16) && khi >= 1 ? 1 : k;
                k = (se1 & 0xffffu) && khi >= 2 ? 2 : k;
                k = (se1 >> 16) && khi >= 3 ? 3 : k;
                j = 2 * sq + k;
                if (k < 0 && 2 * sq > nbeg) while (j >= nbeg && LCELL16(cell_at(j)) == 0) --j;      // rare
                j = max(j, nbeg - 1);
                j = nbeg > end ? end : j;
                beg = nbeg;
            } else {
                while (j < end && LCELL(cell_at(j)) < 32u) ++j;
                beg = j;
                j = end;
                while (j >= beg && LCELL(cell_at(j)) < 32u) --j;
            }
            end = min(j + 2, qlen);
        }
        gbx_bsw_result r;
        r.score = best; r.tle = best_i + 1; r.gtle = g_i + 1; r.qle = best_j + 1; r.gscore = g_score; r.max_off = off;
        if (have) P.out[pair] = r;
#ifdef GBX_BSW_LANE_STATS
        __builtin_amdgcn_s_waitcnt(0);
        for (int k = threadIdx.x; k < 2048; k += 64) { const int m = st_rowmax[k]; if (m) { st_wcells += 64ull * (unsigned)(m - 1); st_wrows += 64; } st_rowmax[k] = 0; }
#endif
    }
#ifdef GBX_BSW_LANE_STATS
    atomicAdd(&g_bsw_lane_stats[slot][0], st_cells); atomicAdd(&g_bsw_lane_stats[slot][1], st_wcells);
    atomicAdd(&g_bsw_lane_stats[slot][2], st_rows); atomicAdd(&g_bsw_lane_stats[slot][3], st_wrows);
#endif
#undef LCELL
#undef LCELL16
#undef LQ8
#undef LTAB
}

// ---- kernel shapes ----------------------------------------------------------
// class c (query length) -> (lanes per pair, columns per lane).  Short queries use narrow groups:
// the per-row fixed cost (scan, reductions, epilogue) is paid once per wavefront row, so 16 pairs per
// wavefront amortise it 4x better than 4 pairs; the width is bounded by the registers CPL columns need.
struct RowShape { int lpp, cpl; };
typedef void (*RowsFn)(BswDev, BswPairs, BswWork, int);
struct RowKernel { int lpp, cpl; RowsFn fn[2]; std::atomic<int> bpc[2]; const char *name; };   // bpc: resident blocks per CU, cached (same on every MI355X of a node)
#define GBX_ROW_KERNEL(L, C) { L, C, { bsw_rows_kernel<L, C, false>, bsw_rows_kernel<L, C, true> }, { 0, 0 }, "bsw_rows_" #L "x" #C }
RowKernel row_kernels[] = {
    // the default table (class_shapes) ...
    GBX_ROW_KERNEL(2, 4),  GBX_ROW_KERNEL(2, 8),  GBX_ROW_KERNEL(2, 12), GBX_ROW_KERNEL(2, 16), GBX_ROW_KERNEL(2, 20), GBX_ROW_KERNEL(2, 24),
    GBX_ROW_KERNEL(4, 14), GBX_ROW_KERNEL(4, 16), GBX_ROW_KERNEL(4, 18), GBX_ROW_KERNEL(4, 20), GBX_ROW_KERNEL(4, 22), GBX_ROW_KERNEL(4, 24),
    GBX_ROW_KERNEL(8, 13), GBX_ROW_KERNEL(8, 14), GBX_ROW_KERNEL(8, 15), GBX_ROW_KERNEL(8, 16),
    GBX_ROW_KERNEL(16, 10), GBX_ROW_KERNEL(16, 12), GBX_ROW_KERNEL(16, 16), GBX_ROW_KERNEL(64, 16),
    // ... and alternatives for scripts/tune_bsw_shapes.sh
    GBX_ROW_KERNEL(4, 8), GBX_ROW_KERNEL(4, 12), GBX_ROW_KERNEL(8, 10), GBX_ROW_KERNEL(8, 12), GBX_ROW_KERNEL(16, 8),
    // ... and for the direct launch of small jobs (bsw_launch_direct): a job of a few hundred pairs is as long as one pair's rows,
    // so its pairs get many lanes and few columns each
    GBX_ROW_KERNEL(64, 3), GBX_ROW_KERNEL(64, 4),
};
#undef GBX_ROW_KERNEL
// widest query a class holds: classes 0..7 = 16,32,..,128; 8..11 = 160,192,256,1024
constexpr int class_qmax[NCLS - 1] = {8, 16, 24, 32, 40, 48, 56, 64, 72, 80, 88, 96, 104, 112, 120, 128, 160, 192, 256, 1024};

// Class modes.  Every class is a kernel launch that lasts at least as long as its longest pair, and a stream runs
// its classes one after the other: with few pairs the 20 fine classes are latency, not work (512 pairs: 1.3 ms on
// the device for 0.05 ms of arithmetic).  Small jobs therefore run narrower queries on a wider class's kernel
// (idle lanes are free there): mode 2 = three classes, mode 1 = six, mode 0 = all twenty.  Thresholds from
// `bench.py --size` sweeps; GBX_BSW_CLASSMODE overrides.
constexpr uint8_t CLASS_REMAP[3][NCLS] = {
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20},
    {3, 3, 3, 3, 7, 7, 7, 7, 11, 11, 11, 11, 15, 15, 15, 15, 18, 18, 18, 19, 20},
    {15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 18, 18, 18, 19, 20},
};
int class_mode_for(int64_t n)
{
    const char *e = getenv("GBX_BSW_CLASSMODE");           /* read per call: the tests vary it */
    if (e && *e >= '0' && *e <= '2') return *e - '0';
    return n < 32768 ? 2 : n < 250000 ? 1 : 0;
}

RowKernel *find_row_kernel(int lpp, int cpl)
{
    for (RowKernel &k : row_kernels)
        if (k.lpp == lpp && k.cpl == cpl) return &k;
    return nullptr;
}

// GBX_BSW_SHAPES="4x4,4x8,..." (12 entries) overrides the table; a tuning aid, entries that do not
// cover their class or have no kernel are ignored.
const RowShape *class_shapes()
{
    static RowShape shapes[NCLS - 1] = {{2, 4},  {2, 8},  {2, 12}, {2, 16}, {2, 20}, {2, 24}, {4, 14}, {4, 16}, {4, 18}, {4, 20},
                                        {4, 22}, {4, 24}, {8, 13}, {8, 14}, {8, 15}, {8, 16}, {16, 10}, {16, 12}, {16, 16}, {64, 16}};
    static bool parsed = false;
    if (!parsed) {
        parsed = true;
        const char *e = getenv("GBX_BSW_SHAPES");
        for (int c = 0; e && *e && c < NCLS - 1; ++c) {
            int l = 0, k = 0, used = 0;
            if (sscanf(e, "%dx%d%n", &l, &k, &used) != 2) break;
            if (find_row_kernel(l, k) && l * k >= class_qmax[c]) shapes[c] = {l, k};
            e += used;
            if (*e == ',') ++e;
        }
    }
    return shapes;
}

int make_dev_params(const gbx_bsw_params *p, BswDev *d)
{
    if (p->e_del < 1 || p->e_ins < 1 || p->o_del < 0 || p->o_ins < 0 || p->e_del > 4096 || p->e_ins > 4096 ||
        p->o_del > (1 << 16) || p->o_ins > (1 << 16) || p->w < 0) {
        set_error("bsw: gap penalties must satisfy 0<=o<=65536, 1<=e<=4096, w>=0");
        return GBX_ERR_ARG;
    }
    d->o_del = p->o_del; d->e_del = p->e_del; d->o_ins = p->o_ins; d->e_ins = p->e_ins;
    d->oe_del = p->o_del + p->e_del; d->oe_ins = p->o_ins + p->e_ins;
    d->zdrop = p->zdrop; d->end_bonus = p->end_bonus; d->w = p->w;
    int mx = 0;
    for (int k = 0; k < 25; ++k) mx = p->mat[k] > mx ? p->mat[k] : mx;   // bandedSWA.cpp:160-162
    d->max_mat = mx;
    for (int q = 0; q < 5; ++q) {
        uint32_t wv = 0;
        for (int t = 0; t < 4; ++t) wv |= (uint32_t)(uint8_t)p->mat[t * 5 + q] << (8 * t);
        d->colword[q] = wv;
        d->col4[q] = p->mat[4 * 5 + q];
    }
    for (int t = 0; t < 5; ++t) {
        uint32_t wv = 0;
        for (int q = 0; q < 4; ++q) wv |= (uint32_t)(uint8_t)p->mat[t * 5 + q] << (8 * q);
        d->row0[t] = wv;
        d->row1[t] = (uint32_t)(uint8_t)p->mat[t * 5 + 4];
    }
    // lane path: the matrix row as five signed 6-bit fields; scorings outside [-32, 31] keep every pair on the row kernels
    d->lane_on = 0;
    bool fits6 = true;
    for (int k = 0; k < 25; ++k) fits6 = fits6 && p->mat[k] >= -32 && p->mat[k] <= 31;
    for (int t = 0; t < 5; ++t) {
        uint32_t wv = 0;
        for (int q = 0; q < 5; ++q) wv |= ((uint32_t)p->mat[t * 5 + q] & 63u) << (6 * q);
        d->lrow[t] = wv;
    }
    if (fits6 && p->o_del + p->e_del < LANE_SCORE_LIMIT && p->o_ins + p->e_ins < LANE_SCORE_LIMIT) d->lane_on = -1;   // -1: allowed, the launch decides
    return GBX_OK;
}

}  // namespace

// ints: header | order[n] | wband[n] | lorder[n] | lrank[n] | lbase[LANE_BINS + 1] (+pad) | lchunk[64]
constexpr int64_t WS_LANE = (int64_t)(LANE_BINS + 64) + 64;
size_t bsw_workspace_bytes(int64_t n)
{
    return (size_t)(WS_HDR + 4 * (n > 0 ? n : 0) + WS_LANE) * sizeof(int32_t);
}

// Expands the packed image of a byte arena (two base codes per byte, host_pipeline.h: pack4) over [lo, hi) of the
// arena, lo even: 4 x 8 packed bytes -> 4 x 16 codes per thread, a block's four rounds each one contiguous 4 KB (in a
// pipelined host call this kernel runs beside the previous chunk's lane kernels and gets a wavefront slot or two per
// SIMD: few wavefronts with four loads in flight each, not many with one - 0.5-1.0 ms -> see profiles/r05af_*).
constexpr int UNPACK_ROUNDS = 4;
__global__ void __launch_bounds__(256) bsw_unpack4_kernel(const uint8_t *__restrict__ packed, uint8_t *__restrict__ out, int64_t lo, int64_t hi)
{
    const int64_t base = lo + (int64_t)blockIdx.x * (256 * 16 * UNPACK_ROUNDS) + threadIdx.x * 16;
    const bool aligned = (((uintptr_t)(packed + (base >> 1)) & 7) == 0) && (((uintptr_t)(out + base) & 15) == 0);
    if (aligned && base + (int64_t)(UNPACK_ROUNDS - 1) * 4096 + 16 <= hi) {
        uint2 v[UNPACK_ROUNDS];
#pragma unroll
        for (int r = 0; r < UNPACK_ROUNDS; ++r) v[r] = *(const uint2 *)(packed + ((base + r * 4096) >> 1));
        // byte b = lo | hi << 4  ->  two bytes (lo, hi); four packed bytes give two dwords
        auto spread = [](unsigned h) -> unsigned { return (h & 0xfu) | ((h & 0xf0u) << 4) | ((h & 0xf00u) << 8) | ((h & 0xf000u) << 12); };
#pragma unroll
        for (int r = 0; r < UNPACK_ROUNDS; ++r) {
            uint4 q;
            q.x = spread(v[r].x & 0xffffu); q.y = spread(v[r].x >> 16); q.z = spread(v[r].y & 0xffffu); q.w = spread(v[r].y >> 16);
            *(uint4 *)(out + base + r * 4096) = q;
        }
        return;
    }
    for (int r = 0; r < UNPACK_ROUNDS; ++r) {
        const int64_t o = base + r * 4096;
        if (o >= hi) return;
        if (aligned && o + 16 <= hi) {
            const uint2 v = *(const uint2 *)(packed + (o >> 1));
            auto spread = [](unsigned h) -> unsigned { return (h & 0xfu) | ((h & 0xf0u) << 4) | ((h & 0xf00u) << 8) | ((h & 0xf000u) << 12); };
            uint4 q;
            q.x = spread(v.x & 0xffffu); q.y = spread(v.x >> 16); q.z = spread(v.y & 0xffffu); q.w = spread(v.y >> 16);
            *(uint4 *)(out + o) = q;
        } else {
            for (int64_t k = o; k < o + 16 && k < hi; ++k) out[k] = (uint8_t)((packed[k >> 1] >> ((k & 1) * 4)) & 0xf);
        }
    }
}

int bsw_unpack4(const uint8_t *d_packed, uint8_t *d_out, int64_t lo, int64_t hi, hipStream_t s)
{
    if (hi <= lo) return GBX_OK;
    const int64_t per_block = 256 * 16 * UNPACK_ROUNDS;
    Stage st("bsw_unpack4", s);
    hipLaunchKernelGGL(bsw_unpack4_kernel, dim3((unsigned)((hi - lo + per_block - 1) / per_block)), dim3(256), 0, s, d_packed, d_out, lo, hi);
    GBX_HIP(hipGetLastError());
    return GBX_OK;
}

// Small jobs whose queries all fit one register class (1..256 columns, no empty sequence, scores below the
// packed-key limit; the caller has checked): one launch of the 8x16 or 16x16 kernel over the pairs in input order, without
// the binning passes, the stream fork and the join - the dependent launch chain is what a 512-pair call costs.
int bsw_launch_direct(const gbx_bsw_params *p, int64_t n, int max_qlen,
                      const uint8_t *d_ref, const uint8_t *d_qer, const int64_t *d_idr, const int64_t *d_idq,
                      const int32_t *d_len1, const int32_t *d_len2, const int32_t *d_h0, gbx_bsw_result *d_out, hipStream_t s)
{
    if (n <= 0) return GBX_OK;
    BswDev dev;
    int rc = make_dev_params(p, &dev);
    if (rc) return rc;
    for (int c = 0; c < NCLS; ++c) dev.remap[c] = (uint8_t)c;
    BswPairs P = {d_ref, d_qer, d_idr, d_idq, d_len1, d_len2, d_h0, d_out};
    BswWork W = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    dev.lane_on = 0;
    if (max_qlen > 256) { set_error("bsw: direct launch needs queries of at most 256"); return GBX_ERR_ARG; }
    // A job this small is as long as its slowest pair's rows, and a row's time grows with the columns a lane holds: many lanes per
    // pair, few columns each.  Measured (profiles/r06q_bsw_direct_shapes.txt, 151-bp pairs, whole host call): 512 pairs 0.340 ms on
    // 16x16, 0.261 on 16x10, 0.216 on 64x3; 8 192 pairs 0.550 / 0.448 / 0.589 - a wavefront per pair up to a few thousand pairs,
    // then the narrowest sixteen-lane shape that holds the longest query.
    const int cols = max_qlen + 1;
    int lpp = 16, cpl = cols <= 128 ? 8 : cols <= 160 ? 10 : cols <= 192 ? 12 : 16;
    if (n < 4096) { lpp = 64; cpl = cols <= 192 ? 3 : 4; }
    if (const char *e = getenv("GBX_BSW_DIRECT_SHAPE")) {        /* tuning aid: "LxC", must cover max_qlen + 1 columns */
        int l = 0, c = 0;
        if (sscanf(e, "%dx%d", &l, &c) == 2 && l * c > max_qlen && find_row_kernel(l, c)) { lpp = l; cpl = c; }
    }
    RowKernel *k = find_row_kernel(lpp, cpl);
    if (!k) { set_error("bsw: no %dx%d row kernel", lpp, cpl); return GBX_ERR_UNSUPPORTED; }
    const bool sym = dev.oe_ins == dev.oe_del;
    const int gpb = 256 / lpp;                                // groups (pairs in flight) per block
    const int blocks = (int)((n + gpb - 1) / gpb);
    Stage st(k->name, s);
    hipLaunchKernelGGL(k->fn[sym], dim3(blocks), dim3(256), 0, s, dev, P, W, -(int)n);
    GBX_HIP(hipGetLastError());
    return GBX_OK;
}

// lane path: large jobs only (a wavefront holds 64 pairs: the chip wants a few thousand wavefronts), GBX_BSW_LANE=0/1 overrides
static bool lane_wanted(int64_t n)
{
    const char *lane_env = getenv("GBX_BSW_LANE");              /* read per call: the tests vary it */
    const int64_t lane_min = getenv("GBX_BSW_LANE_MIN") ? atoll(getenv("GBX_BSW_LANE_MIN")) : 262144;
    return lane_env ? atoi(lane_env) != 0 : n >= lane_min;
}

int bsw_lane_rule(const gbx_bsw_params *p, int64_t n, BswLaneRule *r)
{
    BswDev dev;
    const int rc = make_dev_params(p, &dev);
    if (rc) return rc;
    r->on = dev.lane_on && lane_wanted(n) ? 1 : 0;
    r->max_mat = dev.max_mat > 0 ? dev.max_mat : 0;
    r->qmax = LANE_QMAX;
    r->limit = LANE_SCORE_LIMIT;
    r->compact_limit = LANE_COMPACT_LIMIT;
    static_assert(LANE_NRANGE == 5, "BswLaneRule::range_hi / BswChunkPrep::class_pairs");
    for (int f = 0; f < 2; ++f) for (int k = 0; k < LANE_NRANGE; ++k) r->range_hi[f][k] = LANE_RANGE_HI[f][k];
    return GBX_OK;
}

int bsw_launch(const gbx_bsw_params *p, int64_t n,
               const uint8_t *d_ref, const uint8_t *d_qer,
               const int64_t *d_idr, const int64_t *d_idq,
               const int32_t *d_len1, const int32_t *d_len2, const int32_t *d_h0,
               gbx_bsw_result *d_out, void *d_work, size_t work_bytes, hipStream_t s, hipEvent_t *join_events,
               const BswChunkPrep *prep)
{
    if (n == 0) return GBX_OK;
    if (n > 0x7fffffffLL - 1024) { set_error("bsw: more than 2^31 pairs in one call"); return GBX_ERR_UNSUPPORTED; }
    if (work_bytes < bsw_workspace_bytes(n)) { set_error("bsw: workspace too small"); return GBX_ERR_ARG; }
    BswDev dev;
    int rc = make_dev_params(p, &dev);
    if (rc) return rc;
    const int mode = class_mode_for(n);
    for (int c = 0; c < NCLS; ++c) dev.remap[c] = CLASS_REMAP[mode][c];
    BswPairs P = {d_ref, d_qer, d_idr, d_idq, d_len1, d_len2, d_h0, d_out};
    int32_t *wi = (int32_t *)d_work;
    int32_t *wl = wi + WS_HDR + 4 * n;
    BswWork W = {wi, wi + HDR, wi + 2 * HDR, wi + 3 * HDR, wi + WS_HDR, wi + WS_HDR + n, wl, wi + WS_HDR + 3 * n, wi + WS_HDR + 2 * n,
                 wl + LANE_BINS + 64};
    dev.lane_on = dev.lane_on && lane_wanted(n) ? 1 : 0;
    // the classes are independent and every kernel ends in a tail of a few long pairs: they go to four
    // streams so that a tail overlaps the next class (GBX_BSW_SERIAL=1 keeps them on the caller's stream)
    static const bool serial = getenv("GBX_BSW_SERIAL") != nullptr;
    SideStreams *ss = nullptr;
    std::unique_lock<std::mutex> side_lock;
    if (!serial) {
        if ((rc = side_streams(&ss))) return rc;
        side_lock = std::unique_lock<std::mutex>(ss->mu);
    }
    // A chunk of a pipelined host call: its preparing passes go to the two urgent streams and wait for the uploads only.
    // On the caller's stream they would queue behind the previous chunk's kernels there, and with them every kernel of
    // this chunk: the chunks then ran one after the other, each with its own tails (2.5 ms a chunk of 'large' against
    // 1.9 ms for a third of the job; GBX_BSW_PREP=0 keeps that order).
    static const bool prep_off = getenv("GBX_BSW_PREP") && atoi(getenv("GBX_BSW_PREP")) == 0;
    const bool ahead = prep && prep->uploaded && join_events && !serial && !prep_off;
    hipStream_t s_cls = ahead ? ss->pre[0] : s;
    // A chunk whose pairs all go to the lane kernels (the host entry has counted: rows_pairs == 0) leaves out the row-kernel
    // classes, twenty-one near-empty launches that each wait for LDS behind the lane kernels, and its bases stay packed: the
    // lane kernels read the nibbles (PACKED).  Beside the previous chunk's lane kernels the unpacking took 0.5-1.0 ms
    // instead of 0.05, and the chunk's kernels wait for it (profiles/r05af_host_timeline.txt).
    const bool no_rows = dev.lane_on && prep && prep->rows_pairs == 0 && !(getenv("GBX_BSW_SKIP_ROWS") && atoi(getenv("GBX_BSW_SKIP_ROWS")) == 0);
    const bool packed_lanes = no_rows && prep->ref_packed && !(getenv("GBX_BSW_PACKED_LANES") && atoi(getenv("GBX_BSW_PACKED_LANES")) == 0);
    // Such a chunk's launch can also come in two calls (BswChunkPrep::phase): the preparing passes as soon as the chunk's index
    // arrays are up - they read nothing else - and the kernels when its bases are.  A phase-1 call that cannot be split
    // (a chunk with row-kernel pairs, a switch set) queues nothing and returns 1: the caller then makes one whole call.
    const int phase = prep ? prep->phase : 0;
    if (phase && !(ahead && packed_lanes && prep->ev_pre && prep->ev_aux)) {
        if (phase == 1) return 1;
        set_error("bsw: the kernels of a chunk whose preparing passes were not queued");
        return GBX_ERR_ARG;
    }
    hipEvent_t ev_pre = phase ? prep->ev_pre : ss ? ss->ev_pre : nullptr, ev_aux = phase ? prep->ev_aux : ss ? ss->ev_aux : nullptr;
    const bool sort_aside = dev.lane_on && !serial;
    if (packed_lanes) { P.ref = prep->ref_packed; P.qer = prep->qer_packed; P.packed = 1; }
    if (phase != 2) {
        if (ahead) {
            GBX_HIP(hipStreamWaitEvent(ss->pre[0], prep->uploaded, 0));
            GBX_HIP(hipStreamWaitEvent(ss->pre[1], prep->uploaded, 0));
        }
        if (!packed_lanes && prep && prep->ref_packed) {
            // from the call's watermark (everything below it is expanded; chunks that ran PACKED expanded nothing) - s_cls waits
            // for this chunk's uploads, and the host entry queues the chunks' uploads in order, so all of [from, hi) is up
            const int64_t from_r = prep->unp_r ? *prep->unp_r : prep->lo_r, from_q = prep->unp_q ? *prep->unp_q : prep->lo_q;
            if ((rc = bsw_unpack4(prep->ref_packed, prep->ref_bytes, from_r, prep->hi_r, s_cls)) ||
                (rc = bsw_unpack4(prep->qer_packed, prep->qer_bytes, from_q, prep->hi_q, s_cls)))
                return rc;
            if (prep->unp_r && prep->hi_r > *prep->unp_r) *prep->unp_r = prep->hi_r;
            if (prep->unp_q && prep->hi_q > *prep->unp_q) *prep->unp_q = prep->hi_q;
        }
        GBX_HIP(hipMemsetAsync(d_work, 0, WS_HDR * sizeof(int32_t), s_cls));
        // The lane sort (0.3 ms on 'large': two passes of scattered atomics) runs on a side stream of its own, beside
        // classify and the row-kernel classes on the caller's stream, which do not need it; the lane launches wait for it.
        if (dev.lane_on) {
            hipStream_t so = s;
            if (ahead) {
                so = ss->pre[1];
                GBX_HIP(hipMemsetAsync(wl, 0, (size_t)WS_LANE * sizeof(int32_t), so));
            } else {
                GBX_HIP(hipMemsetAsync(wl, 0, (size_t)WS_LANE * sizeof(int32_t), s));
                if (sort_aside) {
                    if ((rc = ss->fork(s))) return rc;
                    so = ss->side[SideStreams::N - 1];
                }
            }
            Stage st("bsw_lane_sort", so);
            const int sblocks = (int)((n + 255) / 256);
            hipLaunchKernelGGL(bsw_lane_sort_kernel, dim3(sblocks), dim3(256), 0, so, dev, P, n, W, 0);
            hipLaunchKernelGGL(bsw_lane_scan_kernel, dim3(1), dim3(1024), 0, so, W);
            hipLaunchKernelGGL(bsw_lane_sort_kernel, dim3(sblocks), dim3(256), 0, so, dev, P, n, W, 1);
            if (sort_aside) GBX_HIP(hipEventRecord(ev_aux, so));
        }
        const int cblocks = (int)((n + CLS_THREADS - 1) / CLS_THREADS);
        {
            Stage st("bsw_classify", s_cls);
            hipLaunchKernelGGL(bsw_classify_kernel, dim3(cblocks), dim3(CLS_THREADS), 0, s_cls, dev, P, n, W, 0);
            hipLaunchKernelGGL(bsw_scan_kernel, dim3(1), dim3(HDR), 0, s_cls, W);
            hipLaunchKernelGGL(bsw_classify_kernel, dim3(cblocks), dim3(CLS_THREADS), 0, s_cls, dev, P, n, W, 1);
        }
        if (ahead) GBX_HIP(hipEventRecord(ev_pre, s_cls));
        GBX_HIP(hipGetLastError());
        if (phase == 1) return GBX_OK;
    }
    if (ahead) {                                                 // the kernel streams wait for classify (and, below, for the sort)
        GBX_HIP(hipStreamWaitEvent(s, ev_pre, 0));
        for (int k = 0; k < SideStreams::N; ++k) GBX_HIP(hipStreamWaitEvent(ss->side[k], ev_pre, 0));
        if (phase == 2) {                                        // ... and for the bases, which the preparing passes did not
            GBX_HIP(hipStreamWaitEvent(s, prep->uploaded, 0));
            for (int k = 0; k < SideStreams::N; ++k) GBX_HIP(hipStreamWaitEvent(ss->side[k], prep->uploaded, 0));
        }
    }

    int dev_id = 0, cus = 256;
    (void)hipGetDevice(&dev_id);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev_id);
    // persistent groups: enough blocks to fill the chip, never more groups than pairs
    auto grid_for = [&](int groups_per_block, int blocks_per_cu) {
        int64_t want = (n + groups_per_block - 1) / groups_per_block;
        int64_t cap = (int64_t)cus * blocks_per_cu;
        return (int)(want < cap ? want : cap);
    };
    // the grid is the number of blocks that are resident at once (occupancy query), so that the static
    // round-robin over the longest-first list starts every group on the longest pairs together
    const bool sym = dev.oe_ins == dev.oe_del;
    const RowShape *shapes = class_shapes();
    if (!serial && !sort_aside && !ahead && (rc = ss->fork(s))) return rc;
    // The row-kernel classes first.  With the lane path on they hold next to nothing (what the lane kernels cannot take):
    // twenty near-empty launches, all on the caller's stream, in the shadow of the lane sort.
    // (none at all when the host entry has counted the chunk's pairs and the lane kernels take every one: no_rows)
    int launched = 0;
    for (int c = 0; c < NCLS - 1 && !no_rows; ++c) {
        if (CLASS_REMAP[mode][c] != c) continue;          // this class's pairs run on a wider class's kernel
        // Four kernel streams when the inputs are resident.  The host pipeline (join_events) uses three: the
        // runtime maps streams onto four hardware queues, and with all four busy with class kernels its copy
        // stream shares one and the uploads stall behind kernels (measured; GBX_BSW_KSTREAMS overrides).
        static const int nk_env = getenv("GBX_BSW_KSTREAMS") ? atoi(getenv("GBX_BSW_KSTREAMS")) : 0;
        const int nk = nk_env > 0 ? nk_env : join_events ? 3 : 4;
        const int lane_k = mode ? launched++ % (nk > 4 ? 4 : nk) : nk >= 4 ? (c & 3) : c % nk;
        hipStream_t sc = serial || sort_aside || lane_k == 0 ? s : ss->side[lane_k - 1];
        RowKernel *k = find_row_kernel(shapes[c].lpp, shapes[c].cpl);
        if (!k) { set_error("bsw: no row kernel for class %d", c); return GBX_ERR_UNSUPPORTED; }
        int bpc = k->bpc[sym].load(std::memory_order_relaxed);
        if (!bpc) {
            int q = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&q, k->fn[sym], 256, 0) != hipSuccess || q < 1) {
                (void)hipGetLastError();
                q = 2;
            }
            bpc = q > 8 ? 8 : q;
            k->bpc[sym].store(bpc, std::memory_order_relaxed);
        }
        Stage st(k->name, sc);
        hipLaunchKernelGGL(k->fn[sym], dim3(grid_for(256 / k->lpp, bpc)), dim3(256), 0, sc, dev, P, W, c);
    }
    if (sort_aside) {                                          // the lane launches need the sorted lists (the sort's own stream has them in order)
        GBX_HIP(hipStreamWaitEvent(s, ev_aux, 0));
        for (int k = 0; k + (ahead ? 0 : 1) < SideStreams::N; ++k) GBX_HIP(hipStreamWaitEvent(ss->side[k], ev_aux, 0));
    }
    if (dev.lane_on) {
        // longest queries first, one launch per format and LDS class, spread over the streams; grids = resident wavefronts
        static const char *names[2][LANE_NRANGE] = {{"bsw_lane_c47", "bsw_lane_c79", "bsw_lane_c99", "bsw_lane_c135", "bsw_lane_c159"},
                                                    {"bsw_lane_w39", "bsw_lane_w79", "bsw_lane_w103", "bsw_lane_w127", "bsw_lane_w159"}};
        // The wide launches go first, while the chip is empty: a launch has to be given its LDS before its wavefronts can
        // see that their list is empty (the usual case for short reads), and behind a working compact launch that wait
        // held up the stream for up to a millisecond (6.36 -> 6.03 ms on 'large').
        for (int fmt = 1; fmt >= 0; --fmt) {
            int nl = 0;
            for (int r = LANE_NRANGE - 1; r >= 0; --r) {
                const int qlo = r ? LANE_RANGE_HI[fmt][r - 1] + 1 : 1, qhi = LANE_RANGE_HI[fmt][r];
                // GBX_BSW_SKIP_EMPTY=1: a class the host entry has counted empty is not launched and takes no stream's turn
                // (BswChunkPrep::class_pairs).  Measured on 'large', where six of the ten classes are empty, and NOT faster: medians
                // 9.97 / 10.25 ms with, 9.75 / 9.96 without (profiles/r06h_bsw_skip_empty_ab.txt) - the time an empty launch shows on
                // its stream is a wait for LDS its working successor would have spent just the same.  Off by default.
                static const bool skip_on = getenv("GBX_BSW_SKIP_EMPTY") && atoi(getenv("GBX_BSW_SKIP_EMPTY")) == 1;
                if (skip_on && prep && prep->class_known && prep->class_pairs[fmt * LANE_NRANGE + r] == 0) continue;
                ++nl;                                          // the launch's ordinal, longest first
                const int cols = (qhi + 3) & ~1;                   // columns 0..qlen, and even
                // compact: cols / 2 dword rows of cells, cols / 2 halfword rows of query codes, and what the look-ahead of the
                // query plane reads past its end (the cells' look-ahead lands in the query plane); wide: 2 columns of look-ahead
                // (four-bit query codes for the two longest compact classes: see bsw_lane_kernel; GBX_BSW_CODE4=0 / 1 forces none / all)
                const char *c4e = getenv("GBX_BSW_CODE4");
                const bool code4 = fmt == 0 && (c4e ? atoi(c4e) != 0 : qhi > 99);
                const size_t lds = fmt ? (size_t)(cols + 2) * 256 : (size_t)(cols / 2) * (code4 ? 320 : 384) + 640;
                int per_cu = (int)((size_t)160 * 1024 / lds);
                if (per_cu > 16) per_cu = 16;
                const int sk = (nl - 1) & 3;
                hipStream_t sc = serial || sk == 0 ? s : ss->side[sk - 1];
                int64_t blocks = (int64_t)cus * per_cu, want = (n + 63) / 64;
                if (blocks > want) blocks = want;
                const int rlo = fmt * (LANE_QMAX + 1) + qlo, rhi = fmt * (LANE_QMAX + 1) + qhi, slot = fmt * LANE_NRANGE + r;
                Stage st(names[fmt][r], sc);
                auto go = [&](auto kern) { hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(64), lds, sc, dev, P, W, rlo, rhi, cols, slot); };
                if (P.packed) {
                    if (fmt == 0 && code4) { if (sym) go(bsw_lane_kernel<true, true, true, true>); else go(bsw_lane_kernel<false, true, true, true>); }
                    else if (fmt == 0) { if (sym) go(bsw_lane_kernel<true, true, false, true>); else go(bsw_lane_kernel<false, true, false, true>); }
                    else { if (sym) go(bsw_lane_kernel<true, false, false, true>); else go(bsw_lane_kernel<false, false, false, true>); }
                } else if (fmt == 0 && code4) {
                    if (sym) go(bsw_lane_kernel<true, true, true>); else go(bsw_lane_kernel<false, true, true>);
                } else if (fmt == 0) {
                    if (sym) go(bsw_lane_kernel<true, true>); else go(bsw_lane_kernel<false, true>);
                } else {
                    if (sym) go(bsw_lane_kernel<true, false>); else go(bsw_lane_kernel<false, false>);
                }
            }
        }
    }
    // join_events == nullptr: the caller's stream waits for the side streams (everything of this call is then
    // ordered on `s`).  Otherwise nothing waits: one event per stream is recorded (join_events[0] on `s`,
    // join_events[1+k] on side stream k), and `s` and the side streams run on into the caller's next launch:
    // the host pipeline queues chunk after chunk like that, so that the single-wavefront tails of one chunk
    // overlap the next chunk's kernels, and its downloader waits for the events.
    if (!serial && !join_events && (rc = ss->join(s))) return rc;
    if (!no_rows) {
        const size_t lds_bytes = (size_t)(GBX_BSW_MAX_QLEN + 1) * 2 * sizeof(int);
        // per device (a process may drive several GPUs through gbx_set_device), set at most once each
        static std::atomic<uint64_t> attr_set[2];
        int cur = 0;
        GBX_HIP(hipGetDevice(&cur));
        const uint64_t bit = (uint64_t)1 << (cur & 63);
        if (cur >= 128 || !(attr_set[cur >> 6].load(std::memory_order_acquire) & bit)) {
            GBX_HIP(hipFuncSetAttribute((const void *)bsw_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
            if (cur < 128) attr_set[cur >> 6].fetch_or(bit, std::memory_order_release);
        }
        int blocks = (int)(n < (int64_t)cus * 2 ? n : (int64_t)cus * 2);
        Stage st("bsw_lds", s);
        hipLaunchKernelGGL(bsw_lds_kernel, dim3(blocks), dim3(64), lds_bytes, s, dev, P, W, CLS_LDS);
    }
    if (join_events) {
        if (!ss && (rc = side_streams(&ss))) return rc;
        GBX_HIP(hipEventRecord(join_events[0], s));
        for (int k = 0; k < SideStreams::N; ++k) GBX_HIP(hipEventRecord(join_events[1 + k], ss->side[k]));
    }
    GBX_HIP(hipGetLastError());
    return GBX_OK;
}

}  // namespace gbx

#ifdef GBX_BSW_LANE_STATS
extern "C" int gbx_debug_bsw_lane_stats(unsigned long long *out, int reset)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(gbx::g_bsw_lane_stats), sizeof(unsigned long long) * 64) != hipSuccess) return -1;
    if (reset) { unsigned long long z[64] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(gbx::g_bsw_lane_stats), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#endif
