// phmm_kernels.hip — GATK/GKL Pair-HMM forward likelihoods for gfx950 (MI355X).
//
// Replaces libgkl_pairhmm_c's computelikelihoodsboth(testcase*, double*, int)
// (declared R/benchmarks/phmm/PairHMMUnitTest.cpp:86, called :245).  The GKL
// sources are an empty submodule in the reference checkout; the algorithm is
// the published GKL/GATK logless Pair-HMM:
//   fp32 pass scaled by 2^120, redone in fp64 (2^1020) when the fp32 result is
//   below MIN_ACCEPTED = 1e-28f (pairhmm_common.h:16); output
//   log10(result) - log10(INITIAL_CONSTANT).
//
// Design: one (read, haplotype) pair per wavefront, swept as a systolic
// anti-diagonal wave.  Lane l owns RPL consecutive read rows ("slots"); at step
// s slot sigma handles haplotype column s - sigma.  A cell needs (r-1,c-1),
// (r-1,c) and (r,c-1): the last is the slot's own previous step, the first two
// are the slot above one and two steps ago — registers inside a lane, one
// wave_shr:1 DPP move per value across the lane boundary.  The haplotype base
// walks down the slots as a shift register fed at lane 0.  No scan, no LDS, no
// MFMA (a recurrence, not a contraction).  Rows are right-aligned so a tile's
// last row always sits in slot RPL-1 of the last used lane: the only slot that
// accumulates sum_c M[R][c] + X[R][c].  Reads longer than 64*RPL rows run as
// several row tiles, the tile's bottom DP row handed to the next tile through
// a small global scratch row (ping-pong).
//
// Reads of up to 248 rows — the case GATK produces — take the *stream* path
// (phmm_stream_kernel): pairs are grouped by read, the haplotypes of a read are
// laid end to end in one byte stream (a 0 byte in front of each = DP column 0),
// and half a wavefront (31 lanes x RPL rows) keeps the read's rows resident
// while the whole stream flows through: lane l is one column behind lane l-1,
// so the systolic pipeline fills once per read instead of once per pair, and
// two reads share a wavefront.  The older one-pair-per-wavefront kernel below
// serves longer reads and the fp64 redo pass.
#include <cmath>
#include <mutex>
#include <vector>
#include "gbx_internal.h"
#include "phmm_split.h"

namespace gbx {
namespace {

constexpr int QUAL_LIMIT = 128;                               // qualities are masked & 127
constexpr int MM_USED = ((QUAL_LIMIT - 1) * QUAL_LIMIT) / 2 + QUAL_LIMIT;   // entries reachable with quals <= 127
constexpr int NPCLS = 6;                                      // RPL 1,2,3,4,6,8(+tiles)
constexpr int TILED_BLOCKS = 256;                             // wave slots that own a scratch row

// probability tables in device memory (one hipMalloc per device, filled by upload_tables)
struct DevTables {
    const float *ph2pr_f, *mm_f;
    const double *ph2pr_d, *mm_d;
};

__host__ __device__ inline int class_of_rows(int R)
{
    return R <= 64 ? 0 : R <= 128 ? 1 : R <= 192 ? 2 : R <= 256 ? 3 : R <= 384 ? 4 : 5;
}

template <typename T> struct Tab;
template <> struct Tab<float> {
    const float *ph, *mmt;
    __device__ explicit Tab(const DevTables &t) : ph(t.ph2pr_f), mmt(t.mm_f) {}
    __device__ float ph2pr(int x) const { return ph[x]; }
    __device__ float mm(int x) const { return mmt[x]; }
    __device__ static float init() { return ldexpf(1.f, 120); }
};
template <> struct Tab<double> {
    const double *ph, *mmt;
    __device__ explicit Tab(const DevTables &t) : ph(t.ph2pr_d), mmt(t.mm_d) {}
    __device__ double ph2pr(int x) const { return ph[x]; }
    __device__ double mm(int x) const { return mmt[x]; }
    __device__ static double init() { return ldexp(1.0, 1020); }
};

struct PhmmArgs {
    const int32_t *pair_read, *pair_hap;
    const int64_t *read_off; const int32_t *read_len;
    const uint8_t *rs, *q, *qi, *qd, *qc;
    const int64_t *hap_off; const int32_t *hap_len; const uint8_t *hap;
    double *out;
    DevTables tab;
};

struct ReadPart;
struct PhmmWork {
    int32_t *counts;     // [8]  pairs per row class
    int32_t *cursors;    // [8]
    int32_t *next;       // [8]  work cursors: one per class, [7] = fp64 pass
    int32_t *dcount;     // [1]  length of the fp64 redo list
    int32_t *order;      // [n_pairs] pairs binned by class
    int32_t *dlist;      // [n_pairs] pairs to redo in fp64
    char *scratch;       // TILED_BLOCKS * scratch_stride bytes of tile boundary rows
    int64_t scratch_stride;
    // stream path (reads of <= stream_rows rows), all indexed by read id unless noted
    int32_t stream_rows; // STREAM_MAX_ROWS, or 0 for small jobs: every pair takes the one-pair-per-wavefront kernels
    int32_t seg_max;     // pairs per unit of the stream path (SEG_MAX_PAIRS; fewer for jobs too small to fill the chip with long units)
    int32_t lut;         // 1: the stream holds symbol codes (phmm_sym_code) and the stream kernels take the priors from LDS tables
    int64_t n_reads;
    int32_t *rcount;     // pairs of the read
    int32_t *rslen;      // stream symbols of the read: sum (H+1)
    int32_t *rcur;       // scatter cursor
    int32_t *rfirst;     // first position of the read's pairs in porder[]
    int64_t *rsbase;     // byte offset of the read's stream
    int32_t *porder;     // [n_pairs] pairs grouped by read
    int64_t *soff;       // [n_pairs] stream offset of the pair's boundary byte (grouped order)
    float *yin;          // [n_pairs] INITIAL_CONSTANT / haplen (grouped order)
    float *tmp;          // [n_pairs] fp32 sums (grouped order)
    struct ReadPart *part;   // [ceil(n_reads/1024)] block sums, then block offsets, of the read scan
    int32_t *ucount;     // [UBINS] units per (row class, stream-length bucket); ucur = ucount + UBINS: scatter cursors
    int32_t *ucur;
    int32_t *ubase;      // [UBINS+1]
    int32_t *ulist;      // [n_pairs] units = first grouped pair of a (read, segment), class-major, long streams first
    uint8_t *stream;
};

constexpr int STREAM_LANES = 31;                              // row lanes per half-wavefront (lane 31 / 63 stays all-zero)
constexpr int STREAM_MAX_ROWS = STREAM_LANES * 8;
constexpr int SEG_MAX_PAIRS = 8;                              // a read's pairs are cut into units of at most this many
constexpr int UBUCKETS = 32;
constexpr int UBINS = 8 * UBUCKETS;
__host__ __device__ inline int unit_bin(int R, int slen)
{
    const int cls = (R - 1) / STREAM_LANES;                   // rows per lane - 1
    const int b = slen >> 9;
    return cls * UBUCKETS + (UBUCKETS - 1 - (b < UBUCKETS - 1 ? b : UBUCKETS - 1));
}
// unit = (read, segment): pairs [seg*q, min(c, seg*q+q)) of the read's c grouped pairs
__host__ __device__ inline int seg_count(int c, int seg_max) { return (c + seg_max - 1) / seg_max; }
__host__ __device__ inline int seg_pairs(int c, int seg_max) { const int ns = seg_count(c, seg_max); return (c + ns - 1) / ns; }
__host__ __device__ inline int64_t stream_bytes_of(int slen) { return ((int64_t)slen + 1 + 15 + 16) & ~(int64_t)15; }

__device__ inline float shr1(float fill, float x)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(x), 0x138, 0xf, 0xf, false));
}
__device__ inline double shr1(double fill, double x)
{
    const long long xi = __double_as_longlong(x), fi = __double_as_longlong(fill);
    const int lo = __builtin_amdgcn_update_dpp((int)fi, (int)xi, 0x138, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp((int)(fi >> 32), (int)(xi >> 32), 0x138, 0xf, 0xf, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
__device__ inline int shr1(int fill, int x) { return __builtin_amdgcn_update_dpp(fill, x, 0x138, 0xf, 0xf, false); }
// zero for lane 0: with bound_ctrl the shift needs no register holding the fill, and the compiler may fold it into the
// instruction that uses the value (a DPP operand) instead of a move
#ifndef GBX_PHMM_DPP_BC
#define GBX_PHMM_DPP_BC 1            // 0: the round-5 form (a register holds the fill), for A/B builds
#endif
__device__ inline float shr1z(float x) { return GBX_PHMM_DPP_BC ? __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x138, 0xf, 0xf, true)) : shr1(0.f, x); }
__device__ inline int shr1z(int x) { return GBX_PHMM_DPP_BC ? __builtin_amdgcn_update_dpp(0, x, 0x138, 0xf, 0xf, true) : shr1(0, x); }

__device__ inline float fmaT(float a, float b, float c) { return fmaf(a, b, c); }
__device__ inline double fmaT(double a, double b, double c) { return fma(a, b, c); }
__device__ inline float readlaneT(float x, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), l)); }
__device__ inline double readlaneT(double x, int l)
{
    const long long xi = __double_as_longlong(x);
    const int lo = __builtin_amdgcn_readlane((int)xi, l), hi = __builtin_amdgcn_readlane((int)(xi >> 32), l);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}

__global__ void __launch_bounds__(256) phmm_classify_kernel(PhmmArgs A, int64_t n, PhmmWork W, int pass)
{
    __shared__ int lcount[NPCLS];
    __shared__ int lbase[NPCLS];
    const int tid = threadIdx.x;
    if (tid < NPCLS) lcount[tid] = 0;
    __syncthreads();
    const int64_t k = (int64_t)blockIdx.x * 256 + tid;
    int cls = -1, slot = 0;
    if (k < n) {
        const int R = A.read_len[A.pair_read[k]], H = A.hap_len[A.pair_hap[k]];
        if (R <= 0 || H <= 0) {
            // Degenerate pairs never reach the reference kernel from its driver (token parsing
            // cannot produce an empty string).  Empty read: the result row is DP row 0 (M=X=0),
            // log10(0) = -inf; empty haplotype: INITIAL_CONSTANT/0 -> treat the same way.
            if (pass == 0) A.out[k] = -HUGE_VAL;
        } else if (R <= W.stream_rows) {
            const int rd = A.pair_read[k];
            if (pass == 0) { atomicAdd(&W.rcount[rd], 1); atomicAdd(&W.rslen[rd], H + 1); }
            else W.porder[W.rfirst[rd] + atomicAdd(&W.rcur[rd], 1)] = (int)k;
        } else {
            cls = max(class_of_rows(R), 3);       // kernels exist for classes 3..5; shorter reads (small jobs) fit class 3
            slot = atomicAdd(&lcount[cls], 1);
        }
    }
    __syncthreads();
    if (pass == 0) {
        if (tid < NPCLS && lcount[tid]) atomicAdd(&W.counts[tid], lcount[tid]);
        return;
    }
    if (tid < NPCLS) {
        int base = 0;
        for (int c = 0; c < tid; ++c) base += W.counts[c];
        lbase[tid] = lcount[tid] ? base + atomicAdd(&W.cursors[tid], lcount[tid]) : 0;
    }
    __syncthreads();
    if (cls >= 0) W.order[lbase[cls] + slot] = (int)k;
}

// One pair on one wavefront: returns sum_c (M[R][c] + X[R][c]) (scaled by INITIAL_CONSTANT), wave-uniform.
template <typename T, int RPL>
__device__ T phmm_pair(const PhmmArgs &A, int pair, T *scr0, T *scr1)
{
    constexpr int TILE = 64 * RPL;
    const int lane = threadIdx.x & 63;
    const int rd = A.pair_read[pair], hp = A.pair_hap[pair];
    const int R = A.read_len[rd], H = A.hap_len[hp];
    const int64_t ro = A.read_off[rd];
    const uint8_t *hap = A.hap + A.hap_off[hp];
    const int ntiles = (R + TILE - 1) / TILE;
    const int rows0 = R - (ntiles - 1) * TILE;              // tile 0 takes the remainder, later tiles are full
    const T zero = (T)0, one = (T)1;
    const Tab<T> tab(A.tab);
    const T yinit = Tab<T>::init() / (T)H;                   // Y[0][c], every column
    T result = zero;

    for (int tile = 0; tile < ntiles; ++tile) {
        const int rows = tile == 0 ? rows0 : TILE;
        const int row_base = tile == 0 ? 0 : rows0 + (tile - 1) * TILE;     // first read row of the tile
        const int lanes_used = (rows + RPL - 1) / RPL;
        const int pad = lanes_used * RPL - rows;             // leading dummy slots (tile 0 only, < RPL)
        const int llast = lanes_used - 1;
        const int sig_last = lanes_used * RPL - 1;           // slot of the tile's last row
        const bool first = tile == 0, last = tile == ntiles - 1;
        const T *top = (tile & 1) ? scr0 : scr1;             // bottom DP row of the previous tile: M|X|Y planes of H+1
        T *bot = (tile & 1) ? scr1 : scr0;

        T pMM[RPL], pGap[RPL], pMX[RPL], pXX[RPL], pMY[RPL], pYY[RPL], pm[RPL], px[RPL];
        int rch[RPL], hc[RPL];
        T cM[RPL], cX[RPL], cY[RPL], vM[RPL], vX[RPL], vY[RPL];      // c = step s-1, v = step s-2
#pragma unroll
        for (int k = 0; k < RPL; ++k) {
            const int sig = lane * RPL + k;
            const bool real = sig >= pad && sig < pad + rows;
            const int r = row_base + sig - pad;
            // a dummy slot of tile 0 reproduces DP row 0: M=0, X=0, Y=INIT/H at every column
            pMM[k] = zero; pGap[k] = zero; pMX[k] = zero; pXX[k] = zero; pMY[k] = zero; pYY[k] = one;
            pm[k] = zero; px[k] = zero; rch[k] = 0;
            // DP row 0 holds Y = INIT/H from DP column 0 on and nothing before it: dummy slot sig shows that
            // value from step sig on (switched on in the step loop), so that the real rows below keep
            // computing exact zeros until their own first column arrives.
            T y0 = (sig < pad && sig == 0) ? yinit : zero;
            if (real) {
                const int _i = A.qi[ro + r] & 127, _d = A.qd[ro + r] & 127, _c = A.qc[ro + r] & 127;
                const int _q = A.q[ro + r] & 127;
                const int mn = min(_i, _d), mx = max(_i, _d);
                pMM[k] = tab.mm(((mx * (mx + 1)) >> 1) + mn);
                pGap[k] = one - tab.ph2pr(_c);
                pMX[k] = tab.ph2pr(_i); pXX[k] = tab.ph2pr(_c);
                pMY[k] = tab.ph2pr(_d); pYY[k] = tab.ph2pr(_c);
                const T e = tab.ph2pr(_q);
                rch[k] = A.rs[ro + r];
                pm[k] = one - e;                                         // prior when the bases match
                px[k] = rch[k] == 'N' ? one - e : e / (T)3;              // otherwise ('N' always matches)
                y0 = zero;
            }
            cM[k] = zero; cX[k] = zero; cY[k] = y0;                      // DP column 0
            vM[k] = zero; vX[k] = zero; vY[k] = zero;
            hc[k] = 0;
        }
        const int steps = lanes_used * RPL + H - 1;
        // row above the tile as seen by lane 0 / slot 0: `b*` = its column s+1 (DP index), `sd*` = column s
        // (only lane 0 sits under the tile boundary; every other lane's slot 0 starts with DP column 0 = zeros,
        //  so that rows which have not started yet keep computing exact zeros)
        T bM = zero, bX = zero, bY = first ? yinit : zero;
        T sdM = zero, sdX = zero, sdY = (first && lane == 0) ? yinit : zero;   // DP column 0 of the row above
        if (!first) { bM = top[1]; bX = top[(H + 1) + 1]; bY = top[2 * (H + 1) + 1]; }
        int hcur = hap[0], hnxt = hap[min(1, H - 1)];
        T acc = zero;

        for (int s = 0; s < steps; ++s) {
            if (pad > 1) {                                               // switch dummy slot s on (lane 0 only)
#pragma unroll
                for (int k = 1; k < RPL - 1; ++k)
                    if (k < pad && s == k) cY[k] = lane == 0 ? yinit : cY[k];
            }
            // values of the slot above slot 0 (previous lane's bottom slot, or the tile boundary in lane 0)
            T nM = shr1(zero, cM[RPL - 1]);
            T nX = shr1(zero, cX[RPL - 1]);
            T nY = shr1(zero, cY[RPL - 1]);
            int nh = shr1(0, hc[RPL - 1]);
            if (lane == 0) { nM = bM; nX = bX; nY = bY; nh = hcur; }
            hcur = hnxt;
            hnxt = hap[min(s + 2, H - 1)];
            if (!first) {
                const int c2 = min(s + 2, H);
                bM = top[c2]; bX = top[(H + 1) + c2]; bY = top[2 * (H + 1) + c2];
            }
            T newM = zero, newX = zero, newY = zero;
#pragma unroll
            for (int k = RPL - 1; k >= 0; --k) {
                constexpr int Z = 0;
                const int ka = k ? k - 1 : Z;                                       // slot above, inside the lane
                const T aM = k ? cM[ka] : nM, aX = k ? cX[ka] : nX;                 // (r-1, c)
                const T dM = k ? vM[ka] : sdM, dX = k ? vX[ka] : sdX, dY = k ? vY[ka] : sdY;   // (r-1, c-1)
                const int h = k ? hc[ka] : nh;
                const T distm = (h == rch[k] || h == 'N') ? pm[k] : px[k];
                const T m = distm * fmaT(dX + dY, pGap[k], dM * pMM[k]);
                const T x = fmaT(aM, pMX[k], aX * pXX[k]);
                const T y = fmaT(cM[k], pMY[k], cY[k] * pYY[k]);                   // (r, c-1)
                vM[k] = cM[k]; vX[k] = cX[k]; vY[k] = cY[k];
                cM[k] = m; cX[k] = x; cY[k] = y;
                hc[k] = h;
                if (k == RPL - 1) { newM = m; newX = x; newY = y; }
            }
            sdM = nM; sdX = nX; sdY = nY;
            // the tile's last row: accumulate the answer, or hand the row to the next tile
            const int g = s - sig_last;                                  // its haplotype column
            if (g >= 0 && g < H) {
                if (last) acc += (lane == llast) ? newM + newX : zero;
                else if (lane == llast) { bot[g + 1] = newM; bot[(H + 1) + g + 1] = newX; bot[2 * (H + 1) + g + 1] = newY; }
            }
        }
        if (last) result = readlaneT(acc, llast);
        else {
            if (lane == 0) { bot[0] = zero; bot[H + 1] = zero; bot[2 * (H + 1)] = zero; }   // DP column 0
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
    }
    return result;
}

// fp32 pass over one row class; pairs whose result is below MIN_ACCEPTED are queued for the fp64 pass
template <int RPL>
__global__ void __launch_bounds__(64) phmm_f32_kernel(PhmmArgs A, PhmmWork W, int cls)
{
    int cnt = W.counts[cls], first = 0;
    for (int c = 0; c < cls; ++c) first += W.counts[c];
    const int32_t *order = W.order + first;
    const int lane = threadIdx.x;
    float *scr = (float *)(W.scratch + (int64_t)(blockIdx.x % TILED_BLOCKS) * W.scratch_stride);
    float *scr1 = scr + W.scratch_stride / (2 * sizeof(float));
    const float log_init = log10f(ldexpf(1.f, 120));
    for (int slot = blockIdx.x; slot < cnt; slot += gridDim.x) {
        const int pair = order[slot];
        const float r = phmm_pair<float, RPL>(A, pair, scr, scr1);
        if (lane == 0) {
            if (r < 1e-28f) W.dlist[atomicAdd(W.dcount, 1)] = pair;               // MIN_ACCEPTED
            else A.out[pair] = (double)(log10f(r) - log_init);
        }
    }
}

template <int RPL>
__global__ void __launch_bounds__(64) phmm_f64_kernel(PhmmArgs A, PhmmWork W)
{
    const int cnt = *W.dcount;
    const int lane = threadIdx.x;
    double *scr = (double *)(W.scratch + (int64_t)(blockIdx.x % TILED_BLOCKS) * W.scratch_stride);
    double *scr1 = scr + W.scratch_stride / (2 * sizeof(double));
    const double log_init = log10(ldexp(1.0, 1020));
    for (int slot = blockIdx.x; slot < cnt; slot += gridDim.x) {
        const int pair = W.dlist[slot];
        const double r = phmm_pair<double, RPL>(A, pair, scr, scr1);
        if (lane == 0) A.out[pair] = log10(r) - log_init;
    }
}

// ---- stream path ------------------------------------------------------------
// Exclusive scans over the reads (pair positions, stream byte offsets) and the unit list, binned by (row
// class, stream length) with the longest streams first: block sums -> one-block scan -> placement.
constexpr int SCAN_THREADS = 1024;
struct ReadPart { long long bytes; int pairs; int pad; };

__device__ inline void read_contrib(const PhmmWork &W, int64_t r, int &c, long long &bytes)
{
    c = r < W.n_reads ? W.rcount[r] : 0;
    bytes = c ? stream_bytes_of(W.rslen[r]) : 0;
}

__global__ void __launch_bounds__(SCAN_THREADS) phmm_read_sum_kernel(PhmmArgs A, PhmmWork W)
{
    __shared__ long long sb[SCAN_THREADS / 64];
    __shared__ int sc[SCAN_THREADS / 64];
    __shared__ int bins[UBINS];
    const int tid = threadIdx.x;
    for (int b = tid; b < UBINS; b += SCAN_THREADS) bins[b] = 0;
    __syncthreads();
    const int64_t r = (int64_t)blockIdx.x * SCAN_THREADS + tid;
    int c; long long bytes;
    read_contrib(W, r, c, bytes);
    if (c) { const int ns = seg_count(c, W.seg_max); atomicAdd(&bins[unit_bin(A.read_len[r], W.rslen[r] / ns)], ns); }
    long long wb = bytes; int wc = c;
    for (int d = 32; d; d >>= 1) { wb += __shfl_down(wb, d); wc += __shfl_down(wc, d); }
    if ((tid & 63) == 0) { sb[tid >> 6] = wb; sc[tid >> 6] = wc; }
    __syncthreads();
    if (tid == 0) {
        long long tb = 0; int tc = 0;
        for (int k = 0; k < SCAN_THREADS / 64; ++k) { tb += sb[k]; tc += sc[k]; }
        W.part[blockIdx.x].bytes = tb; W.part[blockIdx.x].pairs = tc;
    }
    for (int b = tid; b < UBINS; b += SCAN_THREADS)
        if (bins[b]) atomicAdd(&W.ucount[b], bins[b]);
}

__global__ void __launch_bounds__(SCAN_THREADS) phmm_read_scan_kernel(PhmmWork W, int nblk)
{
    __shared__ long long sb[SCAN_THREADS];
    __shared__ int sc[SCAN_THREADS];
    const int tid = threadIdx.x;
    long long base_b = 0; int base_c = 0;
    for (int k0 = 0; k0 < nblk; k0 += SCAN_THREADS) {
        const int k = k0 + tid;
        const long long vb = k < nblk ? W.part[k].bytes : 0;
        const int vc = k < nblk ? W.part[k].pairs : 0;
        sb[tid] = vb; sc[tid] = vc;
        __syncthreads();
        for (int d = 1; d < SCAN_THREADS; d <<= 1) {
            const long long ab = tid >= d ? sb[tid - d] : 0;
            const int ac = tid >= d ? sc[tid - d] : 0;
            __syncthreads();
            sb[tid] += ab; sc[tid] += ac;
            __syncthreads();
        }
        if (k < nblk) { W.part[k].bytes = base_b + sb[tid] - vb; W.part[k].pairs = base_c + sc[tid] - vc; }
        base_b += sb[SCAN_THREADS - 1]; base_c += sc[SCAN_THREADS - 1];
        __syncthreads();
    }
    if (tid == 0) {
        W.next[0] = base_c;                                    // pairs on the stream path
        int acc = 0;
        for (int b = 0; b < UBINS; ++b) { W.ubase[b] = acc; acc += W.ucount[b]; }
        W.ubase[UBINS] = acc;
    }
}

__global__ void __launch_bounds__(SCAN_THREADS) phmm_read_place_kernel(PhmmArgs A, PhmmWork W)
{
    __shared__ long long sb[SCAN_THREADS / 64];
    __shared__ int sc[SCAN_THREADS / 64];
    __shared__ int lbin[UBINS], lbase[UBINS];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int b = tid; b < UBINS; b += SCAN_THREADS) lbin[b] = 0;
    const int64_t r = (int64_t)blockIdx.x * SCAN_THREADS + tid;
    int c; long long bytes;
    read_contrib(W, r, c, bytes);
    long long ib = bytes; int ic = c;                          // inclusive scan inside the wavefront
    for (int d = 1; d < 64; d <<= 1) {
        const long long ub = __shfl_up(ib, d); const int uc = __shfl_up(ic, d);
        if (lane >= d) { ib += ub; ic += uc; }
    }
    if (lane == 63) { sb[wv] = ib; sc[wv] = ic; }
    __syncthreads();
    long long off_b = W.part[blockIdx.x].bytes; int off_c = W.part[blockIdx.x].pairs;
    for (int k = 0; k < wv; ++k) { off_b += sb[k]; off_c += sc[k]; }
    const int poff = off_c + ic - c;
    if (r < W.n_reads) { W.rfirst[r] = poff; W.rsbase[r] = off_b + ib - bytes; }
    // units: block-aggregated allocation inside the (class, length) bins
    const int ns = c ? seg_count(c, W.seg_max) : 0;
    const int bin = c ? unit_bin(A.read_len[r], W.rslen[r] / ns) : 0;
    const int local = c ? atomicAdd(&lbin[bin], ns) : 0;
    __syncthreads();
    for (int b = tid; b < UBINS; b += SCAN_THREADS) lbase[b] = lbin[b] ? W.ubase[b] + atomicAdd(&W.ucur[b], lbin[b]) : 0;
    __syncthreads();
    if (c) {
        const int q = seg_pairs(c, W.seg_max), at = lbase[bin] + local;
        for (int g = 0; g < ns; ++g) W.ulist[at + g] = poff + g * q;         // first grouped pair of the unit
    }
}

// One thread per read: stream offsets and Y[0][*] = INITIAL_CONSTANT / haplen of its pairs, the closing
// boundary byte and the padding (boundary bytes as well).
__global__ void __launch_bounds__(256) phmm_unit_walk_kernel(PhmmArgs A, PhmmWork W)
{
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= W.n_reads) return;
    const int c = W.rcount[r];
    if (!c) return;
    const int first = W.rfirst[r];
    int64_t off = W.rsbase[r];
    const float init = ldexpf(1.f, 120);
    for (int j = first; j < first + c; ++j) {
        const int H = A.hap_len[A.pair_hap[W.porder[j]]];
        W.soff[j] = off;
        W.yin[j] = init / (float)H;
        off += H + 1;
    }
    const int64_t end = W.rsbase[r] + stream_bytes_of(W.rslen[r]);
    for (; off < end; ++off) W.stream[off] = 0;
}

// Symbol codes of the stream in its table form (PhmmWork::lut): 0 = boundary, 1..5 = A C T G N.  The priors of a cell depend on
// the haplotype symbol only through "equal to the read's base, or one of them N" (PairHMMUnitTest.cpp's scalar semantics, literal
// bytes); with the five symbols the reference's tables know (pairhmm_common.h:34-38) that is a table of six entries per read row,
// built once per unit.  A haplotype with any other byte cannot be coded: its pair gets Y[0][*] = 0, so that its fp32 sum is an
// exact zero and the pair goes to the fp64 pass, which compares literal bytes.
constexpr int PHMM_SYMS = 6;                                 // 0 = boundary, 1..5 = A C T G N (phmm_sym_char)
// Four bytes at a time: bits 1-3 of the five letters are distinct (A 0, C 1, T 2, G 3, N 7), so they select the code and - to tell a
// letter from any other byte with the same three bits - the letter itself out of two eight-byte tables (v_perm_b32).
__device__ constexpr int phmm_sym_char(int code) { return code == 1 ? 'A' : code == 2 ? 'C' : code == 3 ? 'T' : code == 4 ? 'G' : code == 5 ? 'N' : -1; }
__device__ inline uint32_t phmm_sym_code4(uint32_t w, bool &bad)
{
    const uint32_t x = (w >> 1) & 0x07070707u;
    const uint32_t code = __builtin_amdgcn_perm(0x05010101u, 0x04030201u, x);              // entries 7..4 | 3..0
    const uint32_t canon = __builtin_amdgcn_perm(0x4e000000u, 0x47544341u, x);             // 'N' 0 0 0 | 'G' 'T' 'C' 'A'
    bad |= canon != w;
    return code;
}
__device__ inline uint32_t phmm_sym_code(uint32_t c, bool &bad)
{
    bool b4 = false;
    const uint32_t k = phmm_sym_code4(c | 0x41414100u, b4) & 0xff;                          // (the upper bytes: 'A's)
    bad |= b4;
    return k;
}

// One wavefront per grouped pair: boundary byte + haplotype bytes (or their symbol codes) into the read's stream.  The body is
// copied as aligned 4-byte words of the destination; a source word straddles two aligned source words and
// is put together with v_alignbyte (the haplotype arena must be readable a few bytes past its end).
__global__ void __launch_bounds__(256) phmm_stream_copy_kernel(PhmmArgs A, PhmmWork W)
{
    const int lane = threadIdx.x & 63;
    const int64_t n = W.next[0];
    const bool lut = W.lut != 0;
    for (int64_t j = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); j < n; j += (int64_t)gridDim.x * 4) {
        const int hp = A.pair_hap[W.porder[j]];
        const int H = A.hap_len[hp];
        const uint8_t *src = A.hap + A.hap_off[hp];
        uint8_t *dst = W.stream + W.soff[j];
        bool bad = false;
        if (lane == 0) dst[0] = 0;
        ++dst;                                                  // haplotype bytes follow the boundary byte
        const int head = min(H, (int)((4 - ((uintptr_t)dst & 3)) & 3));
        if (lane < head) dst[lane] = lut ? (uint8_t)phmm_sym_code(src[lane], bad) : src[lane];
        const int nw = (H - head) >> 2;                         // aligned destination words
        const uint8_t *sb = src + head;
        const int m = (int)((uintptr_t)sb & 3);
        const uint32_t *s32 = (const uint32_t *)(sb - m);
        uint32_t *d32 = (uint32_t *)(dst + head);
        for (int t = lane; t < nw; t += 64) {
            const uint32_t lo = s32[t], hi = s32[t + 1];
            const uint32_t w = m ? __builtin_amdgcn_alignbyte(hi, lo, (unsigned)m) : lo;
            d32[t] = lut ? phmm_sym_code4(w, bad) : w;
        }
        const int done = head + 4 * nw;
        if (lane < H - done) dst[done + lane] = lut ? (uint8_t)phmm_sym_code(src[done + lane], bad) : src[done + lane];
        if (lut && __any(bad) && lane == 0) W.yin[j] = 0.f;
    }
}

// Two units per wavefront (lanes 0-30 / 32-62 hold the rows, lanes 31 / 63 stay zero so that nothing leaks
// from one half into the other through wave_shr), RPL rows per lane, lane l one column behind lane l-1.
// State of a slot = its (M, X, Y) of the previous column; two copies used alternately (even steps read
// P and write N, odd steps the reverse), so that "previous column" (left), "previous column of the row
// above" (diagonal, same copy) and "this column of the row above" (up, the copy being written) are all
// addressable without register moves.  Across lanes the row above is the previous lane's last slot: its
// previous-step value is this lane's `up`, its value two steps ago the diagonal (both wave_shr:1).
// A 0 byte is DP column 0 of the next haplotype: every slot that meets it resets to zero, the lane that
// owns the last read row emits the finished sum.  Unused slots below the last row (same lane) copy M+X
// downwards (pMX = pXX = 1), so the sum is always read from slot RPL-1 of that lane.
//
// LUT (round 6): the prior of a cell - `(h == base || h == 'N' || base == 'N') ? 1 - e : e / 3` - was a compare and a select per
// cell, two of the ten instructions of a cell and both of the half-rate kind (profiles/valu_peak.json: 4.2 against 2.25 SIMD
// cycles).  With the stream in symbol codes the lane looks its RPL priors up instead: a table of PHMM_SYMS x RPL floats per lane
// in LDS (entry (sym, k) of lane l at word (sym * RPL + k) * 64 + l: every lane its own bank whatever the symbols), written once
// per unit, read with one address (`sym * RPL * 256 + 4 * lane`, a v_mad) and RPL immediate offsets per step.  7.5 KB a wavefront
// at RPL = 5: the sixteen wavefronts per CU the kernel's registers allow fit the CU's 160 KB.
#ifndef GBX_PHMM_SFEED
#define GBX_PHMM_SFEED 1            // 0: the table form takes its symbols from vector registers like the other (A/B builds)
#endif
#ifndef GBX_PHMM_CARRY
#define GBX_PHMM_CARRY 0            // 1: the table form carries the diagonal M, X over from the step before (its uM, uX) instead of shifting them in again: one instruction MORE per step - the shifts it saves were operand modifiers of a multiply, the carried values need moves of their own
#endif
template <int RPL, bool LUT>
__global__ void __launch_bounds__(64) phmm_stream_kernel(PhmmArgs A, PhmmWork W, int cls)
{
    constexpr bool SFEED = LUT && GBX_PHMM_SFEED, CARRY = LUT && GBX_PHMM_CARRY;
    __shared__ float lut[LUT ? PHMM_SYMS * RPL * 64 : 1];
    const int lane = threadIdx.x, hl = lane & 31, half = lane >> 5;
    const int ufirst = W.ubase[cls * UBUCKETS];
    const int ucnt = W.ubase[(cls + 1) * UBUCKETS] - ufirst;
    const int32_t *ulist = W.ulist + ufirst;
    const Tab<float> tab(A.tab);
    const float zero = 0.f, one = 1.f;
    const bool top = hl == 0;

    // Unit pairs are drawn from a cursor (the class's, zeroed with the workspace header), not strided over the grid: the seven
    // kernels of a launch set become ready together, and when workgroups of a smaller class land on a CU first, some of this
    // class's cannot become resident until those leave - with a fixed share of the units such a latecomer ends that much later
    // (the launch set's slow mode: 145 against 175 ms for the same host call, profiles/r06zza_phmm_host_bimodal.txt); drawn from
    // the cursor - longest streams first - it simply takes fewer.
    // (a workgroup's first pair is its own index - a launch of a class without units, or with fewer pairs than workgroups, ends
    // without touching the cursor: eight kernels x 5 000 workgroups of atomics on one cache line were 0.4 ms of a small job -, the
    // cursor hands out the pairs from gridDim.x on)
    int32_t *const ucursor = W.counts + 32 + cls;
    for (int slot = blockIdx.x; 2 * slot < ucnt;) {
        const int ui = 2 * slot + half;
        const bool have = ui < ucnt;
        // the unit: `cnt` grouped pairs from `first` on, all of read `rd`
        const int first = have ? ulist[ui] : 0;
        const int rd = have ? A.pair_read[W.porder[first]] : 0;
        const int R = have ? A.read_len[rd] : 0;
        int cnt = 0;
        int64_t s_beg = 0, s_end = 0;                           // stream bytes [s_beg, s_end]: s_end = closing boundary
        if (have) {
            const int c = W.rcount[rd], r0 = W.rfirst[rd];
            cnt = min(seg_pairs(c, W.seg_max), c - (first - r0));
            s_beg = W.soff[first];
            s_end = first + cnt < r0 + c ? W.soff[first + cnt] : W.rsbase[rd] + W.rslen[rd];
        }
        const int slen = (int)(s_end - s_beg) + 1;              // symbols including the closing boundary
        // 4-byte aligned window over the stream: symbol p of the unit is byte p + skew of the words from wp on
        const int skew = (int)(s_beg & 3);
        const uint32_t *wp = (const uint32_t *)(W.stream + (s_beg - skew));
        const int64_t ro = A.read_off[rd];
        const int llast = R > 0 ? (R - 1) / RPL : 0;
        const int my_steps = have ? slen + llast : 0;
        const int steps = max(__builtin_amdgcn_readlane(my_steps, 0), __builtin_amdgcn_readlane(my_steps, 32));
        const bool is_last = have && hl == llast;

        float pMM[RPL], pGap[RPL], pMX[RPL], pXX[RPL], pMY[RPL], pYY[RPL], pm[RPL], px[RPL];
        int rch[RPL];
        float S0[3][RPL], S1[3][RPL];
#pragma unroll
        for (int k = 0; k < RPL; ++k) {
            const int r = hl * RPL + k;
            const float cp = is_last ? one : zero;               // copy slot: x = up.M + up.X; matches nothing
            pMM[k] = zero; pGap[k] = zero; pMX[k] = cp; pXX[k] = cp; pMY[k] = zero; pYY[k] = zero;
            pm[k] = zero; px[k] = zero; rch[k] = 0x100;
            if (r < R && hl < STREAM_LANES) {
                const int _i = A.qi[ro + r] & 127, _d = A.qd[ro + r] & 127, _c = A.qc[ro + r] & 127;
                const int _q = A.q[ro + r] & 127;
                const int mn = min(_i, _d), mx = max(_i, _d);
                pMM[k] = tab.mm(((mx * (mx + 1)) >> 1) + mn);
                pGap[k] = one - tab.ph2pr(_c);
                pMX[k] = tab.ph2pr(_i); pXX[k] = tab.ph2pr(_c);
                pMY[k] = tab.ph2pr(_d); pYY[k] = tab.ph2pr(_c);
                const float e = tab.ph2pr(_q);
                rch[k] = A.rs[ro + r];
                pm[k] = one - e;
                px[k] = rch[k] == 'N' ? one - e : e / 3.f;
            }
#pragma unroll
            for (int v = 0; v < 3; ++v) { S0[v][k] = zero; S1[v][k] = zero; }
            if (LUT) {
                // (entry 0, the boundary symbol: any finite value - the slots that meet it are reset; px[k] is pm[k] already when the base is N)
#pragma unroll
                for (int y = 0; y < PHMM_SYMS; ++y) lut[(y * RPL + k) * 64 + lane] = (phmm_sym_char(y) == rch[k] || y == 5) ? pm[k] : px[k];
            }
        }
        int kk = 0;                                            // boundaries this lane has met
        float ycur = zero, ynext = W.yin[first];
        float acc = zero;                                      // sum over the columns of the current haplotype (lane llast)
        int h = 1;                                             // symbol of this lane's column (1: nothing yet, matches nothing)
        // symbol words: w0 holds the symbols of this group of 4 steps (after the skew shift), w1/w2 the next words
        uint32_t wa = 0, wb = 0, wc = 0;
        if (!SFEED) { wa = wp[0]; wb = wp[1]; wc = wp[2]; }
        // LUT: the priors are looked up one step ahead (the first of them feeds the chain m -> x of the row below -> ... at the very
        // start of a step): hn = this lane's symbol of the coming step, dnx[] its priors, on their way from LDS during the step before
        // (a symbol travels down the lanes as its table's byte offset, SYM_STEP per code: the look-up address is one OR away, and 0
        // is still the boundary)
        constexpr int SYM_STEP = RPL * 64 * 4;
        int hn = SYM_STEP;
        float dnx[RPL];
        const int lane4 = lane * 4;
        auto lookup = [&](int symoff) {
            const float *lp = (const float *)((const char *)lut + (symoff | lane4));
#pragma unroll
            for (int k = 0; k < RPL; ++k) dnx[k] = lp[k * 64];
        };
        // LUT: the symbols of the two top lanes are fed from scalar registers - the two units' stream words come by scalar loads, the
        // skew shift, the end-of-unit guard, the byte extracts and the scaling are scalar instructions, and a step writes its two
        // symbols into lanes 0 and 32 (v_writelane) - where the other form spends a compare, a select and a share of the word
        // arithmetic per step in every lane for the sake of two.
        typedef const __attribute__((address_space(4))) uint32_t sword_t;
        sword_t *qa = nullptr, *qb = nullptr;
        uint32_t a0 = 0, a1 = 0, a2 = 0, b0 = 0, b1 = 0, b2 = 0, w4a = 0, w4b = 0;
        int skew_a = 0, skew_b = 0, slen_a = 0, slen_b = 0;
        auto uni64 = [](uint64_t v, int l) -> uint64_t {
            return (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l) | (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l) << 32;
        };
        auto group = [](uint32_t lo, uint32_t hi, int sk, int rem) -> uint32_t {       // the four symbols of a group, `rem` of them inside the unit
            const uint32_t w = (uint32_t)(((uint64_t)hi << 32 | lo) >> (8 * sk));
            return rem >= 4 ? w : rem <= 0 ? 0u : w & (0xffffffffu >> (8 * (4 - rem)));
        };
        if (LUT && !SFEED) {
            const uint32_t w4 = skew ? (uint32_t)(((uint64_t)wb << 32 | wa) >> (8 * skew)) : wa;
            if (top) hn = (int)(w4 & 0xff) * SYM_STEP;         // (symbol 0 of a unit is its opening boundary: slen >= 1)
            lookup(hn);
        }
        if (SFEED) {
            qa = (sword_t *)uni64((uint64_t)wp, 0); qb = (sword_t *)uni64((uint64_t)wp, 32);
            skew_a = __builtin_amdgcn_readlane(skew, 0); skew_b = __builtin_amdgcn_readlane(skew, 32);
            slen_a = __builtin_amdgcn_readlane(slen, 0); slen_b = __builtin_amdgcn_readlane(slen, 32);
            a0 = qa[0]; a1 = qa[1]; a2 = qa[2]; b0 = qb[0]; b1 = qb[1]; b2 = qb[2];
            w4a = group(a0, a1, skew_a, slen_a); w4b = group(b0, b1, skew_b, slen_b);
            hn = lane == 0 ? (int)(w4a & 0xff) * SYM_STEP : lane == 32 ? (int)(w4b & 0xff) * SYM_STEP : hn;
            lookup(hn);
        }
        float cM = zero, cX = zero;                            // LUT: uM, uX of the step before = this step's dM, dX (the same array, the same shift)

        // sym_in: the symbol entering the top lane at this step - LUT: sym_in / sym_b enter lanes 0 / 32 at the NEXT step
        auto step = [&](int sym_in, int sym_b, const float (&P)[3][RPL], float (&N)[3][RPL]) {
            // the row above slot 0: previous lane's last slot (this column = its previous step, P; the column
            // before = two steps ago, N before it is overwritten); DP row 0 for the top lane
            const float uM = shr1z(P[0][RPL - 1]), uX = shr1z(P[1][RPL - 1]);
            const float dM = CARRY ? cM : shr1z(N[0][RPL - 1]), dX = CARRY ? cX : shr1z(N[1][RPL - 1]);
            float dY = shr1z(N[2][RPL - 1]);
            float dist[RPL];
            if (LUT) {
                if (CARRY) { cM = uM; cX = uX; }
                h = hn;
#pragma unroll
                for (int k = 0; k < RPL; ++k) dist[k] = dnx[k];
                hn = shr1z(h);
                if (SFEED) {
                    // (v_writelane_b32: one instruction per symbol - a select on a scalar operand would need the operand moved to a
                    // vector register first, the mask being the one scalar operand a VOP3 instruction may read on this chip)
                    asm("s_nop 1\n\tv_writelane_b32 %0, %1, 0\n\tv_writelane_b32 %0, %2, 32\n\ts_nop 0" : "+v"(hn) : "s"(sym_in), "s"(sym_b));
                } else if (top) hn = sym_in * SYM_STEP;
                if (top) dY = ycur;
                lookup(hn);
            } else {
                h = shr1z(h);
                if (top) { dY = ycur; h = sym_in; }
            }
            const bool isb = h == 0;
            const bool hN = !LUT && h == 'N';
#pragma unroll
            for (int k = 0; k < RPL; ++k) {
                const float gM = k ? P[0][k - 1] : dM, gX = k ? P[1][k - 1] : dX, gY = k ? P[2][k - 1] : dY;
                const float aM = k ? N[0][k - 1] : uM, aX = k ? N[1][k - 1] : uX;
                const float distm = LUT ? dist[k] : (h == rch[k] || hN) ? pm[k] : px[k];
                const float m = distm * fmaf(gX + gY, pGap[k], gM * pMM[k]);
                const float x = fmaf(aM, pMX[k], aX * pXX[k]);
                const float y = fmaf(P[0][k], pMY[k], P[2][k] * pYY[k]);
                N[0][k] = m; N[1][k] = x; N[2][k] = y;
            }
            if (__any(isb)) {                                   // some lane is at DP column 0 of a haplotype
                if (isb && is_last && kk >= 1 && kk <= cnt) W.tmp[first + kk - 1] = acc;
                acc = isb ? zero : acc;
                ycur = isb ? ynext : ycur;
                kk += isb ? 1 : 0;
                if (isb) ynext = W.yin[first + max(0, min(kk, cnt - 1))];
#pragma unroll
                for (int k = 0; k < RPL; ++k) {
                    N[0][k] = isb ? zero : N[0][k]; N[1][k] = isb ? zero : N[1][k]; N[2][k] = isb ? zero : N[2][k];
                }
            }
            acc += N[0][RPL - 1] + N[1][RPL - 1];               // only lane llast's sum is ever read
        };
        if (SFEED) {
            for (int s = 0; s < steps; s += 4) {
                // this group's symbols are in w4a / w4b; the next group's are put together now: its first symbol rides in this group's last step
                a0 = a1; a1 = a2; a2 = qa[(s >> 2) + 3];
                b0 = b1; b1 = b2; b2 = qb[(s >> 2) + 3];
                const uint32_t na = group(a0, a1, skew_a, slen_a - s - 4), nb = group(b0, b1, skew_b, slen_b - s - 4);
                step((int)(w4a >> 8 & 0xff) * SYM_STEP, (int)(w4b >> 8 & 0xff) * SYM_STEP, S0, S1);
                __builtin_amdgcn_sched_barrier(0);              // keep the steps apart: interleaving them only costs registers
                step((int)(w4a >> 16 & 0xff) * SYM_STEP, (int)(w4b >> 16 & 0xff) * SYM_STEP, S1, S0);
                __builtin_amdgcn_sched_barrier(0);
                step((int)(w4a >> 24) * SYM_STEP, (int)(w4b >> 24) * SYM_STEP, S0, S1);
                __builtin_amdgcn_sched_barrier(0);
                step((int)(na & 0xff) * SYM_STEP, (int)(nb & 0xff) * SYM_STEP, S1, S0);
                __builtin_amdgcn_sched_barrier(0);
                w4a = na; w4b = nb;
            }
        } else {
        for (int s = 0; s < steps; s += 4) {
            // the 4 symbols entering the top lane at steps s..s+3 (0 = boundary beyond the unit's end)
            const uint32_t w4 = skew ? (uint32_t)(((uint64_t)wb << 32 | wa) >> (8 * skew)) : wa;
            wa = wb; wb = wc;
            wc = wp[(s >> 2) + 3];
            const int s0 = s < slen ? (int)(w4 & 0xff) : 0, s1 = s + 1 < slen ? (int)(w4 >> 8 & 0xff) : 0;
            const int s2 = s + 2 < slen ? (int)(w4 >> 16 & 0xff) : 0, s3 = s + 3 < slen ? (int)(w4 >> 24) : 0;
            int s4 = 0;                                         // LUT: the first symbol of the next group
            if (LUT) {
                const uint32_t w4n = skew ? (uint32_t)(((uint64_t)wb << 32 | wa) >> (8 * skew)) : wa;
                s4 = s + 4 < slen ? (int)(w4n & 0xff) : 0;
            }
            step(LUT ? s1 : s0, 0, S0, S1);
            __builtin_amdgcn_sched_barrier(0);                  // keep the steps apart: interleaving them only costs registers
            step(LUT ? s2 : s1, 0, S1, S0);
            __builtin_amdgcn_sched_barrier(0);
            step(LUT ? s3 : s2, 0, S0, S1);
            __builtin_amdgcn_sched_barrier(0);
            step(LUT ? s4 : s3, 0, S1, S0);
            __builtin_amdgcn_sched_barrier(0);
        }
        }
        // the next pair: from the cursor
        if (lane == 0) slot = (int)gridDim.x + atomicAdd(ucursor, 1);
        slot = __builtin_amdgcn_readfirstlane(slot);
    }
}

// fp32 sums of the stream path -> log10 likelihoods, or the fp64 redo list
__global__ void __launch_bounds__(256) phmm_stream_finish_kernel(PhmmArgs A, PhmmWork W)
{
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= W.next[0]) return;
    const float r = W.tmp[j];
    const int pair = W.porder[j];
    if (r < 1e-28f) W.dlist[atomicAdd(W.dcount, 1)] = pair;                     // MIN_ACCEPTED
    else A.out[pair] = (double)(log10f(r) - log10f(ldexpf(1.f, 120)));
}

// ---- host-side tables: GKL Context<NUMBER>::initializeStaticMembers ----
struct HostTables {
    std::vector<float> ph_f, mm_f;
    std::vector<double> ph_d, mm_d;
};

template <typename NUM>
double approx_log10_sum_log10(double small, double big, const std::vector<NUM> &jac)
{
    const double TOL = 8.0, INV_STEP = 1.0 / 0.0001;
    if (small > big) std::swap(small, big);
    if (std::isinf(small) && small < 0) return big;
    if (std::isinf(big) && big < 0) return big;
    const double diff = big - small;
    if (diff >= TOL) return big;
    const NUM v = (NUM)(diff * INV_STEP);
    const int ind = v > (NUM)0 ? (int)(v + (NUM)0.5) : (int)(v - (NUM)0.5);      // fastRound
    return big + jac[ind];
}

const HostTables &host_tables()
{
    static HostTables t;
    static std::once_flag once;
    std::call_once(once, [] {
        t.ph_f.resize(QUAL_LIMIT); t.ph_d.resize(QUAL_LIMIT);
        for (int x = 0; x < QUAL_LIMIT; ++x) {
            t.ph_f[x] = powf(10.0f, -((float)x) / 10.0f);
            t.ph_d[x] = pow(10.0, -((double)x) / 10.0);
        }
        const int JS = (int)(8.0 / 0.0001) + 1;
        std::vector<double> jd(JS);
        std::vector<float> jf(JS);
        for (int k = 0; k < JS; ++k) {
            jd[k] = log10(1.0 + pow(10.0, -((double)k) * 0.0001));
            jf[k] = (float)jd[k];
        }
        const double inv_ln10 = 1.0 / log(10.0);
        t.mm_f.resize(MM_USED); t.mm_d.resize(MM_USED);
        for (int i = 0, offset = 0; i < QUAL_LIMIT; offset += ++i)
            for (int j = 0; j <= i; ++j) {
                const double lf = approx_log10_sum_log10<float>(-0.1 * i, -0.1 * j, jf);
                const double ld = approx_log10_sum_log10<double>(-0.1 * i, -0.1 * j, jd);
                t.mm_f[offset + j] = (float)pow(10, log1p(-std::min(1.0, pow(10, lf))) * inv_ln10);
                t.mm_d[offset + j] = pow(10, log1p(-std::min(1.0, pow(10, ld))) * inv_ln10);
            }
    });
    return t;
}

int upload_tables(DevTables *out)
{
    static std::mutex mu;
    static std::vector<std::pair<int, DevTables>> done;
    int dev = 0;
    GBX_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(mu);
    for (auto &d : done) if (d.first == dev) { *out = d.second; return GBX_OK; }
    const HostTables &t = host_tables();
    const size_t bf = sizeof(float) * (QUAL_LIMIT + MM_USED), bd = sizeof(double) * (QUAL_LIMIT + MM_USED);
    char *buf = nullptr;
    GBX_HIP(hipMalloc((void **)&buf, bf + bd + 64));
    double *dd = (double *)buf;                    // doubles first (alignment)
    float *df = (float *)(buf + bd);
    GBX_HIP(hipMemcpy(dd, t.ph_d.data(), sizeof(double) * QUAL_LIMIT, hipMemcpyHostToDevice));
    GBX_HIP(hipMemcpy(dd + QUAL_LIMIT, t.mm_d.data(), sizeof(double) * MM_USED, hipMemcpyHostToDevice));
    GBX_HIP(hipMemcpy(df, t.ph_f.data(), sizeof(float) * QUAL_LIMIT, hipMemcpyHostToDevice));
    GBX_HIP(hipMemcpy(df + QUAL_LIMIT, t.mm_f.data(), sizeof(float) * MM_USED, hipMemcpyHostToDevice));
    DevTables d = {df, df + QUAL_LIMIT, dd, dd + QUAL_LIMIT};
    done.emplace_back(dev, d);
    *out = d;
    return GBX_OK;
}

size_t scratch_stride_bytes(int max_hap_len)
{
    // two ping-pong rows of three planes of (H+1) doubles, 64-byte aligned
    size_t b = (size_t)2 * 3 * ((size_t)max_hap_len + 1) * sizeof(double);
    return (b + 63) & ~(size_t)63;
}

}  // namespace

namespace {
// workspace carve-up, shared by the size query and the launch
struct WorkLayout {
    size_t order, dlist, scratch, rcount, rslen, rcur, ucount, rfirst, rsbase, porder, soff, yin, tmp, part, ubase, ulist, stream, total;
};
// stream_syms: sum over the pairs of haplen+1 when the caller knows it (the host entry counts it while it validates
// the pair list), < 0 for the bound n_pairs*(max_hap_len+1) that needs no pass over the pairs.  The stream is the last
// piece of the layout, so every other offset is the same for both.
WorkLayout work_layout(int64_t n_pairs, int64_t n_reads, int max_hap_len, int64_t stream_syms = -1)
{
    if (n_pairs < 0) n_pairs = 0;
    if (n_reads < 0) n_reads = 0;
    if (max_hap_len < 1) max_hap_len = 1;
    WorkLayout L;
    size_t off = 64 * sizeof(int32_t);
    auto take = [&](size_t bytes) { const size_t at = off; off = (off + bytes + 63) & ~(size_t)63; return at; };
    L.order = take((size_t)n_pairs * 4);
    L.dlist = take((size_t)n_pairs * 4);
    L.scratch = take((size_t)TILED_BLOCKS * scratch_stride_bytes(max_hap_len));
    L.rcount = take((size_t)n_reads * 4);          // rcount | rslen | rcur are zeroed together
    L.rslen = take((size_t)n_reads * 4);
    L.rcur = take((size_t)n_reads * 4);
    L.ucount = take((size_t)2 * UBINS * 4);        // + scatter cursors; zeroed with the three arrays above
    L.rfirst = take((size_t)n_reads * 4);
    L.rsbase = take((size_t)n_reads * 8);
    L.porder = take((size_t)n_pairs * 4);
    L.soff = take((size_t)n_pairs * 8);
    L.yin = take((size_t)(n_pairs + 1) * 4);
    L.tmp = take((size_t)(n_pairs + 1) * 4);
    L.part = take((size_t)(n_reads / 1024 + 2) * 16);
    L.ubase = take((size_t)(UBINS + 1) * 4);
    L.ulist = take((size_t)n_pairs * 4);
    // every pair contributes haplen+1 symbols, every read with pairs < 48 bytes of closing boundary + padding
    if (stream_syms < 0) stream_syms = n_pairs * ((int64_t)max_hap_len + 1);
    L.stream = take((size_t)stream_syms + (size_t)(n_pairs < n_reads ? n_pairs : n_reads) * 48 + 64);
    L.total = off;
    return L;
}
}  // namespace

size_t phmm_workspace_bytes(int64_t n_pairs, int64_t n_reads, int max_hap_len, int64_t stream_syms)
{
    return work_layout(n_pairs, n_reads, max_hap_len, stream_syms).total;
}

int phmm_init_tables() { DevTables t; return upload_tables(&t); }

const float *phmm_host_mm_table_f(int *n) { if (n) *n = MM_USED; return host_tables().mm_f.data(); }

static thread_local const std::function<int()> *t_phmm_between = nullptr;
void phmm_set_between(const std::function<int()> *between) { t_phmm_between = between; }

int phmm_launch(int64_t n_pairs, const int32_t *pair_read, const int32_t *pair_hap,
                int64_t n_reads, const int64_t *read_off, const int32_t *read_len,
                const uint8_t *rs, const uint8_t *q, const uint8_t *qi, const uint8_t *qd, const uint8_t *qc,
                const int64_t *hap_off, const int32_t *hap_len, const uint8_t *hap, int max_hap_len,
                double *out, void *d_work, size_t work_bytes, hipStream_t s, int64_t stream_syms)
{
    const std::function<int()> *between = t_phmm_between;      // (phmm_split.h) this launch's, and only this launch's
    t_phmm_between = nullptr;
    if (n_pairs == 0) return GBX_OK;
    if (n_pairs > 0x7fffffffLL - 1024 || n_reads > 0x7fffffffLL - 1024) {
        set_error("phmm: more than 2^31 pairs or reads in one call");
        return GBX_ERR_UNSUPPORTED;
    }
    const WorkLayout L = work_layout(n_pairs, n_reads, max_hap_len, stream_syms);
    if (work_bytes < L.total) { set_error("phmm: workspace too small"); return GBX_ERR_ARG; }
    DevTables tabs;
    int rc = upload_tables(&tabs);
    if (rc) return rc;
    PhmmArgs A = {pair_read, pair_hap, read_off, read_len, rs, q, qi, qd, qc, hap_off, hap_len, hap, out, tabs};
    char *wb = (char *)d_work;
    int32_t *wi = (int32_t *)d_work;
    PhmmWork W;
    W.counts = wi; W.cursors = wi + 8; W.next = wi + 16; W.dcount = wi + 24;
    W.order = (int32_t *)(wb + L.order); W.dlist = (int32_t *)(wb + L.dlist);
    W.scratch = wb + L.scratch;
    W.scratch_stride = (int64_t)scratch_stride_bytes(max_hap_len);
    W.n_reads = n_reads;
    // Small jobs: grouping the pairs by read and laying out the haplotype streams is seven dependent launches
    // before the first cell is computed, and the eight stream classes end in single-unit tails; below a few
    // thousand pairs one pair per wavefront on the tiled kernels (one round of the chip) is faster.  The
    // reference's driver calls per batch of a few hundred pairs (PairHMMUnitTest.cpp:228-245), so this is its path.
    const char *small_env = getenv("GBX_PHMM_SMALL");
    const bool small_job = small_env ? atoi(small_env) != 0 : n_pairs < 12000;      // (with short units the stream path wins from ~12 000 pairs on: same sweep)
    W.stream_rows = small_job ? 0 : STREAM_MAX_ROWS;
    // Units of up to eight pairs keep a half-wavefront busy for a long stream, which is what a job that oversubscribes the chip
    // wants; a job of a few ten thousand pairs (sixty-four of the reference driver's batches combined, host_combine.h) is then a
    // handful of long units per SIMD and as long as its longest one: such jobs take shorter units (GBX_PHMM_SEG overrides)
    {
        const char *se = getenv("GBX_PHMM_SEG");
        // (measured, profiles/r06i_phmm_seg_sweep.txt: 64 batches 2.10 -> 1.64 ms, 32 batches 1.86 -> 1.12 ms with one pair per unit;
        // from about a thousand batches on eight per unit is the faster form again)
        int sm = se ? atoi(se) : n_pairs < 200000 ? 1 : n_pairs < 400000 ? 2 : SEG_MAX_PAIRS;
        W.seg_max = sm < 1 ? 1 : sm > SEG_MAX_PAIRS ? SEG_MAX_PAIRS : sm;
    }
    {
        const char *le = getenv("GBX_PHMM_LUT");                // 0: the compare-and-select form of round 5 (A/B runs)
        W.lut = le ? atoi(le) != 0 : 1;
    }
    W.rcount = (int32_t *)(wb + L.rcount); W.rslen = (int32_t *)(wb + L.rslen); W.rcur = (int32_t *)(wb + L.rcur);
    W.rfirst = (int32_t *)(wb + L.rfirst); W.rsbase = (int64_t *)(wb + L.rsbase);
    W.porder = (int32_t *)(wb + L.porder); W.soff = (int64_t *)(wb + L.soff);
    W.yin = (float *)(wb + L.yin); W.tmp = (float *)(wb + L.tmp);
    W.part = (ReadPart *)(wb + L.part);
    W.ucount = (int32_t *)(wb + L.ucount); W.ucur = W.ucount + UBINS; W.ubase = (int32_t *)(wb + L.ubase); W.ulist = (int32_t *)(wb + L.ulist);
    W.stream = (uint8_t *)(wb + L.stream);
    GBX_HIP(hipMemsetAsync(d_work, 0, 64 * sizeof(int32_t), s));
    GBX_HIP(hipMemsetAsync(wb + L.rcount, 0, L.rfirst - L.rcount, s));
    GBX_HIP(hipMemsetAsync(wb + L.yin, 0, 64, s));             // yin[0] is read by idle half-wavefronts
    const int cb = (int)((n_pairs + 255) / 256);
    const int rb = (int)((n_reads + 255) / 256);
    {
        // group the pairs by read, lay the haplotype streams out
        Stage st("phmm_group", s);
        hipLaunchKernelGGL(phmm_classify_kernel, dim3(cb), dim3(256), 0, s, A, n_pairs, W, 0);
        const int nblk = small_job ? 0 : (int)((n_reads + SCAN_THREADS - 1) / SCAN_THREADS);
        if (nblk) hipLaunchKernelGGL(phmm_read_sum_kernel, dim3(nblk), dim3(SCAN_THREADS), 0, s, A, W);
        if (!small_job) hipLaunchKernelGGL(phmm_read_scan_kernel, dim3(1), dim3(SCAN_THREADS), 0, s, W, nblk);
        if (nblk) hipLaunchKernelGGL(phmm_read_place_kernel, dim3(nblk), dim3(SCAN_THREADS), 0, s, A, W);
        hipLaunchKernelGGL(phmm_classify_kernel, dim3(cb), dim3(256), 0, s, A, n_pairs, W, 1);
        if (rb && !small_job) hipLaunchKernelGGL(phmm_unit_walk_kernel, dim3(rb), dim3(256), 0, s, A, W);
        if (!small_job) {
            int dev_c = 0, cus_c = 256;
            (void)hipGetDevice(&dev_c);
            (void)hipDeviceGetAttribute(&cus_c, hipDeviceAttributeMultiprocessorCount, dev_c);
            const int64_t want = (n_pairs + 3) / 4, cap_c = (int64_t)cus_c * 32;
            hipLaunchKernelGGL(phmm_stream_copy_kernel, dim3((int)(want < cap_c ? want : cap_c)), dim3(256), 0, s, A, W);
        }
    }
    // everything above reads the pair lists, the length tables and the haplotypes; from here on the reads' bases and qualities too
    if (between) { const int brc = (*between)(); if (brc) return brc; }
    int dev_id = 0, cus = 256;
    (void)hipGetDevice(&dev_id);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev_id);
    auto grid = [&](int per_cu) { int64_t cap = (int64_t)cus * per_cu; return (int)(n_pairs < cap ? n_pairs : cap); };
    // the number of grouped pairs is only known on the device (work.next[0]): the copy / finish kernels are
    // launched over n_pairs slots and stop there themselves
    // the row classes are independent and each ends in a tail of single long units: they run side by side
    if (!small_job) {
    SideStreams *ss = nullptr;
    if ((rc = side_streams(&ss))) return rc;
    std::unique_lock<std::mutex> side_lock(ss->mu);
    if ((rc = ss->fork(s))) return rc;
#define GBX_STREAM2(RPL_, LUT_, STREAM_)                                                                                \
    {                                                                                                                   \
        static int per_cu = 0;                                                                                          \
        if (!per_cu) {                                                                                                  \
            int qb = 0;                                                                                                 \
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&qb, phmm_stream_kernel<RPL_, LUT_>, 64, 0) != hipSuccess || qb < 1) { \
                (void)hipGetLastError(); qb = 8;                                                                        \
            }                                                                                                           \
            per_cu = qb > 32 ? 32 : qb;                                                                                 \
        }                                                                                                               \
        Stage st("phmm_stream_rpl" #RPL_, STREAM_);                                                                     \
        hipLaunchKernelGGL((phmm_stream_kernel<RPL_, LUT_>), dim3(grid(per_cu)), dim3(64), 0, STREAM_, A, W, RPL_ - 1); \
    }
#define GBX_STREAM(RPL_, STREAM_) { if (W.lut) GBX_STREAM2(RPL_, true, STREAM_) else GBX_STREAM2(RPL_, false, STREAM_) }
    GBX_STREAM(5, s) GBX_STREAM(4, ss->side[0]) GBX_STREAM(6, ss->side[1]) GBX_STREAM(3, ss->side[2])
    GBX_STREAM(2, ss->side[0]) GBX_STREAM(7, ss->side[1]) GBX_STREAM(1, ss->side[2]) GBX_STREAM(8, ss->side[1])
#undef GBX_STREAM
#undef GBX_STREAM2
    if ((rc = ss->join(s))) return rc;
    side_lock.unlock();
    { Stage st("phmm_stream_finish", s); hipLaunchKernelGGL(phmm_stream_finish_kernel, dim3(cb), dim3(256), 0, s, A, W); }
    }
    // reads longer than STREAM_MAX_ROWS rows: one pair per wavefront, row tiles
    { Stage st("phmm_f32_rpl4", s); hipLaunchKernelGGL(phmm_f32_kernel<4>, dim3(grid(20)), dim3(64), 0, s, A, W, 3); }
    { Stage st("phmm_f32_rpl6", s); hipLaunchKernelGGL(phmm_f32_kernel<6>, dim3(grid(12)), dim3(64), 0, s, A, W, 4); }
    { Stage st("phmm_f32_rpl8", s); hipLaunchKernelGGL(phmm_f32_kernel<8>, dim3(std::min(grid(8), TILED_BLOCKS)), dim3(64), 0, s, A, W, 5); }
    { Stage st("phmm_f64_redo", s); hipLaunchKernelGGL(phmm_f64_kernel<4>, dim3(std::min(grid(8), TILED_BLOCKS)), dim3(64), 0, s, A, W); }
    GBX_HIP(hipGetLastError());
    return GBX_OK;
}

}  // namespace gbx
