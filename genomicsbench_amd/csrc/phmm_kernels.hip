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
#include <cmath>
#include <mutex>
#include <vector>
#include "gbx_internal.h"

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

struct PhmmWork {
    int32_t *counts;     // [8]  pairs per row class
    int32_t *cursors;    // [8]
    int32_t *next;       // [8]  work cursors: one per class, [7] = fp64 pass
    int32_t *dcount;     // [1]  length of the fp64 redo list
    int32_t *order;      // [n_pairs] pairs binned by class
    int32_t *dlist;      // [n_pairs] pairs to redo in fp64
    char *scratch;       // TILED_BLOCKS * scratch_stride bytes of tile boundary rows
    int64_t scratch_stride;
};

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
        } else {
            cls = class_of_rows(R);
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

size_t phmm_workspace_bytes(int64_t n_pairs, int max_hap_len)
{
    if (n_pairs < 0) n_pairs = 0;
    if (max_hap_len < 1) max_hap_len = 1;
    return 64 * sizeof(int32_t) + (size_t)n_pairs * 2 * sizeof(int32_t) + 64 + (size_t)TILED_BLOCKS * scratch_stride_bytes(max_hap_len);
}

int phmm_init_tables() { DevTables t; return upload_tables(&t); }

const float *phmm_host_mm_table_f(int *n) { if (n) *n = MM_USED; return host_tables().mm_f.data(); }

int phmm_launch(int64_t n_pairs, const int32_t *pair_read, const int32_t *pair_hap,
                const int64_t *read_off, const int32_t *read_len,
                const uint8_t *rs, const uint8_t *q, const uint8_t *qi, const uint8_t *qd, const uint8_t *qc,
                const int64_t *hap_off, const int32_t *hap_len, const uint8_t *hap, int max_hap_len,
                double *out, void *d_work, size_t work_bytes, hipStream_t s)
{
    if (n_pairs == 0) return GBX_OK;
    if (n_pairs > 0x7fffffffLL - 1024) { set_error("phmm: more than 2^31 pairs in one call"); return GBX_ERR_UNSUPPORTED; }
    if (work_bytes < phmm_workspace_bytes(n_pairs, max_hap_len)) { set_error("phmm: workspace too small"); return GBX_ERR_ARG; }
    DevTables tabs;
    int rc = upload_tables(&tabs);
    if (rc) return rc;
    PhmmArgs A = {pair_read, pair_hap, read_off, read_len, rs, q, qi, qd, qc, hap_off, hap_len, hap, out, tabs};
    int32_t *wi = (int32_t *)d_work;
    PhmmWork W;
    W.counts = wi; W.cursors = wi + 8; W.next = wi + 16; W.dcount = wi + 24;
    W.order = wi + 64; W.dlist = wi + 64 + n_pairs;
    size_t off = (64 + (size_t)n_pairs * 2) * sizeof(int32_t);
    off = (off + 63) & ~(size_t)63;
    W.scratch = (char *)d_work + off;
    W.scratch_stride = (int64_t)scratch_stride_bytes(max_hap_len);
    GBX_HIP(hipMemsetAsync(d_work, 0, 64 * sizeof(int32_t), s));
    const int cb = (int)((n_pairs + 255) / 256);
    {
        Stage st("phmm_classify", s);
        hipLaunchKernelGGL(phmm_classify_kernel, dim3(cb), dim3(256), 0, s, A, n_pairs, W, 0);
        hipLaunchKernelGGL(phmm_classify_kernel, dim3(cb), dim3(256), 0, s, A, n_pairs, W, 1);
    }
    int dev_id = 0, cus = 256;
    (void)hipGetDevice(&dev_id);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev_id);
    auto grid = [&](int per_cu) { int64_t cap = (int64_t)cus * per_cu; return (int)(n_pairs < cap ? n_pairs : cap); };
    { Stage st("phmm_f32_rpl1", s); hipLaunchKernelGGL(phmm_f32_kernel<1>, dim3(grid(32)), dim3(64), 0, s, A, W, 0); }
    { Stage st("phmm_f32_rpl2", s); hipLaunchKernelGGL(phmm_f32_kernel<2>, dim3(grid(32)), dim3(64), 0, s, A, W, 1); }
    { Stage st("phmm_f32_rpl3", s); hipLaunchKernelGGL(phmm_f32_kernel<3>, dim3(grid(24)), dim3(64), 0, s, A, W, 2); }
    { Stage st("phmm_f32_rpl4", s); hipLaunchKernelGGL(phmm_f32_kernel<4>, dim3(grid(20)), dim3(64), 0, s, A, W, 3); }
    { Stage st("phmm_f32_rpl6", s); hipLaunchKernelGGL(phmm_f32_kernel<6>, dim3(grid(12)), dim3(64), 0, s, A, W, 4); }
    { Stage st("phmm_f32_rpl8", s); hipLaunchKernelGGL(phmm_f32_kernel<8>, dim3(std::min(grid(8), TILED_BLOCKS)), dim3(64), 0, s, A, W, 5); }
    { Stage st("phmm_f64_redo", s); hipLaunchKernelGGL(phmm_f64_kernel<4>, dim3(std::min(grid(8), TILED_BLOCKS)), dim3(64), 0, s, A, W); }
    GBX_HIP(hipGetLastError());
    return GBX_OK;
}

}  // namespace gbx
