/* phmm_oracle.c — CPU restatement of the GATK/GKL logless Pair-HMM forward
 * algorithm as the GenomicsBench `phmm` driver calls it
 * (computelikelihoodsboth, R/benchmarks/phmm/PairHMMUnitTest.cpp:86,245).
 * TEST INFRASTRUCTURE ONLY (see gbx_oracle.h).
 *
 * Parity: UNPINNED.  The arithmetic lives in libgkl_pairhmm_c.so, built from
 * the un-vendored submodule tools/GKL (arun-sub/GKL, branch pv_c_interface,
 * commit unknown: R/.gitmodules:5-8; the directory is empty in this checkout)
 * and the reference tree holds no test vectors for it.  This file restates the
 * published GKL/GATK algorithm (Intel GKL `pairhmm` scalar template
 * compute_full_prob<NUMBER> and Context<NUMBER>):
 *   - ph2pr[q] = 10^(-q/10)
 *   - match-to-match transition from the Jacobian-log table approximation
 *   - fp32 pass scaled by 2^120, redone in fp64 (2^1020) iff result < 1e-28f
 *     (MIN_ACCEPTED, R/benchmarks/phmm/pairhmm_common.h:16)
 *   - output log10(result) - log10(INITIAL_CONSTANT)
 * It is anchored by hand-computable known answers and an independent fp64
 * evaluation in tests/test_phmm_cpu.py.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "gbx_oracle.h"

#define MAX_QUAL 254
#define JAC_TOL 8.0
#define JAC_STEP 0.0001
#define JAC_INV_STEP (1.0 / JAC_STEP)
#define JAC_SIZE 80001 /* (int)(8.0 / 0.0001) + 1 */
#define MM_TABLE_SIZE (((MAX_QUAL + 1) * (MAX_QUAL + 2)) >> 1)

static int g_init = 0;
static float ph2pr_f[128];
static double ph2pr_d[128];
static double jac_d[JAC_SIZE];
static float jac_f[JAC_SIZE];
static float mm_f[MM_TABLE_SIZE];
static double mm_d[MM_TABLE_SIZE];

/* GKL Context<NUMBER>::approximateLog10SumLog10 with the table of that NUMBER type */
static double approx_log10_sum(double small, double big, int use_float)
{
    if (small > big) { double t = big; big = small; small = t; }
    if (isinf(small) && small < 0) return big;
    if (isinf(big) && big < 0) return big;
    const double diff = big - small;
    if (diff >= JAC_TOL) return big;
    int ind;
    if (use_float) {
        const float v = (float)(diff * JAC_INV_STEP);
        ind = v > 0.0f ? (int)(v + 0.5f) : (int)(v - 0.5f);
        return big + jac_f[ind];
    } else {
        const double v = diff * JAC_INV_STEP;
        ind = v > 0.0 ? (int)(v + 0.5) : (int)(v - 0.5);
        return big + jac_d[ind];
    }
}

void oracle_phmm_init(void)
{
    if (g_init) return;
    for (int x = 0; x < 128; ++x) {
        ph2pr_f[x] = powf(10.0f, -((float)x) / 10.0f);
        ph2pr_d[x] = pow(10.0, -((double)x) / 10.0);
    }
    for (int k = 0; k < JAC_SIZE; ++k) {
        jac_d[k] = log10(1.0 + pow(10.0, -((double)k) * JAC_STEP));
        jac_f[k] = (float)jac_d[k];
    }
    const double inv_ln10 = 1.0 / log(10.0);
    for (int i = 0, offset = 0; i <= MAX_QUAL; offset += ++i)
        for (int j = 0; j <= i; ++j) {
            for (int uf = 0; uf < 2; ++uf) {
                const double ls = approx_log10_sum(-0.1 * i, -0.1 * j, uf);
                double pr = pow(10, ls);
                if (pr > 1.0) pr = 1.0;
                const double mm = pow(10, log1p(-pr) * inv_ln10);
                if (uf) mm_f[offset + j] = (float)mm; else mm_d[offset + j] = mm;
            }
        }
    g_init = 1;
}

/* exported so that the GPU path's table upload can be checked against it */
const float *oracle_phmm_mm_table_f(void) { oracle_phmm_init(); return mm_f; }
const double *oracle_phmm_mm_table_d(void) { oracle_phmm_init(); return mm_d; }

static inline int mm_index(int ins, int del)
{
    int mn = del, mx = ins;
    if (ins <= del) { mn = ins; mx = del; }
    return ((mx * (mx + 1)) >> 1) + mn;      /* quals are masked to 0..127, always <= MAX_QUAL */
}

#define DEFINE_FULL_PROB(NAME, T, PH2PR, MMTAB, INIT_CONST)                                         \
    static T NAME(int rslen, int haplen, const char *rs, const char *hap, const char *q,           \
                  const char *qi, const char *qd, const char *qc)                                  \
    {                                                                                               \
        const int COLS = haplen + 1;                                                                \
        T *M0 = (T *)calloc((size_t)COLS * 6, sizeof(T));                                           \
        T *X0 = M0 + COLS, *Y0 = X0 + COLS, *M1 = Y0 + COLS, *X1 = M1 + COLS, *Y1 = X1 + COLS;      \
        const T init = (T)(INIT_CONST) / (T)haplen;                                                 \
        for (int c = 0; c < COLS; ++c) { M0[c] = 0; X0[c] = 0; Y0[c] = init; }                      \
        for (int r = 1; r <= rslen; ++r) {                                                          \
            const int _i = qi[r - 1] & 127, _d = qd[r - 1] & 127, _c = qc[r - 1] & 127;             \
            const int _q = q[r - 1] & 127;                                                          \
            const T pMM = MMTAB[mm_index(_i, _d)], pGapM = (T)1.0 - PH2PR[_c];                      \
            const T pMX = PH2PR[_i], pXX = PH2PR[_c], pMY = PH2PR[_d], pYY = PH2PR[_c];             \
            const char _rs = rs[r - 1];                                                             \
            M1[0] = 0; X1[0] = X0[0] * pXX; Y1[0] = 0;                                              \
            for (int c = 1; c < COLS; ++c) {                                                        \
                const char _hap = hap[c - 1];                                                       \
                T distm = PH2PR[_q];                                                                \
                if (_rs == _hap || _rs == 'N' || _hap == 'N') distm = (T)1.0 - distm;               \
                else distm = distm / 3;                                                             \
                M1[c] = distm * (M0[c - 1] * pMM + X0[c - 1] * pGapM + Y0[c - 1] * pGapM);          \
                X1[c] = M0[c] * pMX + X0[c] * pXX;                                                  \
                Y1[c] = M1[c - 1] * pMY + Y1[c - 1] * pYY;                                          \
            }                                                                                       \
            T *t;                                                                                   \
            t = M0; M0 = M1; M1 = t; t = X0; X0 = X1; X1 = t; t = Y0; Y0 = Y1; Y1 = t;              \
        }                                                                                           \
        T result = 0;                                                                               \
        for (int c = 0; c < COLS; ++c) result += M0[c] + X0[c];                                     \
        T *base = M0 < M1 ? M0 : M1;                                                                \
        free(base);                                                                                 \
        return result;                                                                              \
    }

DEFINE_FULL_PROB(full_prob_f, float, ph2pr_f, mm_f, ldexpf(1.f, 120))
DEFINE_FULL_PROB(full_prob_d, double, ph2pr_d, mm_d, ldexp(1.0, 1020))

/* one (read, haplotype) pair; *used_double tells which precision produced the answer */
double oracle_phmm_pair(int rslen, int haplen, const char *rs, const char *hap, const char *q,
                        const char *qi, const char *qd, const char *qc, int *used_double)
{
    oracle_phmm_init();
    const float rf = full_prob_f(rslen, haplen, rs, hap, q, qi, qd, qc);
    if (rf < 1e-28f) {                                         /* MIN_ACCEPTED */
        const double rd = full_prob_d(rslen, haplen, rs, hap, q, qi, qd, qc);
        if (used_double) *used_double = 1;
        return log10(rd) - log10(ldexp(1.0, 1020));
    }
    if (used_double) *used_double = 0;
    return (double)(log10f(rf) - log10f(ldexpf(1.f, 120)));
}

/* always-fp64 evaluation (independent check of the fp32 path) */
double oracle_phmm_pair_f64(int rslen, int haplen, const char *rs, const char *hap, const char *q,
                            const char *qi, const char *qd, const char *qc)
{
    oracle_phmm_init();
    return log10(full_prob_d(rslen, haplen, rs, hap, q, qi, qd, qc)) - log10(ldexp(1.0, 1020));
}

/* flat batch form mirroring gbx_phmm_forward_host: pair p = (read pair_read[p], hap pair_hap[p]) */
void oracle_phmm_forward(int64_t n_pairs, const int32_t *pair_read, const int32_t *pair_hap,
                         const int64_t *read_off, const int32_t *read_len,
                         const char *rs, const char *q, const char *qi, const char *qd, const char *qc,
                         const int64_t *hap_off, const int32_t *hap_len, const char *hap,
                         double *out, int nthreads, int64_t *n_double)
{
    oracle_phmm_init();
    int64_t nd = 0;
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for schedule(dynamic, 16) num_threads(nthreads) reduction(+ : nd)
    for (int64_t p = 0; p < n_pairs; ++p) {
        const int r = pair_read[p], h = pair_hap[p];
        int ud = 0;
        const int64_t ro = read_off[r];
        out[p] = oracle_phmm_pair(read_len[r], hap_len[h], rs + ro, hap + hap_off[h], q + ro, qi + ro, qd + ro,
                                  qc + ro, &ud);
        nd += ud;
    }
    if (n_double) *n_double = nd;
}
