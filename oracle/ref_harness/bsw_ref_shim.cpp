// bsw_ref_shim.cpp — our own flat-array shim around the *reference's* bsw
// kernel object, so tests can call the real reference code.  Compiled by
// oracle/build_ref.sh together with /root/reference/benchmarks/bsw/bandedSWA.cpp
// (compiled where it lies; never copied) into oracle/_ref/libbsw_ref.so.
// TEST INFRASTRUCTURE ONLY.  Only exists in the build container.
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "bandedSWA.h"            // from -I/root/reference/benchmarks/bsw
#include "../../include/gbx.h"

// The driver defines this global (main_banded.cpp:71); bandedSWA.cpp declares it extern (:44).
uint64_t prof[10][112];

static BandedPairWiseSW *make(const gbx_bsw_params *p)
{
    // main_banded.cpp:255-257: (o,e,o,e,zdrop,end_bonus,mat,w_match,w_mismatch,1)
    return new BandedPairWiseSW(p->o_del, p->e_del, p->o_ins, p->e_ins, p->zdrop,
                                p->end_bonus, p->mat, p->mat[0], (int8_t)(-p->mat[1]), 1);
}

extern "C" int ref_bsw_scalar(const gbx_bsw_params *p, int64_t n,
                              const uint8_t *ref, const uint8_t *qer,
                              const int64_t *idr, const int64_t *idq,
                              const int32_t *len1, const int32_t *len2, const int32_t *h0,
                              gbx_bsw_result *out)
{
    BandedPairWiseSW *sw = make(p);
    for (int64_t k = 0; k < n; ++k) {
        int qle, tle, gtle, gscore, max_off;
        int sc = sw->scalarBandedSWA(len2[k], qer + idq[k], len1[k], ref + idr[k], p->w, h0[k],
                                     &qle, &tle, &gtle, &gscore, &max_off);
        out[k].score = sc; out[k].tle = tle; out[k].gtle = gtle; out[k].qle = qle;
        out[k].gscore = gscore; out[k].max_off = max_off;
    }
    delete sw;
    return 0;
}

// Runs getScores16 over batches of `batch` pairs exactly as the driver does
// (main_banded.cpp:283-286): batch-local ids, arrays padded so the kernel's
// pad writes (bandedSWA.cpp:1172-1177) and prefetch reads (:1261) stay in bounds.
extern "C" int ref_bsw_getscores16(const gbx_bsw_params *p, int64_t n,
                                   const uint8_t *ref, const uint8_t *qer,
                                   const int64_t *idr, const int64_t *idq,
                                   const int32_t *len1, const int32_t *len2, const int32_t *h0,
                                   gbx_bsw_result *out, int32_t batch)
{
    if (batch <= 0) batch = 512;
    BandedPairWiseSW *sw = make(p);
    const int64_t cap = ((batch + 15) / 16) * 16 + 16;
    SeqPair *buf = (SeqPair *)_mm_malloc(cap * sizeof(SeqPair), 64);
    for (int64_t b = 0; b < n; b += batch) {
        const int32_t m = (int32_t)((n - b) < batch ? (n - b) : batch);
        memset(buf, 0, cap * sizeof(SeqPair));
        for (int32_t k = 0; k < m; ++k) {
            SeqPair sp;
            sp.idr = idr[b + k]; sp.idq = idq[b + k]; sp.id = k;
            sp.len1 = len1[b + k]; sp.len2 = len2[b + k]; sp.h0 = h0[b + k];
            sp.seqid = sp.regid = sp.score = sp.tle = sp.gtle = sp.qle = -1;
            sp.gscore = sp.max_off = -1;
            buf[k] = sp;
        }
        sw->getScores16(buf, (uint8_t *)ref, (uint8_t *)qer, m, 1, p->w);
        for (int32_t k = 0; k < m; ++k) {
            const SeqPair &sp = buf[k];
            gbx_bsw_result &r = out[b + sp.id];
            r.score = sp.score; r.tle = sp.tle; r.gtle = sp.gtle; r.qle = sp.qle;
            r.gscore = sp.gscore; r.max_off = sp.max_off;
        }
    }
    _mm_free(buf);
    delete sw;
    return 0;
}

// Multi-threaded variant used as the timed CPU baseline: one BandedPairWiseSW per OpenMP thread and
// `omp for schedule(dynamic,1)` over batches, the driver's own structure (main_banded.cpp:253-291).
#include <omp.h>
extern "C" int ref_bsw_getscores16_mt(const gbx_bsw_params *p, int64_t n,
                                      const uint8_t *ref, const uint8_t *qer,
                                      const int64_t *idr, const int64_t *idq,
                                      const int32_t *len1, const int32_t *len2, const int32_t *h0,
                                      gbx_bsw_result *out, int32_t batch, int32_t nthreads,
                                      double *kernel_seconds)
{
    if (batch <= 0) batch = 512;
    if (nthreads < 1) nthreads = 1;
    const int64_t cap = ((batch + 15) / 16) * 16 + 16;
    double t0 = 0, t1 = 0;
#pragma omp parallel num_threads(nthreads)
    {
        // objects are built before the driver's timed region (main_banded.cpp:253-258 vs :272)
        BandedPairWiseSW *sw = make(p);
        SeqPair *buf = (SeqPair *)_mm_malloc(cap * sizeof(SeqPair), 64);
#pragma omp barrier
#pragma omp master
        t0 = omp_get_wtime();
#pragma omp for schedule(dynamic, 1)
        for (int64_t b = 0; b < n; b += batch) {
            const int32_t m = (int32_t)((n - b) < batch ? (n - b) : batch);
            memset(buf, 0, cap * sizeof(SeqPair));
            for (int32_t k = 0; k < m; ++k) {
                SeqPair sp;
                sp.idr = idr[b + k]; sp.idq = idq[b + k]; sp.id = k;
                sp.len1 = len1[b + k]; sp.len2 = len2[b + k]; sp.h0 = h0[b + k];
                sp.seqid = sp.regid = sp.score = sp.tle = sp.gtle = sp.qle = -1;
                sp.gscore = sp.max_off = -1;
                buf[k] = sp;
            }
            sw->getScores16(buf, (uint8_t *)ref, (uint8_t *)qer, m, 1, p->w);
            for (int32_t k = 0; k < m; ++k) {
                const SeqPair &sp = buf[k];
                gbx_bsw_result &r = out[b + sp.id];
                r.score = sp.score; r.tle = sp.tle; r.gtle = sp.gtle; r.qle = sp.qle;
                r.gscore = sp.gscore; r.max_off = sp.max_off;
            }
        }
#pragma omp master
        t1 = omp_get_wtime();
        _mm_free(buf);
        delete sw;
    }
    if (kernel_seconds) *kernel_seconds = t1 - t0;
    return 0;
}
