// bsw_class_shim.cpp — the public members of BandedPairWiseSW (R/benchmarks/bsw/bandedSWA.h:116-315) implemented on the
// C-ABI of libgbx.so: what the GenomicsBench bsw driver calls (main_banded.cpp:255 ctor, :286 getScores16, :347 getTicks)
// and the class's other entry points a caller such as bwa-mem2's extension code binds - scalarBandedSWA (:125, one pair),
// scalarBandedSWAWrapper (:131), getScores8 (:141; and the two batch wrappers the get* members forward to).  Compiled only together with the reference's own bandedSWA.h / main_banded.cpp, from where
// they lie (oracle/build_ref.sh): it shows that the unmodified driver runs on the GPU path when this file
// replaces bandedSWA.cpp.  This is the binding a maintainer of the reference would add; no reference code
// is copied here.
//
// GBX_SHIM_DUMP=<file>: every getScores16 call appends "score tle gtle qle gscore max_off" per pair (the
// driver itself prints no results), so that a test can compare them with the oracle.
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include "bandedSWA.h"                     // the reference's header, -I R/benchmarks/bsw
#include "gbx.h"

BandedPairWiseSW::BandedPairWiseSW(const int o_del, const int e_del, const int o_ins, const int e_ins, const int zdrop,
                                   const int end_bonus, const int8_t *mat_, const int8_t w_match, const int8_t w_mismatch,
                                   int numThreads)
{
    (void)numThreads;
    this->m = 5;
    this->o_del = o_del; this->e_del = e_del; this->o_ins = o_ins; this->e_ins = e_ins;
    this->zdrop = zdrop; this->end_bonus = end_bonus;
    this->mat = mat_;
    this->w_match = w_match; this->w_mismatch = w_mismatch;
    this->SW_cells = 0;
    this->sort1Ticks = 0;
    // One-time device setup belongs here, as the reference allocates its working memory in this constructor
    // (bandedSWA.cpp:52-126), outside the driver's timed region: streams + pinned staging buffers, and one
    // single-pair call that loads the code objects.
    static std::once_flag once;
    std::call_once(once, [&] {
        if (gbx_host_prepare() != GBX_OK) return;              // no device: getScores16 will report it
        gbx_bsw_params prm;
        gbx_bsw_default_params(&prm);
        const uint8_t base[4] = {0, 0, 0, 0};
        const int64_t off = 0;
        const int32_t len = 1, h0 = 1;
        gbx_bsw_result r;
        (void)gbx_bsw_extend_host(&prm, 1, base, 4, base, 4, &off, &off, &len, &len, &h0, &r);
    });
}

BandedPairWiseSW::~BandedPairWiseSW() {}

int64_t BandedPairWiseSW::getTicks() { return 0; }

static void params_of(gbx_bsw_params *prm, int o_del, int e_del, int o_ins, int e_ins, int zdrop, int end_bonus, int w, const int8_t *mat)
{
    gbx_bsw_default_params(prm);
    prm->o_del = o_del; prm->e_del = e_del; prm->o_ins = o_ins; prm->e_ins = e_ins;
    prm->zdrop = zdrop; prm->end_bonus = end_bonus; prm->w = w;
    for (int k = 0; k < 25; ++k) prm->mat[k] = mat[k];
}

// bandedSWA.cpp:128-249 (bwa's ksw_extend2): one pair; returns the score, the five other outputs through the pointers (each
// may be null, as in the reference).  One device call per pair - a caller with many pairs wants one of the batch members
// below; concurrent single-pair callers share device calls (gbx.h: gbx_host_combine_stats).
int BandedPairWiseSW::scalarBandedSWA(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int32_t w, int h0,
                                      int *_qle, int *_tle, int *_gtle, int *_gscore, int *_max_off)
{
    gbx_bsw_params prm;
    params_of(&prm, o_del, e_del, o_ins, e_ins, zdrop, end_bonus, w, mat);
    const int64_t off = 0;
    const int32_t l1 = tlen, l2 = qlen, h = h0;
    gbx_bsw_result r;
    const int rc = gbx_bsw_extend_host(&prm, 1, target, tlen, query, qlen, &off, &off, &l1, &l2, &h, &r);
    if (rc) { fprintf(stderr, "scalarBandedSWA: %s\n", gbx_last_error()); exit(EXIT_FAILURE); }
    if (_qle) *_qle = r.qle;
    if (_tle) *_tle = r.tle;
    if (_gtle) *_gtle = r.gtle;
    if (_gscore) *_gscore = r.gscore;
    if (_max_off) *_max_off = r.max_off;
    return r.score;
}

// bandedSWA.cpp:254-272: scalarBandedSWA over an array of pairs - here one call for all of them (same six fields per pair)
void BandedPairWiseSW::scalarBandedSWAWrapper(SeqPair *seqPairArray, uint8_t *seqBufRef, uint8_t *seqBufQer, int numPairs, int nthreads, int32_t w)
{
    getScores16(seqPairArray, seqBufRef, seqBufQer, numPairs, (uint16_t)nthreads, w);
}

// bandedSWA.cpp:424-446 / :1124-1148: the get* members are their batch wrappers
void BandedPairWiseSW::smithWatermanBatchWrapper16(SeqPair *pairArray, uint8_t *seqBufRef, uint8_t *seqBufQer, int32_t numPairs, uint16_t numThreads,
                                                   int32_t w)
{
    getScores16(pairArray, seqBufRef, seqBufQer, numPairs, numThreads, w);
}

// The 8-bit entry (bandedSWA.cpp:424-1120) exists in the reference for pairs whose scores fit a byte - bwa-mem2's caller
// sends a pair there only when qlen * match + h0 stays below 250 (the un-vendored tools/bwa-mem2 extension code) - and under
// that guarantee nothing saturates, so its outputs are the scalar routine's.  Here: the same kernels as getScores16
// (their compact 8-bit cell format is chosen per pair on the device), i.e. the scalar answer for every pair,
// also for pairs an 8-bit lane would have saturated on.
void BandedPairWiseSW::getScores8(SeqPair *pairArray, uint8_t *seqBufRef, uint8_t *seqBufQer, int32_t numPairs, uint16_t numThreads, int32_t w)
{
    getScores16(pairArray, seqBufRef, seqBufQer, numPairs, numThreads, w);
}

void BandedPairWiseSW::smithWatermanBatchWrapper8(SeqPair *pairArray, uint8_t *seqBufRef, uint8_t *seqBufQer, int32_t numPairs, uint16_t numThreads,
                                                  int32_t w)
{
    getScores16(pairArray, seqBufRef, seqBufQer, numPairs, numThreads, w);
}

void BandedPairWiseSW::getScores16(SeqPair *pairArray, uint8_t *seqBufRef, uint8_t *seqBufQer, int32_t numPairs,
                                   uint16_t numThreads, int32_t w)
{
    (void)numThreads;
    gbx_bsw_params prm;
    params_of(&prm, o_del, e_del, o_ins, e_ins, zdrop, end_bonus, w, mat);
    static_assert(sizeof(SeqPair) == sizeof(gbx_seqpair), "SeqPair layout (bandedSWA.h:91-100) is the C-ABI record");
    // the interface does not say how large the two buffers are (the driver strides them by its own
    // MAX_SEQ_LEN_REF / _QER, main_banded.cpp:56-58): the pairs themselves bound what is read
    int64_t ref_bytes = 0, qer_bytes = 0;
    for (int32_t k = 0; k < numPairs; ++k) {
        if (pairArray[k].idr + pairArray[k].len1 > ref_bytes) ref_bytes = pairArray[k].idr + pairArray[k].len1;
        if (pairArray[k].idq + pairArray[k].len2 > qer_bytes) qer_bytes = pairArray[k].idq + pairArray[k].len2;
    }
    const int rc = gbx_bsw_extend_seqpairs(&prm, (gbx_seqpair *)pairArray, numPairs, seqBufRef, ref_bytes, seqBufQer, qer_bytes);
    if (rc) { fprintf(stderr, "getScores16: %s\n", gbx_last_error()); exit(EXIT_FAILURE); }
    if (const char *path = getenv("GBX_SHIM_DUMP")) {
        static std::mutex mu;
        std::lock_guard<std::mutex> lk(mu);
        if (FILE *f = fopen(path, "a")) {
            for (int32_t k = 0; k < numPairs; ++k)
                fprintf(f, "%ld %d %d %d %d %d %d\n", (long)pairArray[k].id, pairArray[k].score, pairArray[k].tle, pairArray[k].gtle,
                        pairArray[k].qle, pairArray[k].gscore, pairArray[k].max_off);
            fclose(f);
        }
    }
}
