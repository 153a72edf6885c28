// bsw_class_shim.cpp — the members of BandedPairWiseSW that the GenomicsBench bsw driver calls
// (R/benchmarks/bsw/main_banded.cpp:255 ctor, :286 getScores16, :347 getTicks), implemented on the C-ABI of
// libgbx.so.  Compiled only together with the reference's own bandedSWA.h / main_banded.cpp, from where
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

void BandedPairWiseSW::getScores16(SeqPair *pairArray, uint8_t *seqBufRef, uint8_t *seqBufQer, int32_t numPairs,
                                   uint16_t numThreads, int32_t w)
{
    (void)numThreads;
    gbx_bsw_params prm;
    gbx_bsw_default_params(&prm);
    prm.o_del = o_del; prm.e_del = e_del; prm.o_ins = o_ins; prm.e_ins = e_ins;
    prm.zdrop = zdrop; prm.end_bonus = end_bonus; prm.w = w;
    for (int k = 0; k < 25; ++k) prm.mat[k] = mat[k];
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
