// gkl_pairhmm_shim.cpp — drop-in for libgkl_pairhmm_c.so as the GenomicsBench phmm driver uses it.
//
// The driver (R/benchmarks/phmm/PairHMMUnitTest.cpp:84-86) declares three C++-linkage functions and
// expects the library to define them, plus the storage of ConvertChar::conversionTable
// (R/benchmarks/phmm/pairhmm_common.h:27, initialised by the driver at :195):
//     void initPairHMM();
//     void computelikelihoodsboth(testcase *, double *, int batch_size);       called at :245
//     void computelikelihoodsfloat(testcase *, float *);                        declared, never called
// This file defines exactly those symbols on top of the C-ABI of libgbx.so (include/gbx.h), so the
// unmodified driver links against libgkl_pairhmm_c.so built from here and runs its batches on the GPU.
// `testcase` below restates the layout of pairhmm_common.h:20-24 (two ints, six pointers).
//
// The driver calls computelikelihoodsboth once per batch from many OpenMP threads; calls are independent
// and each one is a complete gbx_phmm_forward_host call (upload, kernels, download) — correct, but a
// driver that wants throughput hands all batches over at once.  Reads and haplotypes are
// recognised by every field the testcase record gives them (a read: its bases, its four quality tracks and its
// length; a haplotype: bases and length), so every distinct sequence of a batch is uploaded once and a caller
// that reuses a bases pointer with other qualities or another length gets two distinct entries.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <tuple>
#include <utility>
#include <vector>
#include "../../../include/gbx.h"

typedef struct {
    int rslen, haplen;
    const char *q, *i, *d, *c;
    const char *hap, *rs;
} testcase;

class ConvertChar {
public:
    static uint8_t conversionTable[255];
};
uint8_t ConvertChar::conversionTable[255];

void initPairHMM()
{
    int rc = gbx_phmm_init();
    if (!rc) rc = gbx_host_prepare();          /* streams + pinned staging buffers, outside the driver's timed region */
    if (rc) { fprintf(stderr, "initPairHMM: %s\n", gbx_last_error()); exit(EXIT_FAILURE); }
}

void computelikelihoodsboth(testcase *tc, double *out, int n)
{
    if (n <= 0) return;
    std::map<std::tuple<const char *, const char *, const char *, const char *, const char *, int>, int> rid;
    std::map<std::pair<const char *, int>, int> hid;
    std::vector<int64_t> roff, hoff;
    std::vector<int32_t> rlen, hlen, pr((size_t)n), ph((size_t)n);
    std::vector<uint8_t> rs, q, qi, qd, qc, hap;
    for (int k = 0; k < n; ++k) {
        const testcase &t = tc[k];
        auto r = rid.emplace(std::make_tuple(t.rs, t.q, t.i, t.d, t.c, t.rslen), (int)rlen.size());
        if (r.second) {
            roff.push_back((int64_t)rs.size());
            rlen.push_back(t.rslen);
            rs.insert(rs.end(), t.rs, t.rs + t.rslen);
            q.insert(q.end(), t.q, t.q + t.rslen);
            qi.insert(qi.end(), t.i, t.i + t.rslen);
            qd.insert(qd.end(), t.d, t.d + t.rslen);
            qc.insert(qc.end(), t.c, t.c + t.rslen);
        }
        auto h = hid.emplace(std::make_pair(t.hap, t.haplen), (int)hlen.size());
        if (h.second) {
            hoff.push_back((int64_t)hap.size());
            hlen.push_back(t.haplen);
            hap.insert(hap.end(), t.hap, t.hap + t.haplen);
        }
        pr[(size_t)k] = r.first->second;
        ph[(size_t)k] = h.first->second;
    }
    const int rc = gbx_phmm_forward_host(n, pr.data(), ph.data(), (int64_t)rlen.size(), roff.data(), rlen.data(),
                                         (int64_t)rs.size(), rs.data(), q.data(), qi.data(), qd.data(), qc.data(),
                                         (int64_t)hlen.size(), hoff.data(), hlen.data(), (int64_t)hap.size(), hap.data(), out);
    if (rc) { fprintf(stderr, "computelikelihoodsboth: %s\n", gbx_last_error()); exit(EXIT_FAILURE); }
}

// float results of the fp32 pass only in GKL; the driver never calls it.  One pair, as GKL's signature implies.
void computelikelihoodsfloat(testcase *tc, float *out)
{
    double d = 0.0;
    computelikelihoodsboth(tc, &d, 1);
    *out = (float)d;
}
