// bsw_members_harness.cpp — TEST INFRASTRUCTURE.  Calls every public entry point of the reference's BandedPairWiseSW class
// (R/benchmarks/bsw/bandedSWA.h:116-315) that a caller other than main_banded.cpp may bind, on a pairs file in the
// driver's format (three lines per pair: h0, target, query; digits), and prints what each returned:
//     <member> <pair> score tle gtle qle gscore max_off
// Compiled by oracle/build_ref.sh twice from this one file, with the reference's header from where it lies:
//   oracle/_ref/bsw_members_gbx   + csrc/shims/bsw_class_shim.cpp + libgbx.so  (the drop-in; needs a GPU)
//   oracle/_ref/bsw_members_ref   + R/benchmarks/bsw/bandedSWA.cpp             (the reference itself; scalar members only:
//                                   its getScores8 does not return on arbitrary pairs)
// tests/test_refdrivers_gpu.py compares the two with each other and with the oracle.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <omp.h>
#include "bandedSWA.h"

uint64_t prof[10][112];                    // the driver's global (main_banded.cpp:71; bandedSWA.cpp:44 declares it extern)

static void bwa_fill(int a, int b, int ambig, int8_t mat[25])
{
    int k = 0;
    for (int i = 0; i < 4; ++i) { for (int j = 0; j < 4; ++j) mat[k++] = i == j ? a : -b; mat[k++] = ambig; }
    for (int j = 0; j < 5; ++j) mat[k++] = ambig;
}

int main(int argc, char **argv)
{
    if (argc < 3) { fprintf(stderr, "usage: %s <pairs file> <members: s w 8 b B 6 (any of)> [threads]\n", argv[0]); return 2; }
    FILE *f = fopen(argv[1], "r");
    if (!f) { perror(argv[1]); return 2; }
    const int threads = argc > 3 ? atoi(argv[3]) : 1;
    std::vector<std::string> lines;
    {
        char *buf = nullptr; size_t cap = 0; ssize_t n;
        while ((n = getline(&buf, &cap, f)) > 0) { while (n && (buf[n - 1] == '\n' || buf[n - 1] == '\r')) --n; lines.emplace_back(buf, (size_t)n); }
        free(buf); fclose(f);
    }
    const int np = (int)(lines.size() / 3);
    const int cap = ((np + 15) / 16) * 16 + 16;
    // every pair a slot of its own, strided like the driver's buffers (main_banded.cpp:56-58)
    size_t sr = 16, sq = 16;
    for (int k = 0; k < np; ++k) { if (lines[3 * k + 1].size() + 16 > sr) sr = lines[3 * k + 1].size() + 16; if (lines[3 * k + 2].size() + 16 > sq) sq = lines[3 * k + 2].size() + 16; }
    std::vector<uint8_t> ref((size_t)cap * sr, 0), qer((size_t)cap * sq, 0);
    std::vector<SeqPair> in((size_t)cap);
    memset(in.data(), 0, in.size() * sizeof(SeqPair));
    for (int k = 0; k < np; ++k) {
        SeqPair &p = in[(size_t)k];
        p.h0 = atoi(lines[3 * k].c_str());
        p.idr = (int64_t)k * (int64_t)sr; p.idq = (int64_t)k * (int64_t)sq; p.id = k;
        p.len1 = (int32_t)lines[3 * k + 1].size(); p.len2 = (int32_t)lines[3 * k + 2].size();
        for (int j = 0; j < p.len1; ++j) ref[(size_t)p.idr + j] = (uint8_t)(lines[3 * k + 1][j] - '0');
        for (int j = 0; j < p.len2; ++j) qer[(size_t)p.idq + j] = (uint8_t)(lines[3 * k + 2][j] - '0');
    }
    int8_t mat[25];
    bwa_fill(1, 4, -1, mat);
    const int w = 100;
    BandedPairWiseSW sw(6, 1, 6, 1, 100, 5, mat, 1, 4, 1);
    auto fresh = [&] {
        std::vector<SeqPair> v = in;
        for (int k = 0; k < np; ++k) v[(size_t)k].score = v[(size_t)k].tle = v[(size_t)k].gtle = v[(size_t)k].qle = v[(size_t)k].gscore = v[(size_t)k].max_off = -7;
        return v;
    };
    auto print = [&](const char *who, const std::vector<SeqPair> &v) {
        for (int k = 0; k < np; ++k) {
            const SeqPair &p = v[(size_t)k];
            printf("%s %ld %d %d %d %d %d %d\n", who, (long)p.id, p.score, p.tle, p.gtle, p.qle, p.gscore, p.max_off);
        }
    };
    for (const char *m = argv[2]; *m; ++m) {
        std::vector<SeqPair> v = fresh();
        switch (*m) {
        case 's': {                         // scalarBandedSWA, one pair per call, from `threads` OpenMP threads at once
#pragma omp parallel for num_threads(threads) schedule(dynamic, 1)
            for (int k = 0; k < np; ++k) {
                SeqPair &p = v[(size_t)k];
                p.score = sw.scalarBandedSWA(p.len2, qer.data() + p.idq, p.len1, ref.data() + p.idr, w, p.h0, &p.qle, &p.tle, &p.gtle, &p.gscore, &p.max_off);
            }
            print("scalarBandedSWA", v);
            break;
        }
        case 'w': sw.scalarBandedSWAWrapper(v.data(), ref.data(), qer.data(), np, 1, w); print("scalarBandedSWAWrapper", v); break;
        case '8': sw.getScores8(v.data(), ref.data(), qer.data(), np, 1, w); print("getScores8", v); break;
        case 'b': sw.smithWatermanBatchWrapper8(v.data(), ref.data(), qer.data(), np, 1, w); print("smithWatermanBatchWrapper8", v); break;
        case 'B': sw.smithWatermanBatchWrapper16(v.data(), ref.data(), qer.data(), np, 1, w); print("smithWatermanBatchWrapper16", v); break;
        case '6': sw.getScores16(v.data(), ref.data(), qer.data(), np, 1, w); print("getScores16", v); break;
        default: fprintf(stderr, "unknown member '%c'\n", *m); return 2;
        }
    }
    return 0;
}
