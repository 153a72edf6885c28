// bsw — GPU driver with the CLI of R/benchmarks/bsw/main_banded.cpp:
//   bsw -pairs <InSeqFile> -t <threads> -b <batch_size> [-match N -mismatch N -ambig N -gapo N -gape N]
// Input format (main_banded.cpp:131-141): 3 lines per pair: seed score, target digits 0-4, query digits 0-4.
// -b is accepted for compatibility; the GPU path takes all pairs in one call.  -t = threads of the parallel
// ingest (line split, digit conversion, packing).
// Extras: --dump FILE writes "score tle gtle qle gscore max_off" per pair; --parse-only 1 stops after the ingest and
// prints pair count and checksums of the packed arrays (no GPU needed); --repeat N times the call N times and
// reports the fastest (the first call on fresh host pages also pays for pinning them).  Exit status 0 (the
// reference returns 1).
// --overlap S (round 5, SURVEY 8f rank 1 "ingest overlapped with compute"): the pairs are parsed in S slices and slice k is
// on its way through gbx_bsw_extend_host (a caller thread and host lane of its own: upload, kernels, download) while slice
// k+1 is still being converted; "e2e" = first byte parsed to last result in the caller's array (SURVEY 8d leg iii).
// --cache FILE (round 6, SURVEY 8f rank 1 "binary cache"): the converted arrays are written to FILE after the first run and
// mapped by later runs of the same input (size and modification time are checked) instead of converting the text again.
#include "driver_common.h"
#include <thread>
#include <algorithm>

int main(int argc, char **argv)
{
    const int gpus = take_gpus_flag(argc, argv);
    if (argc < 3) {
        fprintf(stderr, "usage: bsw -pairs <InSeqFile> -t <threads> -b <batch_size>\n");
        return EXIT_FAILURE;
    }
    int a = 1, b = 4, ambig = -1, o = 6, e = 1, threads = 1, batch = 0, repeat = 1, overlap = 0;
    const char *pairs = nullptr, *dump = nullptr, *cache = nullptr;
    bool parse_only = false;
    for (int i = 1; i + 1 < argc; i += 2) {
        const char *k = argv[i], *v = argv[i + 1];
        if (!strcmp(k, "-match")) a = atoi(v);
        else if (!strcmp(k, "-mismatch")) b = atoi(v);
        else if (!strcmp(k, "-ambig")) ambig = atoi(v);
        else if (!strcmp(k, "-gapo")) o = atoi(v);
        else if (!strcmp(k, "-gape")) e = atoi(v);
        else if (!strcmp(k, "-pairs")) pairs = v;
        else if (!strcmp(k, "-t")) threads = atoi(v);
        else if (!strcmp(k, "-b")) batch = atoi(v);
        else if (!strcmp(k, "--dump")) dump = v;
        else if (!strcmp(k, "--repeat")) repeat = atoi(v) > 0 ? atoi(v) : 1;
        else if (!strcmp(k, "--parse-only")) parse_only = atoi(v) != 0;
        else if (!strcmp(k, "--overlap")) overlap = atoi(v);
        else if (!strcmp(k, "--cache")) cache = v;
    }
    (void)batch;
    if (threads < 1) threads = 1;
    if (!pairs) { fprintf(stderr, "ERROR! pairFileName not specified.\n"); return EXIT_FAILURE; }
    // --cache FILE: the converted arrays of this input, written after the first conversion and mapped by later runs (driver_common.h:
    // InputCache; sections: idr, idq, len1, len2, h0, ref, qer).  A hit skips the text altogether: the ingest is a mmap, and the
    // host entry's upload workers read the page-cache pages directly.
    InputCache icache;
    const bool cache_hit = cache && !overlap && icache.open(cache, pairs, 0x777362 /* "bsw" */, 7);
    std::vector<char> text;
    if (!cache_hit && !slurp(pairs, text)) { fprintf(stderr, "Could not open file: %s\n", pairs); return EXIT_FAILURE; }

    if (overlap > 1 && !parse_only) {
        // device and host-lane set-up is not part of the timed region (the reference constructs its aligner objects before its
        // own, main_banded.cpp:262-270); two lanes: a slice's call may still be downloading when the next one starts
        print_device_banner(gpus);
        std::thread second([] { (void)gbx_host_prepare(); });
        (void)gbx_host_prepare();
        second.join();
        // ... nor is the runtime's first use of the kernels (code objects are loaded at the first launch: 40 ms): a call on a few
        // made-up pairs, as the plain flow's warm-up call on the first 64 real ones
        gbx_bsw_params W;
        gbx_bsw_default_params(&W);
        const int wn = 64, wl = 32;
        std::vector<uint8_t> wr((size_t)wn * wl + 8, 1), wq((size_t)wn * wl + 8, 1);
        std::vector<int64_t> wo((size_t)wn);
        std::vector<int32_t> wlen((size_t)wn, wl), wh((size_t)wn, 10);
        std::vector<gbx_bsw_result> wout((size_t)wn);
        for (int k = 0; k < wn; ++k) wo[(size_t)k] = (int64_t)k * wl;
        die_on(gbx_bsw_extend_host(&W, wn, wr.data(), (int64_t)wr.size(), wq.data(), (int64_t)wq.size(), wo.data(), wo.data(), wlen.data(), wlen.data(), wh.data(), wout.data()),
               "gbx_bsw_extend_host");
    }
    const double t_read0 = now_s();
    // split lines; numPairs = lines / 3 (main_banded.cpp:235)
    RawVec<const char *> line; RawVec<int> llen;
    if (!cache_hit) split_lines(text.data(), text.size() - 1, threads, line, llen);
    const int64_t n = cache_hit ? (int64_t)(icache.sec[0].second / 8) : (int64_t)line.size() / 3;
    printf("Number of input pairs: %ld\n", (long)n);
    // (arrays first touched by the threads that fill them: see RawVec)
    RawVec<int64_t> idr, idq;
    RawVec<int32_t> len1, len2, h0;
    RawVec<uint8_t> ref, qer;
    int64_t rb = 0, qb = 0;
    if (cache_hit) {
        idr.adopt((int64_t *)icache.sec[0].first, (size_t)n); idq.adopt((int64_t *)icache.sec[1].first, (size_t)n);
        len1.adopt((int32_t *)icache.sec[2].first, (size_t)n); len2.adopt((int32_t *)icache.sec[3].first, (size_t)n);
        h0.adopt((int32_t *)icache.sec[4].first, (size_t)n);
        ref.adopt((uint8_t *)icache.sec[5].first, icache.sec[5].second); qer.adopt((uint8_t *)icache.sec[6].first, icache.sec[6].second);
        rb = (int64_t)icache.sec[5].second - 8; qb = (int64_t)icache.sec[6].second - 8;
        if (icache.sec[1].second != (size_t)n * 8 || icache.sec[2].second != (size_t)n * 4 || icache.sec[3].second != (size_t)n * 4 ||
            icache.sec[4].second != (size_t)n * 4 || rb < 0 || qb < 0) { fprintf(stderr, "%s: malformed cache\n", cache); return EXIT_FAILURE; }
        printf("Input arrays mapped from the cache %s\n", cache);
    } else {
        idr.resize((size_t)n); idq.resize((size_t)n); len1.resize((size_t)n); len2.resize((size_t)n); h0.resize((size_t)n);
    }
    for (int64_t k = 0; k < n && !cache_hit; ++k) {             // offsets: a serial prefix over two ints per pair
        len1[k] = llen[3 * k + 1]; len2[k] = llen[3 * k + 2];
        if (len1[k] <= 0 || len2[k] <= 0) { fprintf(stderr, "pair %ld has an empty sequence\n", (long)k); return EXIT_FAILURE; }
        idr[k] = rb; idq[k] = qb;
        rb += (len1[k] + 3) & ~3; qb += (len2[k] + 3) & ~3;
    }
    if (!cache_hit) {
        ref.resize((size_t)rb + 8); qer.resize((size_t)qb + 8);
        memset(ref.data() + rb, 0, 8); memset(qer.data() + qb, 0, 8);
    }
    gbx_bsw_params P;
    gbx_bsw_default_params(&P);
    P.o_del = P.o_ins = o; P.e_del = P.e_ins = e;
    gbx_bsw_fill_scmat(a, b, ambig, P.mat);
    RawVec<gbx_bsw_result> out((size_t)n);
    auto convert_one = [&](int64_t k) {
        h0[k] = atoi(line[3 * k]);
        uint8_t *r = ref.data() + idr[k], *q = qer.data() + idq[k];
        const char *lr = line[3 * k + 1], *lq = line[3 * k + 2];
        const int n1 = len1[k], n2 = len2[k];
        for (int l = 0; l < n1; ++l) r[l] = (uint8_t)(lr[l] - 48);
        for (int l = n1; l < ((n1 + 3) & ~3); ++l) r[l] = 0;                  // the slot's padding
        for (int l = 0; l < n2; ++l) q[l] = (uint8_t)(lq[l] - 48);
        for (int l = n2; l < ((n2 + 3) & ~3); ++l) q[l] = 0;
        out[k] = gbx_bsw_result{-1, -1, -1, -1, -1, -1};                       // (the result array's pages, touched here too)
    };
    auto convert = [&](int64_t lo, int64_t hi) {
#pragma omp parallel for num_threads(threads) schedule(static)
        for (int64_t k = lo; k < hi; ++k) convert_one(k);
    };
    // the same with threads that exist only while they work: an idle OpenMP team spins, and in the overlapped flow it would
    // spin beside the host entry's upload workers (measured on a box with fewer cores than it reports: calls 10 x slower)
    std::vector<int64_t> rel_r, rel_q;
    auto convert_plain_threads = [&](int64_t lo, int64_t hi) {
        const int T = (int)std::min<int64_t>(threads, (hi - lo + 4095) / 4096);
        auto part = [&](int t) {
            const int64_t a = lo + (hi - lo) * t / T, b = lo + (hi - lo) * (t + 1) / T;
            for (int64_t k = a; k < b; ++k) { convert_one(k); rel_r[(size_t)k] = idr[k] - idr[lo]; rel_q[(size_t)k] = idq[k] - idq[lo]; }
        };
        std::vector<std::thread> th;
        for (int t = 1; t < T; ++t) th.emplace_back(part, t);
        part(0);
        for (auto &x : th) x.join();
    };
    double t_e2e = 0, t_gpu_span = 0, t_conv = now_s() - t_read0;      // t_conv: line split + offsets so far, the conversion is added below
    if (overlap > 1 && !parse_only && n >= overlap) {
        // slice k: its own stretch of the arenas (offsets re-based to it), sent off as soon as it is converted
        rel_r.resize((size_t)n); rel_q.resize((size_t)n);
        std::vector<std::thread> calls;
        std::vector<int> rcs((size_t)overlap, 0);
        std::vector<std::string> errs((size_t)overlap);
        double t_first = 0;
        for (int s = 0; s < overlap; ++s) {
            const int64_t lo = n * s / overlap, hi = n * (s + 1) / overlap;
            const double tc0 = now_s();
            convert_plain_threads(lo, hi);
            t_conv += now_s() - tc0;
            const int64_t br = idr[lo], bq = idq[lo];
            const int64_t er = hi < n ? idr[hi] : rb, eq = hi < n ? idq[hi] : qb;
            if (s == 0) t_first = now_s();
            calls.emplace_back([&, s, lo, hi, br, bq, er, eq] {
                rcs[(size_t)s] = gbx_bsw_extend_host(&P, hi - lo, ref.data() + br, er - br + 8, qer.data() + bq, eq - bq + 8, rel_r.data() + lo, rel_q.data() + lo,
                                                     len1.data() + lo, len2.data() + lo, h0.data() + lo, out.data() + lo);
                if (rcs[(size_t)s]) errs[(size_t)s] = gbx_last_error();
            });
        }
        for (auto &c : calls) c.join();
        t_e2e = now_s() - t_read0;
        t_gpu_span = now_s() - t_first;
        for (int s = 0; s < overlap; ++s)
            if (rcs[(size_t)s]) { fprintf(stderr, "gbx_bsw_extend_host failed (%d) on slice %d: %s\n", rcs[(size_t)s], s, errs[(size_t)s].c_str()); return EXIT_FAILURE; }
    } else if (cache_hit) {
        overlap = 0;
#pragma omp parallel for num_threads(threads) schedule(static)
        for (int64_t k = 0; k < n; ++k) out[k] = gbx_bsw_result{-1, -1, -1, -1, -1, -1};      // (the result array's pages)
    } else {
        overlap = 0;
        convert(0, n);
    }
    const double t_read = overlap ? t_conv : now_s() - t_read0;
    if (parse_only) {
        uint64_t h = fnv1a(h0.data(), (size_t)n * 4);
        h = fnv1a(len1.data(), (size_t)n * 4, h); h = fnv1a(len2.data(), (size_t)n * 4, h);
        for (int64_t k = 0; k < n; ++k) { h = fnv1a(&ref[idr[k]], (size_t)len1[k], h); h = fnv1a(&qer[idq[k]], (size_t)len2[k], h); }
        printf("{\"benchmark\":\"bsw\",\"pairs\":%ld,\"ingest_threads\":%d,\"ingest_seconds\":%.4f,\"ingest_mb_per_s\":%.1f,\"checksum\":\"%016llx\"}\n",
               (long)n, threads, t_read, text.size() / 1e6 / t_read, (unsigned long long)h);
        if (cache && !cache_hit && !InputCache::write(cache, pairs, 0x777362, {{idr.data(), (size_t)n * 8}, {idq.data(), (size_t)n * 8}, {len1.data(), (size_t)n * 4},
                {len2.data(), (size_t)n * 4}, {h0.data(), (size_t)n * 4}, {ref.data(), (size_t)rb + 8}, {qer.data(), (size_t)qb + 8}}))
            fprintf(stderr, "warning: could not write the cache %s\n", cache);
        return 0;
    }

    double dt = 0;
    if (!overlap) {
        print_device_banner(gpus);
        // runtime initialisation is not billed to the timed region (the reference constructs its aligner objects
        // before it, main_banded.cpp:262-270): staging buffers (print_device_banner), then a warm-up call on a tiny prefix
        if (n > 0) die_on(gbx_bsw_extend_host(&P, n < 64 ? n : 64, ref.data(), rb + 8, qer.data(), qb + 8, idr.data(), idq.data(),
                                              len1.data(), len2.data(), h0.data(), out.data()), "gbx_bsw_extend_host");
        for (int r = 0; r < repeat; ++r) {
            const double t0 = now_s();
            die_on(gbx_bsw_extend_host(&P, n, ref.data(), rb + 8, qer.data(), qb + 8, idr.data(), idq.data(), len1.data(),
                                       len2.data(), h0.data(), out.data()), "gbx_bsw_extend_host");
            const double t = now_s() - t0;
            if (repeat > 1) printf("call %d: %.4f s\n", r, t);
            if (r == 0 || t < dt) dt = t;
        }
        t_e2e = t_read + dt;
    } else {
        dt = t_gpu_span;
        printf("Parse and device calls overlapped in %d slices: first byte parsed to last result %.4f s (conversion alone is the read time below)\n", overlap, t_e2e);
    }
    printf("Executed MI355X HIP code...\n");
    printf("Read time = %0.2lf s\n", t_read);
    printf("Overall SW time (H2D + kernels + D2H) = %0.4lf s\n", dt);
    printf("Total Pairs processed: %ld\n", (long)n);
    double cells = 0;
    for (int64_t k = 0; k < n; ++k) cells += (double)len1[k] * len2[k];
    printf("SW cells(T)  = %.0f\nSW GCUPS  = %lf\n", cells, cells / dt / 1e9);
    printf("{\"benchmark\":\"bsw\",\"pairs\":%ld,\"cells\":%.0f,\"seconds\":%.6f,\"gcups\":%.3f,\"ingest_threads\":%d,\"ingest_seconds\":%.6f,\"overlap_slices\":%d,\"e2e_seconds\":%.6f}\n",
           (long)n, cells, dt, cells / dt / 1e9, threads, t_read, overlap, t_e2e);
    if (cache && !cache_hit && !overlap &&
        !InputCache::write(cache, pairs, 0x777362, {{idr.data(), (size_t)n * 8}, {idq.data(), (size_t)n * 8}, {len1.data(), (size_t)n * 4}, {len2.data(), (size_t)n * 4},
                                                   {h0.data(), (size_t)n * 4}, {ref.data(), (size_t)rb + 8}, {qer.data(), (size_t)qb + 8}}))
        fprintf(stderr, "warning: could not write the cache %s\n", cache);
    if (dump) {
        FILE *f = fopen(dump, "w");
        if (!f) { fprintf(stderr, "cannot write %s\n", dump); return EXIT_FAILURE; }
        for (int64_t k = 0; k < n; ++k)
            fprintf(f, "%d %d %d %d %d %d\n", out[k].score, out[k].tle, out[k].gtle, out[k].qle, out[k].gscore, out[k].max_off);
        fclose(f);
    }
    return 0;
}
