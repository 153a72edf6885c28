// bsw — GPU driver with the CLI of R/benchmarks/bsw/main_banded.cpp:
//   bsw -pairs <InSeqFile> -t <threads> -b <batch_size> [-match N -mismatch N -ambig N -gapo N -gape N]
// Input format (main_banded.cpp:131-141): 3 lines per pair: seed score, target digits 0-4, query digits 0-4.
// -b is accepted for compatibility; the GPU path takes all pairs in one call.  -t = threads of the parallel
// ingest (line split, digit conversion, packing).
// Extras: --dump FILE writes "score tle gtle qle gscore max_off" per pair; --parse-only 1 stops after the ingest and
// prints pair count and checksums of the packed arrays (no GPU needed); --repeat N times the call N times and
// reports the fastest (the first call on fresh host pages also pays for pinning them).  Exit status 0 (the
// reference returns 1).
#include "driver_common.h"

int main(int argc, char **argv)
{
    const int gpus = take_gpus_flag(argc, argv);
    if (argc < 3) {
        fprintf(stderr, "usage: bsw -pairs <InSeqFile> -t <threads> -b <batch_size>\n");
        return EXIT_FAILURE;
    }
    int a = 1, b = 4, ambig = -1, o = 6, e = 1, threads = 1, batch = 0, repeat = 1;
    const char *pairs = nullptr, *dump = nullptr;
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
    }
    (void)batch;
    if (threads < 1) threads = 1;
    if (!pairs) { fprintf(stderr, "ERROR! pairFileName not specified.\n"); return EXIT_FAILURE; }
    std::vector<char> text;
    if (!slurp(pairs, text)) { fprintf(stderr, "Could not open file: %s\n", pairs); return EXIT_FAILURE; }

    const double t_read0 = now_s();
    // split lines; numPairs = lines / 3 (main_banded.cpp:235)
    std::vector<const char *> line; std::vector<int> llen;
    split_lines(text.data(), text.size() - 1, threads, line, llen);
    const int64_t n = (int64_t)line.size() / 3;
    printf("Number of input pairs: %ld\n", (long)n);
    std::vector<int64_t> idr((size_t)n), idq((size_t)n);
    std::vector<int32_t> len1((size_t)n), len2((size_t)n), h0((size_t)n);
    int64_t rb = 0, qb = 0;
    for (int64_t k = 0; k < n; ++k) {                           // offsets: a serial prefix over two ints per pair
        len1[k] = llen[3 * k + 1]; len2[k] = llen[3 * k + 2];
        if (len1[k] <= 0 || len2[k] <= 0) { fprintf(stderr, "pair %ld has an empty sequence\n", (long)k); return EXIT_FAILURE; }
        idr[k] = rb; idq[k] = qb;
        rb += (len1[k] + 3) & ~3; qb += (len2[k] + 3) & ~3;
    }
    std::vector<uint8_t> ref((size_t)rb + 8), qer((size_t)qb + 8);
#pragma omp parallel for num_threads(threads) schedule(static)
    for (int64_t k = 0; k < n; ++k) {
        h0[k] = atoi(line[3 * k]);
        for (int l = 0; l < len1[k]; ++l) ref[idr[k] + l] = (uint8_t)(line[3 * k + 1][l] - 48);
        for (int l = 0; l < len2[k]; ++l) qer[idq[k] + l] = (uint8_t)(line[3 * k + 2][l] - 48);
    }
    const double t_read = now_s() - t_read0;
    if (parse_only) {
        uint64_t h = fnv1a(h0.data(), (size_t)n * 4);
        h = fnv1a(len1.data(), (size_t)n * 4, h); h = fnv1a(len2.data(), (size_t)n * 4, h);
        for (int64_t k = 0; k < n; ++k) { h = fnv1a(&ref[idr[k]], (size_t)len1[k], h); h = fnv1a(&qer[idq[k]], (size_t)len2[k], h); }
        printf("{\"benchmark\":\"bsw\",\"pairs\":%ld,\"ingest_threads\":%d,\"ingest_seconds\":%.4f,\"ingest_mb_per_s\":%.1f,\"checksum\":\"%016llx\"}\n",
               (long)n, threads, t_read, text.size() / 1e6 / t_read, (unsigned long long)h);
        return 0;
    }

    gbx_bsw_params P;
    gbx_bsw_default_params(&P);
    P.o_del = P.o_ins = o; P.e_del = P.e_ins = e;
    gbx_bsw_fill_scmat(a, b, ambig, P.mat);
    print_device_banner(gpus);
    std::vector<gbx_bsw_result> out((size_t)n);
    // runtime initialisation is not billed to the timed region (the reference constructs its aligner objects
    // before it, main_banded.cpp:262-270): staging buffers (print_device_banner), then a warm-up call on a tiny prefix
    if (n > 0) die_on(gbx_bsw_extend_host(&P, n < 64 ? n : 64, ref.data(), rb + 8, qer.data(), qb + 8, idr.data(), idq.data(),
                                          len1.data(), len2.data(), h0.data(), out.data()), "gbx_bsw_extend_host");
    double dt = 0;
    for (int r = 0; r < repeat; ++r) {
        const double t0 = now_s();
        die_on(gbx_bsw_extend_host(&P, n, ref.data(), rb + 8, qer.data(), qb + 8, idr.data(), idq.data(), len1.data(),
                                   len2.data(), h0.data(), out.data()), "gbx_bsw_extend_host");
        const double t = now_s() - t0;
        if (repeat > 1) printf("call %d: %.4f s\n", r, t);
        if (r == 0 || t < dt) dt = t;
    }
    printf("Executed MI355X HIP code...\n");
    printf("Read time = %0.2lf s\n", t_read);
    printf("Overall SW time (H2D + kernels + D2H) = %0.4lf s\n", dt);
    printf("Total Pairs processed: %ld\n", (long)n);
    double cells = 0;
    for (int64_t k = 0; k < n; ++k) cells += (double)len1[k] * len2[k];
    printf("SW cells(T)  = %.0f\nSW GCUPS  = %lf\n", cells, cells / dt / 1e9);
    printf("{\"benchmark\":\"bsw\",\"pairs\":%ld,\"cells\":%.0f,\"seconds\":%.6f,\"gcups\":%.3f}\n", (long)n, cells, dt, cells / dt / 1e9);
    if (dump) {
        FILE *f = fopen(dump, "w");
        if (!f) { fprintf(stderr, "cannot write %s\n", dump); return EXIT_FAILURE; }
        for (int64_t k = 0; k < n; ++k)
            fprintf(f, "%d %d %d %d %d %d\n", out[k].score, out[k].tle, out[k].gtle, out[k].qle, out[k].gscore, out[k].max_off);
        fclose(f);
    }
    return 0;
}
