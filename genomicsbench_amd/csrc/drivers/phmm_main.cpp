// phmm — GPU driver with the CLI of R/benchmarks/phmm/PairHMMUnitTest.cpp:  phmm -f <testfile> [-l loops] [-t threads]
// Input (PairHMMUnitTest.cpp:95-140): per batch `num_reads num_haps`, reads `bases q i d c` (Phred+33), haplotypes.
// All batches are handed to the GPU in one call (pairs read-major / hap-minor per batch, :232-244).
// --print writes one "%lf" per result like the reference's PRINT_OUTPUT build (:262-267).
// -t = threads of the parallel ingest; --parse-only stops after it and prints counts and a checksum (no GPU needed).
#include <sstream>
#include "driver_common.h"

int main(int argc, char **argv)
{
    const int gpus = take_gpus_flag(argc, argv);
    const char *file = nullptr;
    int loops = 1, threads = 1;
    bool print = false, parse_only = false;
    if (argc == 1) { printf("  -f, --testfile  name of test file\n  -l, --loop  number of loops\n  -t  --threads  number of threads\n"); return EXIT_FAILURE; }
    for (int i = 1; i < argc; ++i) {
        if ((!strcmp(argv[i], "-f") || !strcmp(argv[i], "--testfile")) && i + 1 < argc) file = argv[++i];
        else if ((!strcmp(argv[i], "-l") || !strcmp(argv[i], "--loop")) && i + 1 < argc) loops = atoi(argv[++i]);
        else if ((!strcmp(argv[i], "-t") || !strcmp(argv[i], "--threads")) && i + 1 < argc) threads = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--print")) print = true;
        else if (!strcmp(argv[i], "--parse-only")) parse_only = true;
    }
    if (threads < 1) threads = 1;
    std::vector<char> text;
    if (!file || !slurp(file, text)) { printf("Cannot open file : %s", file ? file : "(null)"); return 0; }
    printf("Reading test data from file: %s\n", file);
    // ---- parallel ingest: whitespace-separated tokens, exactly what `is >> ...` consumes.  "Token starts here" is a
    // local predicate (non-space byte behind a space byte or the file start), so every thread lists the tokens that
    // begin in its own byte range; a serial walk over the token index finds the batches (2 + 5*reads + haps tokens
    // each), prefix sums give the arena offsets, and reads / haplotypes are copied and normalised independently.
    const double t_read0 = now_s();
    const char *tp = text.data();
    const size_t tn = text.size() - 1;
    std::vector<std::vector<std::pair<const char *, int>>> part((size_t)threads);
#pragma omp parallel num_threads(threads)
    {
        const int th = omp_get_thread_num(), T = omp_get_num_threads();
        const size_t lo = tn * (size_t)th / (size_t)T, hi = tn * (size_t)(th + 1) / (size_t)T;
        auto &v = part[(size_t)th];
        for (size_t i = lo; i < hi; ++i) {
            if (isspace((unsigned char)tp[i]) || (i > 0 && !isspace((unsigned char)tp[i - 1]))) continue;
            size_t e = i + 1;
            while (e < tn && !isspace((unsigned char)tp[e])) ++e;
            v.emplace_back(tp + i, (int)(e - i));
            i = e;
        }
    }
    std::vector<std::pair<const char *, int>> tok;
    for (auto &v : part) tok.insert(tok.end(), v.begin(), v.end());
    struct Batch { size_t t0; int nr, nh, r0, h0; int64_t p0; };
    std::vector<Batch> batches;
    size_t t = 0;
    int n_reads = 0, n_haps = 0;
    int64_t n_pairs = 0;
    while (t + 2 <= tok.size()) {
        const int nr = atoi(std::string(tok[t].first, tok[t].second).c_str());
        const int nh = atoi(std::string(tok[t + 1].first, tok[t + 1].second).c_str());
        t += 2;
        if (t + (size_t)nr * 5 + nh > tok.size()) { fprintf(stderr, "truncated batch %zu\n", batches.size()); return EXIT_FAILURE; }
        batches.push_back({t, nr, nh, n_reads, n_haps, n_pairs});
        t += (size_t)nr * 5 + nh; n_reads += nr; n_haps += nh; n_pairs += (int64_t)nr * nh;
    }
    const size_t n_batches = batches.size();
    std::vector<int64_t> read_off((size_t)n_reads), hap_off((size_t)n_haps);
    std::vector<int32_t> read_len((size_t)n_reads), hap_len((size_t)n_haps), pair_read((size_t)n_pairs), pair_hap((size_t)n_pairs);
#pragma omp parallel for num_threads(threads) schedule(static)
    for (size_t b = 0; b < n_batches; ++b) {
        const Batch &B = batches[b];
        for (int r = 0; r < B.nr; ++r) read_len[(size_t)(B.r0 + r)] = tok[B.t0 + (size_t)r * 5].second;
        for (int h = 0; h < B.nh; ++h) hap_len[(size_t)(B.h0 + h)] = tok[B.t0 + (size_t)B.nr * 5 + h].second;
        int64_t k = B.p0;
        for (int r = 0; r < B.nr; ++r)
            for (int h = 0; h < B.nh; ++h, ++k) { pair_read[(size_t)k] = B.r0 + r; pair_hap[(size_t)k] = B.h0 + h; }
    }
    int64_t rbytes = 0, hbytes = 0;
    for (int r = 0; r < n_reads; ++r) { read_off[(size_t)r] = rbytes; rbytes += read_len[(size_t)r]; }
    for (int h = 0; h < n_haps; ++h) { hap_off[(size_t)h] = hbytes; hbytes += hap_len[(size_t)h]; }
    std::vector<uint8_t> rs((size_t)rbytes + 8), q((size_t)rbytes + 8), qi((size_t)rbytes + 8), qd((size_t)rbytes + 8),
        qc((size_t)rbytes + 8), hap((size_t)hbytes + 8);
    auto norm = [](const char *s, int n, int lo, uint8_t *dst) {                       // normalize(), :89-93
        for (int k = 0; k < n; ++k) { const int v = s[k] - 33; dst[k] = (uint8_t)(v < lo ? lo : v); }
    };
#pragma omp parallel for num_threads(threads) schedule(dynamic, 16)
    for (size_t b = 0; b < n_batches; ++b) {
        const Batch &B = batches[b];
        for (int r = 0; r < B.nr; ++r) {
            const size_t tk = B.t0 + (size_t)r * 5;
            const int len = tok[tk].second;
            const int64_t o = read_off[(size_t)(B.r0 + r)];
            memcpy(&rs[(size_t)o], tok[tk].first, (size_t)len);
            // the four quality strings have the read's length in a well-formed file; a shorter one is padded with its floor
            auto track = [&](size_t k, int lo, std::vector<uint8_t> &dst) {
                const int m = tok[k].second < len ? tok[k].second : len;
                norm(tok[k].first, m, lo, &dst[(size_t)o]);
                for (int z = m; z < len; ++z) dst[(size_t)o + z] = (uint8_t)lo;
            };
            track(tk + 1, 6, q); track(tk + 2, 0, qi); track(tk + 3, 0, qd); track(tk + 4, 0, qc);
        }
        for (int h = 0; h < B.nh; ++h) {
            const size_t tk = B.t0 + (size_t)B.nr * 5 + h;
            memcpy(&hap[(size_t)hap_off[(size_t)(B.h0 + h)]], tok[tk].first, (size_t)tok[tk].second);
        }
    }
    const double t_read = now_s() - t_read0;
    if (parse_only) {
        uint64_t h = fnv1a(read_len.data(), (size_t)n_reads * 4);
        h = fnv1a(hap_len.data(), (size_t)n_haps * 4, h);
        h = fnv1a(pair_read.data(), (size_t)n_pairs * 4, h); h = fnv1a(pair_hap.data(), (size_t)n_pairs * 4, h);
        h = fnv1a(rs.data(), (size_t)rbytes, h); h = fnv1a(q.data(), (size_t)rbytes, h); h = fnv1a(qi.data(), (size_t)rbytes, h);
        h = fnv1a(qd.data(), (size_t)rbytes, h); h = fnv1a(qc.data(), (size_t)rbytes, h); h = fnv1a(hap.data(), (size_t)hbytes, h);
        printf("{\"benchmark\":\"phmm\",\"batches\":%zu,\"pairs\":%lld,\"ingest_threads\":%d,\"ingest_seconds\":%.4f,\"ingest_mb_per_s\":%.1f,\"checksum\":\"%016llx\"}\n",
               n_batches, (long long)n_pairs, threads, t_read, tn / 1e6 / t_read, (unsigned long long)h);
        return 0;
    }
    printf("Num Batches %zu, Num threads %d\n", n_batches, threads);
    const int64_t np = (int64_t)pair_read.size();
    std::vector<double> out((size_t)np + 1);
    print_device_banner(gpus);
    die_on(gbx_phmm_init(), "gbx_phmm_init");                    // initPairHMM(), :193
    double dt = 0;
    for (int l = 0; l < (loops < 1 ? 1 : loops); ++l) {
        const double t0 = now_s();
        die_on(gbx_phmm_forward_host(np, pair_read.data(), pair_hap.data(), (int64_t)read_len.size(), read_off.data(),
                                     read_len.data(), (int64_t)rs.size(), rs.data(), q.data(), qi.data(), qd.data(), qc.data(),
                                     (int64_t)hap_len.size(), hap_off.data(), hap_len.data(), (int64_t)hap.size(), hap.data(),
                                     out.data()), "gbx_phmm_forward_host");
        dt += now_s() - t0;
    }
    if (print) for (int64_t k = 0; k < np; ++k) printf("%lf\n", out[k]);
    double cells = 0;
    for (int64_t k = 0; k < np; ++k) cells += (double)read_len[pair_read[k]] * hap_len[pair_hap[k]];
    printf("\nPairHMM completed. Kernel runtime: %.2f sec\n", dt);
    printf("{\"benchmark\":\"phmm\",\"pairs\":%lld,\"cells\":%.0f,\"seconds\":%.6f,\"gcups\":%.3f}\n", (long long)np, cells, dt,
           cells * (loops < 1 ? 1 : loops) / dt / 1e9);
    return 0;
}
