// phmm — GPU driver with the CLI of R/benchmarks/phmm/PairHMMUnitTest.cpp:  phmm -f <testfile> [-l loops] [-t threads]
// Input (PairHMMUnitTest.cpp:95-140): per batch `num_reads num_haps`, reads `bases q i d c` (Phred+33), haplotypes.
// All batches are handed to the GPU in one call (pairs read-major / hap-minor per batch, :232-244).
// --print writes one "%lf" per result like the reference's PRINT_OUTPUT build (:262-267).
#include <sstream>
#include "driver_common.h"

int main(int argc, char **argv)
{
    const char *file = nullptr;
    int loops = 1, threads = 1;
    bool print = false;
    if (argc == 1) { printf("  -f, --testfile  name of test file\n  -l, --loop  number of loops\n  -t  --threads  number of threads\n"); return EXIT_FAILURE; }
    for (int i = 1; i < argc; ++i) {
        if ((!strcmp(argv[i], "-f") || !strcmp(argv[i], "--testfile")) && i + 1 < argc) file = argv[++i];
        else if ((!strcmp(argv[i], "-l") || !strcmp(argv[i], "--loop")) && i + 1 < argc) loops = atoi(argv[++i]);
        else if ((!strcmp(argv[i], "-t") || !strcmp(argv[i], "--threads")) && i + 1 < argc) threads = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--print")) print = true;
    }
    (void)threads;
    std::vector<char> text;
    if (!file || !slurp(file, text)) { printf("Cannot open file : %s", file ? file : "(null)"); return 0; }
    printf("Reading test data from file: %s\n", file);
    // whitespace-separated tokens, exactly what `is >> ...` consumes
    std::vector<std::pair<const char *, int>> tok;
    for (char *p = text.data(), *end = text.data() + text.size() - 1; p < end;) {
        while (p < end && isspace((unsigned char)*p)) ++p;
        if (p >= end) break;
        char *s = p;
        while (p < end && !isspace((unsigned char)*p)) ++p;
        tok.emplace_back(s, (int)(p - s));
    }
    std::vector<int64_t> read_off, hap_off;
    std::vector<int32_t> read_len, hap_len, pair_read, pair_hap;
    std::vector<uint8_t> rs, q, qi, qd, qc, hap;
    size_t t = 0, n_batches = 0;
    auto norm = [](const char *s, int n, int lo, std::vector<uint8_t> &dst) {       // normalize(), :89-93
        for (int k = 0; k < n; ++k) { int v = s[k] - 33; dst.push_back((uint8_t)(v < lo ? lo : v)); }
    };
    while (t + 2 <= tok.size()) {
        const int nr = atoi(std::string(tok[t].first, tok[t].second).c_str());
        const int nh = atoi(std::string(tok[t + 1].first, tok[t + 1].second).c_str());
        t += 2;
        if (t + (size_t)nr * 5 + nh > tok.size()) { fprintf(stderr, "truncated batch %zu\n", n_batches); return EXIT_FAILURE; }
        const int r0 = (int)read_len.size(), h0 = (int)hap_len.size();
        for (int r = 0; r < nr; ++r, t += 5) {
            const int len = tok[t].second;
            read_off.push_back((int64_t)rs.size()); read_len.push_back(len);
            rs.insert(rs.end(), tok[t].first, tok[t].first + len);
            norm(tok[t + 1].first, len, 6, q); norm(tok[t + 2].first, len, 0, qi);
            norm(tok[t + 3].first, len, 0, qd); norm(tok[t + 4].first, len, 0, qc);
        }
        for (int h = 0; h < nh; ++h, ++t) {
            hap_off.push_back((int64_t)hap.size()); hap_len.push_back(tok[t].second);
            hap.insert(hap.end(), tok[t].first, tok[t].first + tok[t].second);
        }
        for (int r = 0; r < nr; ++r)
            for (int h = 0; h < nh; ++h) { pair_read.push_back(r0 + r); pair_hap.push_back(h0 + h); }
        ++n_batches;
    }
    for (auto *v : {&rs, &q, &qi, &qd, &qc, &hap}) v->resize(v->size() + 8);
    printf("Num Batches %zu, Num threads %d\n", n_batches, threads);
    const int64_t np = (int64_t)pair_read.size();
    std::vector<double> out((size_t)np + 1);
    print_device_banner();
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
