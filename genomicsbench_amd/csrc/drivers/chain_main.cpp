// chain — GPU driver with the CLI of R/benchmarks/chain/src/main.cpp:  chain -i <in> -o <out> [-t T] [-h]
// Input: host_data_io.cpp:13-51 (header `n avg_qspan max_dist_x max_dist_y bw n_segs`, n lines `x y`, `EOR`).
// Like the reference the output file is opened "w"; results are written (print_return format,
// host_data_io.cpp:53-60) when --print is given (the reference needs a PRINT_OUTPUT rebuild for that).
// -t = threads of the parallel ingest; --parse-only stops after it and prints counts and a checksum (no GPU needed).
#include <unistd.h>
#include "driver_common.h"

static void help() { fprintf(stderr, "usage: chain -i <input> -o <output> [-t threads] [--print] [--cache <file>]\n"); }

// the call: poff / pax / pay / phdr are the parsed arrays or, with --cache on a later run of the same input, the mapped ones
static int run_calls(int gpus, int64_t nc, int64_t na, const int64_t *poff, const uint64_t *pax, const uint64_t *pay, const gbx_chain_call *phdr,
                     bool print, FILE *fo)
{
    print_device_banner(gpus);
    std::vector<int32_t> score((size_t)na + 1), parent((size_t)na + 1);
    if (nc > 0) {                                               // warm-up on the first call only
        int64_t o2[2] = {0, poff[1]};
        die_on(gbx_chain_host(1, o2, pax, pay, phdr, score.data(), parent.data(), nullptr, nullptr), "gbx_chain_host");
    }
    const double t0 = now_s();
    die_on(gbx_chain_host(nc, poff, pax, pay, phdr, score.data(), parent.data(), nullptr, nullptr), "gbx_chain_host");
    const double dt = now_s() - t0;
    if (print && fo) {
        for (int64_t c = 0; c < nc; ++c) {
            fprintf(fo, "%lld\n", (long long)(poff[c + 1] - poff[c]));
            for (int64_t i = poff[c]; i < poff[c + 1]; ++i) fprintf(fo, "%d\t%d\n", score[i], parent[i]);
            fprintf(fo, "EOR\n");
        }
    }
    fprintf(stderr, "Time in kernel: %.2f sec\n", dt);
    printf("{\"benchmark\":\"chain\",\"calls\":%lld,\"anchors\":%lld,\"seconds\":%.6f,\"manchors_per_s\":%.3f}\n",
           (long long)nc, (long long)na, dt, na / dt / 1e6);
    if (fo) fclose(fo);
    return 0;
}

int main(int argc, char **argv)
{
    const int gpus = take_gpus_flag(argc, argv);
    std::string in, outp, cache;
    bool print = false, parse_only = false;
    int threads = 1;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "-i") && i + 1 < argc) in = argv[++i];
        else if (!strcmp(argv[i], "-o") && i + 1 < argc) outp = argv[++i];
        else if (!strcmp(argv[i], "-t") && i + 1 < argc) threads = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--print")) print = true;
        else if (!strcmp(argv[i], "--parse-only")) parse_only = true;
        else if (!strcmp(argv[i], "--cache") && i + 1 < argc) cache = argv[++i];
        else if (!strcmp(argv[i], "-h")) { help(); return 0; }
        else { help(); return 1; }
    }
    if (threads < 1) threads = 1;
    if (argc == 1) { help(); return EXIT_FAILURE; }
    fprintf(stderr, "Input file: %s\nOutput file: %s\n", in.c_str(), outp.c_str());
    // --cache FILE (round 6, SURVEY 8f rank 1 "binary cache"; driver_common.h: InputCache): the converted arrays - offsets, x, y,
    // headers - written after the first conversion, mapped by later runs of the same input instead of parsing a gigabyte of text
    InputCache icache;
    if (!cache.empty() && !parse_only && icache.open(cache.c_str(), in.c_str(), 0x6e696863 /* "chin" */, 4)) {
        const int64_t nc = (int64_t)(icache.sec[0].second / 8) - 1, na = (int64_t)(icache.sec[1].second / 8) - 1;
        if (nc < 0 || na < 0 || icache.sec[2].second != icache.sec[1].second || icache.sec[3].second != (size_t)nc * sizeof(gbx_chain_call)) {
            fprintf(stderr, "%s: malformed cache\n", cache.c_str());
            return EXIT_FAILURE;
        }
        fprintf(stderr, "Input arrays mapped from the cache %s\n", cache.c_str());
        return run_calls(gpus, nc, na, (const int64_t *)icache.sec[0].first, (const uint64_t *)icache.sec[1].first, (const uint64_t *)icache.sec[2].first,
                         (const gbx_chain_call *)icache.sec[3].first, print, fopen(outp.c_str(), "w"));
    }
    std::vector<char> text;
    if (!slurp(in.c_str(), text)) { fprintf(stderr, "cannot open %s\n", in.c_str()); return EXIT_FAILURE; }
    FILE *fo = fopen(outp.c_str(), "w");
    // ---- parallel ingest.  A call is "n avg_qspan max_dist_x max_dist_y bw n_segs", n anchor pairs "x y", then "EOR"
    // (read_call / skip_to_EOR, host_data_io.cpp:4-51); tokens are separated by any white space.  The "EOR" marks are
    // found by all threads over their own byte ranges, the headers give the anchor offsets, the anchors are converted
    // call by call.
    const double t_read0 = now_s();
    const char *tp = text.data();
    const size_t tn = text.size() - 1;
    std::vector<std::vector<size_t>> marks((size_t)threads);
#pragma omp parallel num_threads(threads)
    {
        const int t = omp_get_thread_num(), T = omp_get_num_threads();
        const size_t lo = tn * (size_t)t / (size_t)T, hi = tn * (size_t)(t + 1) / (size_t)T;
        for (const char *q = tp + lo, *e = tp + hi; q < e;) {
            const char *f = (const char *)memchr(q, 'E', (size_t)(e - q));
            if (!f) break;
            if ((size_t)(f - tp) + 2 < tn && f[1] == 'O' && f[2] == 'R') marks[(size_t)t].push_back((size_t)(f - tp));
            q = f + 1;
        }
    }
    std::vector<size_t> eor;
    for (auto &v : marks) eor.insert(eor.end(), v.begin(), v.end());
    const int64_t nc = (int64_t)eor.size();
    std::vector<int64_t> off((size_t)nc + 1, 0);
    std::vector<gbx_chain_call> hdr((size_t)nc);
    std::vector<const char *> body((size_t)nc);                 // first byte behind each header
    int bad = 0;
#pragma omp parallel for num_threads(threads) schedule(static) reduction(|:bad)
    for (int64_t c = 0; c < nc; ++c) {
        const char *q = tp + (c ? eor[(size_t)c - 1] + 3 : 0);
        char *e = nullptr;
        const long long n = strtoll(q, &e, 10);
        if (e == q) { bad |= 1; continue; }
        gbx_chain_call h;
        q = e; h.avg_qspan = strtof(q, &e); if (e == q) bad |= 1;
        q = e; h.max_dist_x = (int)strtol(q, &e, 10); if (e == q) bad |= 1;
        q = e; h.max_dist_y = (int)strtol(q, &e, 10); if (e == q) bad |= 1;
        q = e; h.bw = (int)strtol(q, &e, 10); if (e == q) bad |= 1;
        q = e; h.n_segs = (int)strtol(q, &e, 10); if (e == q) bad |= 1;
        hdr[(size_t)c] = h; off[(size_t)c + 1] = n < 0 ? 0 : n; body[(size_t)c] = e;
    }
    if (bad) { fprintf(stderr, "malformed call header\n"); return EXIT_FAILURE; }
    for (int64_t c = 0; c < nc; ++c) off[(size_t)c + 1] += off[(size_t)c];
    const int64_t na = off[(size_t)nc];
    std::vector<uint64_t> ax((size_t)na + 1), ay((size_t)na + 1);
#pragma omp parallel for num_threads(threads) schedule(dynamic, 4) reduction(|:bad)
    for (int64_t c = 0; c < nc; ++c) {
        const char *q = body[(size_t)c], *end = tp + eor[(size_t)c];
        auto next_u64 = [&](uint64_t &v) -> bool {
            while (q < end && (unsigned char)(*q - '0') > 9) ++q;
            if (q >= end) return false;
            uint64_t x = 0;
            while (q < end && (unsigned char)(*q - '0') <= 9) x = x * 10 + (uint64_t)(*q++ - '0');
            v = x;
            return true;
        };
        for (int64_t k = off[(size_t)c]; k < off[(size_t)c + 1]; ++k)
            if (!next_u64(ax[(size_t)k]) || !next_u64(ay[(size_t)k])) { bad |= 1; break; }
    }
    if (bad) { fprintf(stderr, "truncated call\n"); return EXIT_FAILURE; }
    const double t_read = now_s() - t_read0;
    if (parse_only) {
        uint64_t h = fnv1a(off.data(), (size_t)(nc + 1) * 8);
        for (int64_t c = 0; c < nc; ++c) {
            const int32_t w[4] = {hdr[(size_t)c].max_dist_x, hdr[(size_t)c].max_dist_y, hdr[(size_t)c].bw, hdr[(size_t)c].n_segs};
            h = fnv1a(&hdr[(size_t)c].avg_qspan, 4, h); h = fnv1a(w, 16, h);
        }
        h = fnv1a(ax.data(), (size_t)na * 8, h); h = fnv1a(ay.data(), (size_t)na * 8, h);
        printf("{\"benchmark\":\"chain\",\"calls\":%lld,\"anchors\":%lld,\"ingest_threads\":%d,\"ingest_seconds\":%.4f,\"ingest_mb_per_s\":%.1f,\"checksum\":\"%016llx\"}\n",
               (long long)nc, (long long)na, threads, t_read, tn / 1e6 / t_read, (unsigned long long)h);
        return 0;
    }
    fprintf(stderr, "Ingest: %.2f s with %d thread(s)\n", t_read, threads);
    if (!cache.empty() && !InputCache::write(cache.c_str(), in.c_str(), 0x6e696863, {{off.data(), (size_t)(nc + 1) * 8}, {ax.data(), (size_t)(na + 1) * 8},
                                                                                     {ay.data(), (size_t)(na + 1) * 8}, {hdr.data(), (size_t)nc * sizeof(gbx_chain_call)}}))
        fprintf(stderr, "warning: could not write the cache %s\n", cache.c_str());
    return run_calls(gpus, nc, na, off.data(), ax.data(), ay.data(), hdr.data(), print, fo);
}
