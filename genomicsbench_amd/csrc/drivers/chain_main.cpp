// chain — GPU driver with the CLI of R/benchmarks/chain/src/main.cpp:  chain -i <in> -o <out> [-t T] [-h]
// Input: host_data_io.cpp:13-51 (header `n avg_qspan max_dist_x max_dist_y bw n_segs`, n lines `x y`, `EOR`).
// Like the reference the output file is opened "w"; results are written (print_return format,
// host_data_io.cpp:53-60) when --print is given (the reference needs a PRINT_OUTPUT rebuild for that).
#include <unistd.h>
#include "driver_common.h"

static void help() { fprintf(stderr, "usage: chain -i <input> -o <output> [-t threads] [--print]\n"); }

int main(int argc, char **argv)
{
    std::string in, outp;
    bool print = false;
    int threads = 1;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "-i") && i + 1 < argc) in = argv[++i];
        else if (!strcmp(argv[i], "-o") && i + 1 < argc) outp = argv[++i];
        else if (!strcmp(argv[i], "-t") && i + 1 < argc) threads = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--print")) print = true;
        else if (!strcmp(argv[i], "-h")) { help(); return 0; }
        else { help(); return 1; }
    }
    (void)threads;
    if (argc == 1) { help(); return EXIT_FAILURE; }
    fprintf(stderr, "Input file: %s\nOutput file: %s\n", in.c_str(), outp.c_str());
    FILE *fi = fopen(in.c_str(), "r");
    if (!fi) { fprintf(stderr, "cannot open %s\n", in.c_str()); return EXIT_FAILURE; }
    FILE *fo = fopen(outp.c_str(), "w");
    std::vector<int64_t> off(1, 0);
    std::vector<uint64_t> ax, ay;
    std::vector<gbx_chain_call> hdr;
    for (;;) {                                                  // read_call until the header no longer parses
        long long n; gbx_chain_call h;
        if (fscanf(fi, "%lld%f%d%d%d%d", &n, &h.avg_qspan, &h.max_dist_x, &h.max_dist_y, &h.bw, &h.n_segs) != 6) break;
        for (long long k = 0; k < n; ++k) {
            unsigned long long x, y;
            if (fscanf(fi, "%llu%llu", &x, &y) != 2) { fprintf(stderr, "truncated call\n"); return EXIT_FAILURE; }
            ax.push_back(x); ay.push_back(y);
        }
        for (const char *loc = "EOR"; *loc;) { int ch = fgetc(fi); if (ch == EOF) break; if (ch == *loc) ++loc; }
        hdr.push_back(h); off.push_back((int64_t)ax.size());
    }
    fclose(fi);
    const int64_t nc = (int64_t)hdr.size(), na = (int64_t)ax.size();
    print_device_banner();
    std::vector<int32_t> score((size_t)na + 1), parent((size_t)na + 1);
    if (nc > 0) {                                               // warm-up on the first call only
        int64_t o2[2] = {0, off[1]};
        die_on(gbx_chain_host(1, o2, ax.data(), ay.data(), hdr.data(), score.data(), parent.data(), nullptr, nullptr), "gbx_chain_host");
    }
    const double t0 = now_s();
    die_on(gbx_chain_host(nc, off.data(), ax.data(), ay.data(), hdr.data(), score.data(), parent.data(), nullptr, nullptr), "gbx_chain_host");
    const double dt = now_s() - t0;
    if (print && fo) {
        for (int64_t c = 0; c < nc; ++c) {
            fprintf(fo, "%lld\n", (long long)(off[c + 1] - off[c]));
            for (int64_t i = off[c]; i < off[c + 1]; ++i) fprintf(fo, "%d\t%d\n", score[i], parent[i]);
            fprintf(fo, "EOR\n");
        }
    }
    fprintf(stderr, "Time in kernel: %.2f sec\n", dt);
    printf("{\"benchmark\":\"chain\",\"calls\":%lld,\"anchors\":%lld,\"seconds\":%.6f,\"manchors_per_s\":%.3f}\n",
           (long long)nc, (long long)na, dt, na / dt / 1e6);
    if (fo) fclose(fo);
    return 0;
}
