// driver_common.h — helpers shared by the four benchmark drivers.  The drivers
// keep the reference CLIs (R/benchmarks/<name>/...) and talk to the GPU only
// through the C-ABI of libgbx.so (include/gbx.h): no HIP, no torch here.
#pragma once
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../../../include/gbx.h"

static inline double now_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static inline void die_on(int rc, const char *what)
{
    if (rc != GBX_OK) {
        fprintf(stderr, "%s failed (%d): %s\n", what, rc, gbx_last_error());
        exit(EXIT_FAILURE);
    }
}

// whole file into memory (input files are parsed once, outside the timed region, like the reference)
static inline bool slurp(const char *path, std::vector<char> &buf)
{
    FILE *f = fopen(path, "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    buf.resize((size_t)n + 1);
    size_t got = n > 0 ? fread(buf.data(), 1, (size_t)n, f) : 0;
    buf[got] = 0;
    buf.resize(got + 1);
    fclose(f);
    return true;
}

// ---- parallel ingest: the input text is split and converted by `threads` OpenMP threads.
// Chunk boundaries never cut a record: every thread finds the delimiters of its own byte range, the ranges
// are stitched by a prefix sum, and the records are then parsed independently.
#include <omp.h>

// An array whose pages are first touched by whoever fills it (std::vector's resize() zero-fills on the calling thread: for the
// 0.4 GB of arrays of a 2 M-pair bsw job that was most of the ingest).  Trivial element types only.
template <class T> struct RawVec {
    T *p = nullptr; size_t n = 0;
    RawVec() {}
    explicit RawVec(size_t k) { resize(k); }
    RawVec(const RawVec &) = delete;
    ~RawVec() { free(p); }
    void resize(size_t k) { free(p); p = k ? (T *)malloc(k * sizeof(T)) : nullptr; n = k; if (k && !p) { fprintf(stderr, "out of memory\n"); exit(EXIT_FAILURE); } }
    T *data() { return p; } const T *data() const { return p; }
    size_t size() const { return n; }
    T &operator[](size_t k) { return p[k]; } const T &operator[](size_t k) const { return p[k]; }
};

// start and length of every '\n'-terminated line of [p, p+n) (LV / IV: std::vector or RawVec of const char * / int)
template <class LV, class IV>
static inline void split_lines(const char *p, size_t n, int threads, LV &line, IV &llen)
{
    if (threads < 1) threads = 1;
    std::vector<std::vector<size_t>> nl((size_t)threads);
#pragma omp parallel num_threads(threads)
    {
        const int t = omp_get_thread_num(), T = omp_get_num_threads();
        const size_t lo = n * (size_t)t / (size_t)T, hi = n * (size_t)(t + 1) / (size_t)T;
        std::vector<size_t> &v = nl[(size_t)t];
        v.reserve((hi - lo) / 24 + 64);                       // (a guess that spares most of the regrowth; any line length works)
        for (const char *q = p + lo, *e = p + hi; q < e;) {
            const char *f = (const char *)memchr(q, '\n', (size_t)(e - q));
            if (!f) break;
            v.push_back((size_t)(f - p));
            q = f + 1;
        }
    }
    size_t total = 0;
    std::vector<size_t> base((size_t)threads + 1, 0);
    for (int t = 0; t < threads; ++t) { base[(size_t)t] = total; total += nl[(size_t)t].size(); }
    line.resize(total); llen.resize(total);
#pragma omp parallel for num_threads(threads) schedule(static)
    for (int t = 0; t < threads; ++t) {
        const std::vector<size_t> &v = nl[(size_t)t];
        for (size_t k = 0; k < v.size(); ++k) {
            const size_t g = base[(size_t)t] + k;
            // the line starts behind the previous newline, which may belong to an earlier thread's range
            size_t start = 0;
            if (k > 0) start = v[k - 1] + 1;
            else for (int u = t - 1; u >= 0; --u) if (!nl[(size_t)u].empty()) { start = nl[(size_t)u].back() + 1; break; }
            line[g] = p + start; llen[g] = (int)(v[k] - start);
        }
    }
}

// 64-bit FNV-1a over a byte range: the checksum `--parse-only` prints, so that tests can compare the parsed
// arrays with an independent reader without a GPU
static inline uint64_t fnv1a(const void *data, size_t bytes, uint64_t h = 1469598103934665603ull)
{
    const unsigned char *b = (const unsigned char *)data;
    for (size_t k = 0; k < bytes; ++k) { h ^= b[k]; h *= 1099511628211ull; }
    return h;
}

// device banner + the library's one-time host-side setup (streams, pinned staging buffers), which the drivers
// keep out of their timed regions like the reference keeps its object construction out of them
// `--gpus N` (every driver; not a reference flag): the host entries spread each call over N devices of this node
// (gbx_host_set_devices: contiguous cost-balanced ranges of the units, one host lane per device).  Takes the flag out of
// argv so that the reference-style option loops never see it; 0 = not given (GBX_GPUS, else one device).
static inline int take_gpus_flag(int &argc, char **argv)
{
    int gpus = 0;
    for (int i = 1; i < argc;) {
        if (!strcmp(argv[i], "--gpus") && i + 1 < argc) {
            gpus = atoi(argv[i + 1]);
            if (gpus < 1) { fprintf(stderr, "--gpus needs a positive device count\n"); exit(EXIT_FAILURE); }
            for (int j = i; j + 2 < argc; ++j) argv[j] = argv[j + 2];
            argc -= 2;
        } else ++i;
    }
    return gpus;
}

static inline void print_device_banner(int gpus = 0)
{
    char name[256];
    if (gpus > 0) die_on(gbx_host_set_devices(gpus), "gbx_host_set_devices");
    if (gbx_device_name(name, sizeof(name)) == GBX_OK) fprintf(stderr, "gbx device: %s x %d\n", name, gbx_host_devices());
    die_on(gbx_host_prepare(), "gbx_host_prepare");
}
