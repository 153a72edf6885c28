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

// ---- binary input cache ("optional"): the converted arrays of an input file, written once beside it and
// mapped read-only by later runs instead of converting the text again.  One file: a 64-byte header - magic, the source
// file's size and modification time (a cache of another or an edited input is ignored), the section count - then the
// sections, each 4 KiB-aligned so that a mapped section is page-aligned: {bytes, payload}.  The caller names the sections'
// meaning by their order.  Nothing of the reference corresponds to this; its drivers convert their text on every run.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
struct InputCache {
    static constexpr uint64_t MAGIC = 0x3143584247ull;        // "GBXC1"
    struct Header { uint64_t magic, kind, src_size, src_mtime_ns, n_sections, pad[3]; };
    char *base = nullptr; size_t bytes = 0;
    std::vector<std::pair<char *, size_t>> sec;
    static bool source_stamp(const char *src, uint64_t *size, uint64_t *mtime_ns)
    {
        struct stat st;
        if (stat(src, &st) != 0) return false;
        *size = (uint64_t)st.st_size;
        *mtime_ns = (uint64_t)st.st_mtim.tv_sec * 1000000000ull + (uint64_t)st.st_mtim.tv_nsec;
        return true;
    }
    // maps `path` if it is a cache of `src` as it is now and of this kind with this many sections
    bool open(const char *path, const char *src, uint64_t kind, size_t want_sections)
    {
        uint64_t ssz = 0, smt = 0;
        if (!source_stamp(src, &ssz, &smt)) return false;
        const int fd = ::open(path, O_RDONLY);
        if (fd < 0) return false;
        struct stat st;
        if (fstat(fd, &st) != 0 || (size_t)st.st_size < sizeof(Header)) { ::close(fd); return false; }
        void *m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
        ::close(fd);
        if (m == MAP_FAILED) return false;
        base = (char *)m; bytes = (size_t)st.st_size;
        const Header *h = (const Header *)base;
        bool ok = h->magic == MAGIC && h->kind == kind && h->src_size == ssz && h->src_mtime_ns == smt && h->n_sections == want_sections;
        size_t at = 4096;
        for (size_t k = 0; ok && k < want_sections; ++k) {
            if (at + 8 > bytes) { ok = false; break; }
            const uint64_t len = *(const uint64_t *)(base + at);
            if (at + 4096 + len > bytes) { ok = false; break; }
            sec.emplace_back(base + at + 4096, (size_t)len);
            at += 4096 + ((len + 4095) & ~(uint64_t)4095);
        }
        if (!ok) { close(); return false; }
        (void)madvise(base, bytes, MADV_WILLNEED);
        return true;
    }
    void close() { if (base) munmap(base, bytes); base = nullptr; bytes = 0; sec.clear(); }
    ~InputCache() { close(); }
    // writes the cache (to a temporary name, renamed into place: a reader never sees half a file)
    static bool write(const char *path, const char *src, uint64_t kind, const std::vector<std::pair<const void *, size_t>> &sections)
    {
        Header h;
        memset(&h, 0, sizeof h);
        h.magic = MAGIC; h.kind = kind; h.n_sections = sections.size();
        if (!source_stamp(src, &h.src_size, &h.src_mtime_ns)) return false;
        const std::string tmp = std::string(path) + ".tmp";
        FILE *f = fopen(tmp.c_str(), "wb");
        if (!f) return false;
        std::vector<char> zero(4096, 0);
        bool ok = fwrite(&h, sizeof h, 1, f) == 1 && fwrite(zero.data(), 4096 - sizeof h, 1, f) == 1;
        for (const auto &s : sections) {
            const uint64_t len = s.second;
            ok = ok && fwrite(&len, 8, 1, f) == 1 && fwrite(zero.data(), 4096 - 8, 1, f) == 1;
            ok = ok && (len == 0 || fwrite(s.first, 1, (size_t)len, f) == (size_t)len);
            const size_t pad = (size_t)(((len + 4095) & ~(uint64_t)4095) - len);
            ok = ok && (pad == 0 || fwrite(zero.data(), 1, pad, f) == pad);
        }
        ok = fclose(f) == 0 && ok;
        if (ok) ok = rename(tmp.c_str(), path) == 0;
        if (!ok) remove(tmp.c_str());
        return ok;
    }
};

// ---- parallel ingest: the input text is split and converted by `threads` OpenMP threads.
// Chunk boundaries never cut a record: every thread finds the delimiters of its own byte range, the ranges
// are stitched by a prefix sum, and the records are then parsed independently.
#include <omp.h>

// An array whose pages are first touched by whoever fills it (std::vector's resize() zero-fills on the calling thread: for the
// 0.4 GB of arrays of a 2 M-pair bsw job that was most of the ingest).  Trivial element types only.
template <class T> struct RawVec {
    T *p = nullptr; size_t n = 0;
    bool own = true;                       // false: a view of memory someone else holds (adopt(): a mapped input cache)
    RawVec() {}
    explicit RawVec(size_t k) { resize(k); }
    RawVec(const RawVec &) = delete;
    ~RawVec() { if (own) free(p); }
    void resize(size_t k) { if (own) free(p); own = true; p = k ? (T *)malloc(k * sizeof(T)) : nullptr; n = k; if (k && !p) { fprintf(stderr, "out of memory\n"); exit(EXIT_FAILURE); } }
    void adopt(T *q, size_t k) { if (own) free(p); own = false; p = q; n = k; }
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
