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

static inline void print_device_banner()
{
    char name[256];
    if (gbx_device_name(name, sizeof(name)) == GBX_OK) fprintf(stderr, "gbx device: %s\n", name);
}
