// h2d_probe — what the host-buffer entry points can expect from this box: allocation cost, pageable / pinned /
// staged host-to-device rates.  Build: hipcc -O2 --offload-arch=gfx950 scripts/h2d_probe.hip -o /tmp/h2d_probe -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static void par_copy(char *dst, const char *src, size_t n, int T)
{
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t)
        th.emplace_back([=] { size_t lo = n * t / T, hi = n * (t + 1) / T; memcpy(dst + lo, src + lo, hi - lo); });
    for (auto &x : th) x.join();
}

int main(int argc, char **argv)
{
    const size_t MB = 1 << 20, N = (argc > 1 ? atoi(argv[1]) : 600) * MB;
    char *h = (char *)malloc(N);
    memset(h, 1, N);
    void *d = nullptr;
    double t = now(); CK(hipMalloc(&d, N)); printf("hipMalloc %zu MB: %.2f ms\n", N / MB, (now() - t) * 1e3);
    t = now(); CK(hipFree(d)); printf("hipFree: %.2f ms\n", (now() - t) * 1e3);
    t = now(); CK(hipMalloc(&d, N)); printf("hipMalloc again: %.2f ms\n", (now() - t) * 1e3);
    for (int r = 0; r < 3; ++r) {
        t = now(); CK(hipMemcpy(d, h, N, hipMemcpyHostToDevice)); double dt = now() - t;
        printf("pageable H2D: %.2f ms  %.1f GB/s\n", dt * 1e3, N / dt / 1e9);
    }
    char *pin = nullptr;
    t = now(); CK(hipHostMalloc((void **)&pin, N, hipHostMallocDefault)); printf("hipHostMalloc %zu MB: %.2f ms\n", N / MB, (now() - t) * 1e3);
    for (int T : {1, 4, 8, 16, 32}) {
        t = now(); par_copy(pin, h, N, T); double dt = now() - t;
        printf("memcpy->pinned %d threads: %.2f ms  %.1f GB/s\n", T, dt * 1e3, N / dt / 1e9);
    }
    for (int r = 0; r < 3; ++r) {
        t = now(); CK(hipMemcpy(d, pin, N, hipMemcpyHostToDevice)); double dt = now() - t;
        printf("pinned H2D: %.2f ms  %.1f GB/s\n", dt * 1e3, N / dt / 1e9);
    }
    // staged: ring of slabs, T threads fill, async DMA
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    for (size_t slab : {8 * MB, 32 * MB}) for (int T : {4, 8, 16}) {
        const int R = 4;
        hipEvent_t ev[R]; for (int k = 0; k < R; ++k) CK(hipEventCreateWithFlags(&ev[k], hipEventDisableTiming));
        t = now();
        int k = 0;
        for (size_t off = 0; off < N; off += slab, ++k) {
            const size_t len = N - off < slab ? N - off : slab;
            char *sl = pin + (size_t)(k % R) * slab;
            if (k >= R) CK(hipEventSynchronize(ev[k % R]));
            par_copy(sl, h + off, len, T);
            CK(hipMemcpyAsync((char *)d + off, sl, len, hipMemcpyHostToDevice, s));
            CK(hipEventRecord(ev[k % R], s));
        }
        CK(hipStreamSynchronize(s));
        double dt = now() - t;
        printf("staged slab %zu MB, %d threads: %.2f ms  %.1f GB/s\n", slab / MB, T, dt * 1e3, N / dt / 1e9);
    }
    t = now(); CK(hipHostRegister(h, N, hipHostRegisterDefault)); printf("hipHostRegister: %.2f ms\n", (now() - t) * 1e3);
    t = now(); CK(hipMemcpy(d, h, N, hipMemcpyHostToDevice)); printf("registered H2D: %.2f ms\n", (now() - t) * 1e3);
    t = now(); CK(hipHostUnregister(h)); printf("hipHostUnregister: %.2f ms\n", (now() - t) * 1e3);
    const size_t O = 48 * MB;
    for (int r = 0; r < 2; ++r) { t = now(); CK(hipMemcpy(h, d, O, hipMemcpyDeviceToHost)); double dt = now() - t; printf("pageable D2H 48 MB: %.2f ms %.1f GB/s\n", dt * 1e3, O / dt / 1e9); }
    for (int r = 0; r < 2; ++r) { t = now(); CK(hipMemcpy(pin, d, O, hipMemcpyDeviceToHost)); double dt = now() - t; printf("pinned D2H 48 MB: %.2f ms %.1f GB/s\n", dt * 1e3, O / dt / 1e9); }
    return 0;
}
