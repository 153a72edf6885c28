// gather_peak — what HBM delivers on MI355X to the access pattern of fmi's backwardExt: 64-byte lines at random places of
// a 1 GB table, one line per quad of lanes (16 bytes per lane), two lines per step, the next step's addresses depending on
// what was loaded (the k / l of an FM-index interval).  The ceiling fmi_smem_kernel's 4.8 TB/s of line requests is to be
// read against - not the 8 TB/s of streaming.
//
//   hipcc -O2 --offload-arch=gfx950 scripts/gather_peak.hip -o scripts/gather_peak && ./scripts/gather_peak > profiles/gather_peak.json
//
// Forms: "independent" = addresses from a counter-based hash (as many loads in flight as the unroll allows);
// "dependent" = the next pair of lines is computed from the loaded words (one round trip per step and quad, hidden only
// by the other wavefronts).  Reported: line requests x 64 B / wall time, per occupancy (wavefronts per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

__device__ inline unsigned mix(unsigned x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// lines: number of 64-byte lines in the table (a power of two)
template <bool DEP, int UNROLL>
__global__ void __launch_bounds__(64) gather(const uint4 *__restrict__ table, unsigned lines_mask, int steps, unsigned *out)
{
    const unsigned quad = (blockIdx.x * 64 + threadIdx.x) >> 2, sub = threadIdx.x & 3;
    unsigned acc = 0, state = mix(quad * 2654435761u + 12345u);
    for (int s = 0; s < steps; s += UNROLL) {
        uint4 v[UNROLL][2];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            // two lines per step: "k" anywhere, "l" a short distance behind it (an interval's two ends)
            const unsigned a = DEP ? state : mix(state + (unsigned)(s + u) * 0x9e3779b9u);
            const unsigned k = a & lines_mask, l = (k + 1 + (a >> 27)) & lines_mask;
            v[u][0] = table[(size_t)k * 4 + sub];
            v[u][1] = table[(size_t)l * 4 + sub];
            if (DEP) {
                // the loaded words of the whole quad decide the next lines (xor over the quad = two DPP steps in fmi's kernel too)
                unsigned w = v[u][0].x ^ v[u][1].y ^ v[u][0].z;
                w ^= __shfl_xor(w, 1); w ^= __shfl_xor(w, 2);
                state = mix(state ^ w);
            }
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc += v[u][0].x ^ v[u][0].w ^ v[u][1].y ^ v[u][1].z;
        if (!DEP) state += acc & 1u;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <bool DEP, int UNROLL>
static double run(const uint4 *table, unsigned lines, int waves_per_simd, int steps, unsigned *out)
{
    int cus = 256;
    CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    const int blocks = cus * 4 * waves_per_simd;
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL((gather<DEP, UNROLL>), dim3(blocks), dim3(64), 0, 0, table, lines - 1, steps / 4, out);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL((gather<DEP, UNROLL>), dim3(blocks), dim3(64), 0, 0, table, lines - 1, steps, out);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    const double requests = (double)blocks * 16 /* quads */ * steps * 2;
    return requests * 64.0 / (ms * 1e-3) / 1e9;
}

int main()
{
    const size_t bytes = (size_t)1 << 30;
    uint4 *table; unsigned *out;
    CHECK(hipMalloc(&table, bytes)); CHECK(hipMalloc(&out, 64));
    std::vector<unsigned> h(bytes / 4);
    unsigned x = 1;
    for (auto &w : h) { x = x * 1664525u + 1013904223u; w = x; }
    CHECK(hipMemcpy(table, h.data(), bytes, hipMemcpyHostToDevice));
    const unsigned lines = (unsigned)(bytes / 64);
    printf("{\"what\": \"random 64-byte line gathers over a 1 GB table, one line per quad, two lines per step (scripts/gather_peak.hip)\",\n \"unit\": \"GB/s of requested lines\", \"rows\": [\n");
    bool first = true;
    for (int w : {2, 4, 6, 8}) {
        const double i1 = run<false, 1>(table, lines, w, 4096, out), i4 = run<false, 4>(table, lines, w, 4096, out);
        const double d1 = run<true, 1>(table, lines, w, 4096, out);
        printf("%s  {\"waves_per_simd\": %d, \"independent_1_step_in_flight\": %.0f, \"independent_4_steps_in_flight\": %.0f, \"dependent\": %.0f}", first ? "" : ",\n", w, i1, i4, d1);
        first = false;
    }
    printf("\n]}\n");
    return 0;
}
