// lds_peak — measured LDS instruction rates of gfx950 for the access forms the lane-per-pair bsw kernel uses
// (column-major [column][lane] planes: every lane its own bank whatever column it is at).
//
//   hipcc -O2 --offload-arch=gfx950 scripts/lds_peak.hip -o scripts/lds_peak && ./scripts/lds_peak > profiles/lds_peak.json
//
// One-wavefront workgroups, W of them per CU (W = 4, 8, 16), each looping over a private 16 KB LDS region with long
// unrolled runs of one instruction form (independent addresses, results folded into a checksum once per 16).
// Reported: LDS cycles per wave-instruction per CU (wall time x clock / instructions issued on the CU).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

enum Form { R8, R16, R32, R64, W8, W16, W32, W64, RW16 /* read u16 + read u8 + write b16: the compact cell step */, RW32, NFORM };
static const char *form_name[NFORM] = {"ds_read_u8", "ds_read_u16", "ds_read_b32", "ds_read_b64", "ds_write_b8", "ds_write_b16", "ds_write_b32",
                                       "ds_write_b64", "cell_step_u16_u8_w16", "cell_step_b32_w32"};
static const int form_ops[NFORM] = {1, 1, 1, 1, 1, 1, 1, 1, 3, 2};

template <int F>
__global__ void __launch_bounds__(64) k(int iters, unsigned *out)
{
    extern __shared__ unsigned char lds[];
    const int lane = threadIdx.x;
    unsigned acc = 0;
    // [column][lane] planes: cells of 1 / 2 / 4 / 8 bytes per lane
    for (int i = lane; i < 4096; i += 64) ((unsigned *)lds)[i] = i;
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int col = (it + u) & 31;
            if (F == R8) acc += lds[col * 64 + lane];
            if (F == R16) acc += ((unsigned short *)lds)[col * 64 + lane];
            if (F == R32) acc += ((unsigned *)lds)[col * 64 + lane];
            if (F == R64) { const uint2 v = ((uint2 *)lds)[col * 64 + lane]; acc += v.x ^ v.y; }
            if (F == W8) lds[col * 64 + lane] = (unsigned char)(acc + u);
            if (F == W16) ((unsigned short *)lds)[col * 64 + lane] = (unsigned short)(acc + u);
            if (F == W32) ((unsigned *)lds)[col * 64 + lane] = acc + u;
            if (F == W64) ((uint2 *)lds)[col * 64 + lane] = make_uint2(acc + u, acc);
            if (F == RW16) {
                const unsigned c = ((unsigned short *)lds)[col * 64 + lane], q = lds[8192 + col * 64 + lane];
                ((unsigned short *)lds)[((col + 7) & 31) * 64 + lane] = (unsigned short)(c + q + acc);
                acc += c ^ q;
            }
            if (F == RW32) {
                const unsigned c = ((unsigned *)lds)[col * 64 + lane];
                ((unsigned *)lds)[((col + 7) & 31) * 64 + lane] = c + acc;
                acc += c;
            }
        }
        asm volatile("" : "+v"(acc));
    }
    out[blockIdx.x * 64 + lane] = acc;
}

typedef void (*kern_t)(int, unsigned *);
static kern_t kerns[NFORM] = {k<R8>, k<R16>, k<R32>, k<R64>, k<W8>, k<W16>, k<W32>, k<W64>, k<RW16>, k<RW32>};

int main(int argc, char **argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    unsigned *out;
    CHECK(hipMalloc(&out, (size_t)cus * 32 * 64 * 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_khz\": %d, \"iters\": %d, \"command\": \"./scripts/lds_peak %d\",\n \"note\": \"one-wavefront workgroups, "
           "W per CU; cyc_per_inst_cu = wall x clock / (W x instructions per wavefront); cell_step forms count 3 / 2 instructions per step\",\n \"forms\": {\n",
           prop.name, cus, prop.clockRate, iters, iters);
    for (int f = 0; f < NFORM; ++f) {
        printf("%s  \"%s\": {", f ? ",\n" : "", form_name[f]);
        const int occ[4] = {4, 8, 16, 32};
        for (int o = 0; o < 4; ++o) {
            const size_t lds = (size_t)160 * 1024 / occ[o] - 512;                // pins the residency to occ[o] workgroups per CU
            CHECK(hipFuncSetAttribute((const void *)kerns[f], hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            const int blocks = cus * occ[o];
            hipLaunchKernelGGL(kerns[f], dim3(blocks), dim3(64), lds, 0, 16, out);
            CHECK(hipDeviceSynchronize());
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                CHECK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL(kerns[f], dim3(blocks), dim3(64), lds, 0, iters, out);
                CHECK(hipEventRecord(e1, 0));
                CHECK(hipEventSynchronize(e1));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            const double insts_per_wave = (double)iters * 16 * form_ops[f];
            const double cyc = best * 1e-3 * prop.clockRate * 1e3 / (insts_per_wave * occ[o]);
            printf("%s\"w%d\": {\"ms\": %.4f, \"cyc_per_inst_cu\": %.3f}", o ? ", " : "", occ[o], best, cyc);
        }
        printf("}");
        fflush(stdout);
    }
    printf("\n }\n}\n");
    return 0;
}
