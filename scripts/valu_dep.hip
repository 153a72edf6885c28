// valu_dep — issue-to-issue latency of DEPENDENT VALU instructions on gfx950, by distance between dependent instructions.
//
//   hipcc -O2 --offload-arch=gfx950 scripts/valu_dep.hip -o scripts/valu_dep && ./scripts/valu_dep > profiles/valu_dep.json
//
// scripts/valu_peak.hip measures throughput with 8 independent accumulators.  Here a loop trip is 64 instructions over D
// accumulators used round-robin (D = 1: every instruction reads the result of the one before it; D = 2: of the one two
// before, ...), run at 1, 2 and 3 wavefronts per SIMD on all CUs.  Reported: wall cycles per instruction of ONE wavefront
// (= its critical path per instruction) and per SIMD.  Nothing here is linked into the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

#define ADD(k)  "v_add_u32 %" #k ", %" #k ", %8\n"
#define MAX3(k) "v_max3_i32 %" #k ", %" #k ", %8, %9\n"
#define SDWA(k) "v_add_u32_sdwa %" #k ", %" #k ", sext(%8) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
#define LSHLOR(k) "v_lshl_or_b32 %" #k ", %" #k ", 1, %8\n"
#define R1(I) I(0) I(0) I(0) I(0) I(0) I(0) I(0) I(0)
#define R2(I) I(0) I(1) I(0) I(1) I(0) I(1) I(0) I(1)
#define R3(I) I(0) I(1) I(2) I(0) I(1) I(2) I(0) I(1)
#define R4(I) I(0) I(1) I(2) I(3) I(0) I(1) I(2) I(3)
#define R8(I) I(0) I(1) I(2) I(3) I(4) I(5) I(6) I(7)
#define BODY(R, I) asm volatile(R(I) R(I) R(I) R(I) R(I) R(I) R(I) R(I) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
#define KERNEL(NAME, R, I)                                                                                             \
    __global__ void __launch_bounds__(256) NAME(int iters, unsigned *out, unsigned long long *ticks)                   \
    {                                                                                                                  \
        unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7, \
                 b = blockIdx.x | 0x01020304u, c = 3u + threadIdx.x;                                                   \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                          \
        for (int i = 0; i < iters; ++i) { BODY(R, I) }                                                                 \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                          \
        out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;                                   \
        if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;                            \
    }
KERNEL(add_d1, R1, ADD) KERNEL(add_d2, R2, ADD) KERNEL(add_d3, R3, ADD) KERNEL(add_d4, R4, ADD) KERNEL(add_d8, R8, ADD)
KERNEL(max3_d1, R1, MAX3) KERNEL(max3_d2, R2, MAX3) KERNEL(max3_d3, R3, MAX3) KERNEL(max3_d4, R4, MAX3) KERNEL(max3_d8, R8, MAX3)
KERNEL(sdwa_d1, R1, SDWA) KERNEL(sdwa_d2, R2, SDWA) KERNEL(sdwa_d4, R4, SDWA) KERNEL(sdwa_d8, R8, SDWA)
KERNEL(lshlor_d1, R1, LSHLOR) KERNEL(lshlor_d2, R2, LSHLOR) KERNEL(lshlor_d4, R4, LSHLOR)
typedef void (*kern_t)(int, unsigned *, unsigned long long *);
struct Form { const char *name; kern_t k; };
#define F(n) {#n, n}
static const Form forms[] = {F(add_d1), F(add_d2), F(add_d3), F(add_d4), F(add_d8), F(max3_d1), F(max3_d2), F(max3_d3), F(max3_d4), F(max3_d8),
                             F(sdwa_d1), F(sdwa_d2), F(sdwa_d4), F(sdwa_d8), F(lshlor_d1), F(lshlor_d2), F(lshlor_d4)};
int main(int argc, char **argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 4000;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    unsigned *out;
    unsigned long long *ticks;
    CHECK(hipMalloc(&out, (size_t)cus * 4 * 256 * 4));
    CHECK(hipMalloc(&ticks, (size_t)cus * 4 * 4 * 8));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    printf("{\"gcn_arch\": \"%s\", \"cus\": %d, \"clock_khz\": %d, \"command\": \"./scripts/valu_dep %d\",\n \"note\": \"name_dD: D accumulators round-robin = "
           "an instruction reads the result of the one D before it; wN = N wavefronts per SIMD; cycles per instruction of one wavefront (wall / its "
           "instructions) and per SIMD (that / N)\",\n \"forms\": {\n", prop.gcnArchName, cus, prop.clockRate, iters);
    bool first = true;
    for (const Form &f : forms) {
        printf("%s  \"%s\": {", first ? "" : ",\n", f.name);
        first = false;
        for (int w = 1; w <= 3; ++w) {
            const int blocks = cus * w;
            hipLaunchKernelGGL(f.k, dim3(blocks), dim3(256), 0, 0, 64, out, ticks);
            CHECK(hipDeviceSynchronize());
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                CHECK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL(f.k, dim3(blocks), dim3(256), 0, 0, iters, out, ticks);
                CHECK(hipEventRecord(e1, 0));
                CHECK(hipEventSynchronize(e1));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                best = ms < best ? ms : best;
            }
            const double cyc_wave = (best * 1e-3) * prop.clockRate * 1e3 / ((double)iters * 64);
            printf("%s\"w%d\": {\"cyc_per_inst_wave\": %.2f, \"cyc_per_inst_simd\": %.2f}", w > 1 ? ", " : "", w, cyc_wave, cyc_wave / w);
        }
        printf("}");
        fflush(stdout);
    }
    printf("\n }\n}\n");
    return 0;
}
