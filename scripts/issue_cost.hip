// Single-wavefront issue cost of dependent instruction chains on gfx950 (clocks per instruction by s_memtime), the
// numbers behind the per-band (abea) and per-chunk (chain) estimates in DESIGN.md; where the wavefronts of a workgroup land.  build: hipcc --offload-arch=gfx950 -O2 issue_cost.hip -o issue_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP 256
template <int OP> __global__ void chain(float *out, uint64_t *clk, float a0, double d0)
{
    float a = a0 + threadIdx.x;
    double d = d0 + threadIdx.x;
    float b = a0 * 3;
    uint64_t t0 = __builtin_readcyclecounter();
#pragma unroll
    for (int i = 0; i < REP; ++i) {
        if (OP == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a) : "v"(b));
        if (OP == 1) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d) : "v"(d0));
        if (OP == 2) { asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d) : "v"(a)); asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a) : "v"(d)); }
        if (OP == 3) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a) : "v"(b));
        if (OP == 4) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1" : "+v"(a));
        if (OP == 5) { asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d) : "v"(a)); asm volatile("v_add_f64 %0, %0, %1" : "+v"(d) : "v"(d0)); asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a) : "v"(d)); }
        if (OP == 6) asm volatile("v_cmp_gt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a) : "v"(b) : "vcc");
        if (OP == 7) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d) : "v"(d0));
        if (OP == 8) asm volatile("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1" : "+v"(a));
        if (OP == 9) { float c; asm volatile("v_add_f32 %0, %1, %2" : "=v"(c) : "v"(a), "v"(b)); asm volatile("v_add_f32 %0, %1, %2" : "=v"(b) : "v"(b), "v"(a0)); a = c; }   // two independent chains
    }
    uint64_t t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = a + (float)d + b;
    if (threadIdx.x == 0) clk[0] = t1 - t0;
}
// a team of 4 wavefronts exchanging one word through LDS per step: write, s_waitcnt lgkmcnt(0) + s_barrier, read
__global__ void __launch_bounds__(256) exchange(float *out, uint64_t *clk, int steps, int extra)
{
    __shared__ int pub[64];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int v = threadIdx.x;
    uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < steps; ++i) {
        if (lane == 0) pub[wv] = v;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        int a = 0;
        for (int w = 0; w < 4; ++w) a += __builtin_amdgcn_readfirstlane(pub[w]);
        v = a + i;
        for (int e = 0; e < extra; ++e) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v) : "v"(a));
    }
    uint64_t t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = (float)v;
    if (threadIdx.x == 0) clk[0] = t1 - t0;
}
// where the wavefronts of a workgroup land (HW_ID: simd bits 5:4, cu 11:8, se 15:13) and what a dependent chain costs each
__global__ void __launch_bounds__(1024) placement(unsigned *hw, uint64_t *clk)
{
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    float a = threadIdx.x;
    uint64_t t0 = __builtin_readcyclecounter();
#pragma unroll
    for (int i = 0; i < 256; ++i) asm volatile("v_add_f32 %0, %0, %0" : "+v"(a));
    uint64_t t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) { const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); hw[w] = id; clk[w] = t1 - t0; }
    if (a == 12345.f) hw[0] = 0;
}
template <int OP> static void run(const char *name, int per_iter, float *o, uint64_t *c)
{
    uint64_t h = 0;
    for (int k = 0; k < 2; ++k) { hipLaunchKernelGGL(chain<OP>, dim3(1), dim3(64), 0, 0, o, c, 1.5f, 0.25); hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost); }
    printf("%-44s %6.2f ticks of s_memtime per instruction (%d per iteration)\n", name, (double)h / REP / per_iter, per_iter);
}
int main()
{
    float *o; uint64_t *c;
    hipMalloc(&o, 4096); hipMalloc(&c, 64);
    run<0>("v_add_f32 dependent", 1, o, c);
    run<9>("v_add_f32 two independent chains", 2, o, c);
    run<1>("v_add_f64 dependent", 1, o, c);
    run<7>("v_fma_f64 dependent", 1, o, c);
    run<2>("v_cvt_f64_f32 + v_cvt_f32_f64 dependent", 2, o, c);
    run<5>("cvt, add_f64, cvt dependent", 3, o, c);
    run<3>("v_max_f32 dependent", 1, o, c);
    run<6>("v_cmp + v_cndmask dependent", 2, o, c);
    run<4>("v_mov_dpp row_shr + s_nop 1", 2, o, c);
    run<8>("v_mov_dpp wave_shr + s_nop 1", 2, o, c);
    for (int extra : {0, 32}) {
        uint64_t h = 0;
        for (int k = 0; k < 2; ++k) { hipLaunchKernelGGL(exchange, dim3(1), dim3(256), 0, 0, o, c, 1000, extra); hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost); }
        printf("team of 4 wavefronts: LDS write + barrier + 4 broadcast reads + a %d-trip loop of one add (4 instructions a trip): %.0f ticks per step\n", extra, (double)h / 1000);
    }
    {
        unsigned *hw; uint64_t *ck; hipMalloc(&hw, 4096); hipMalloc(&ck, 8192);
        for (int waves : {1, 2, 4, 8, 16}) {
            unsigned h[16]; uint64_t c2[16];
            for (int k = 0; k < 2; ++k) hipLaunchKernelGGL(placement, dim3(1), dim3(64 * waves), 0, 0, hw, ck);
            hipMemcpy(h, hw, 4 * waves, hipMemcpyDeviceToHost); hipMemcpy(c2, ck, 8 * waves, hipMemcpyDeviceToHost);
            printf("workgroup of %2d wavefronts:", waves);
            for (int w = 0; w < waves; ++w) printf(" [simd %u cu %u: %.1f]", h[w] >> 4 & 3, h[w] >> 8 & 15, (double)c2[w] / 256);
            printf(" ticks per dependent v_add_f32\n");
        }
    }
    int khz = 0; hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
    int wkhz = 0; hipDeviceGetAttribute(&wkhz, hipDeviceAttributeWallClockRate, 0);
    printf("shader clock %d kHz, wall clock %d kHz (readcyclecounter = s_memtime ticks)\n", khz, wkhz);
    return 0;
}
