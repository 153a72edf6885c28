// bsw_step_bench.hip - development aid: the column step of the compact lane kernel as it is - one pair per lane,
// a column PAIR per 28 hand-scheduled instructions (lane_pair_step, included from csrc/bsw_kernels.hip) - against the variant with
// TWO pairs per lane in the halves of packed int16 arithmetic (v_pk_add / sub / max _i16, cell dword = one column of two pairs,
// un- and re-packed by v_perm_b32, `diag ? diag + s : 0` as an AND with 0 - min_u16(diag, 1)).  Both sweep R rows over C columns of
// cells that live in LDS as in the kernel (3 bytes per column and pair), four column steps per loop trip with the loads ahead, at the
// LDS occupancy the long-query classes have (27 KB per 64 pairs).  The packed step is given its BEST case: both pairs of a lane share
// one window (no per-half masks, which the real kernel would need: windows are per pair) and one target base per row.
// Prints cells per second (best of eight interleaved launches); the static instruction counts quoted in profiles/r05o_* come from the ISA (hipcc -S).
//   hipcc -O3 --offload-arch=gfx950 -I genomicsbench_amd/csrc scripts/bsw_step_bench.hip -Lgenomicsbench_amd -lgbx -Wl,-rpath,$PWD/genomicsbench_amd -o build_tmp/bsw_step_bench
#include "bsw_kernels.hip"
#include <cstdio>
#include <vector>
#include <algorithm>
using namespace gbx;

typedef short v2s __attribute__((ext_vector_type(2)));
__device__ inline v2s pk(unsigned u) { return __builtin_bit_cast(v2s, u); }
__device__ inline unsigned bits(v2s v) { return __builtin_bit_cast(unsigned, v); }
__device__ inline v2s pkmax(v2s a, v2s b) { return __builtin_elementwise_max(a, b); }
typedef unsigned short v2u16 __attribute__((ext_vector_type(2)));
__device__ inline v2s pkminu(v2s a, v2s b) { return __builtin_bit_cast(v2s, __builtin_elementwise_min(__builtin_bit_cast(v2u16, a), __builtin_bit_cast(v2u16, b))); }

// one pair per lane: the kernel's own main loop body (four column pairs per trip)
__global__ void __launch_bounds__(64) step_one(int rows, int cols, unsigned *sink, unsigned rw, unsigned rwn)
{
    extern __shared__ uint32_t lcell[];
    const int lane = threadIdx.x, cb = lane * 4;
    const int qb = (cols >> 1) * 256 + (lane & 31) * 4 + (lane >> 5) * 2;
    for (int k = lane; k < (cols >> 1) * 96 + 512; k += 64) lcell[k] = 0x01020304u * (unsigned)((k & 3) + 1) & 0x07070707u;
    __syncthreads();
#define LCELL(a) (*(uint32_t *)((char *)lcell + (a)))
#define LCELL16(a) (*(uint16_t *)((char *)lcell + (a)))
    int rel0 = 0, rel1 = 2, rel2 = 256, rel3 = 258, rel4 = 512, rel5 = 514, rel6 = 768, rel7 = 770, vzero = 0;
    asm volatile("" : "+v"(rel0), "+v"(rel1), "+v"(rel2), "+v"(rel3), "+v"(rel4), "+v"(rel5), "+v"(rel6), "+v"(rel7), "+v"(vzero));
    uint32_t key = 0;
    for (int i = 0; i < rows; ++i) {
        int f = 0, left = i & 7;
        int pa = cb, qa = qb;
        uint32_t w0 = LCELL(pa), q0 = LCELL16(qa), w1 = LCELL(pa + 256), q1 = LCELL16(qa + 128);
        for (int j = 0; j + 7 < cols; j += 8, pa += 1024, qa += 512) {
            const uint32_t x0 = LCELL(pa + 512), y0 = LCELL16(qa + 256), x1 = LCELL(pa + 768), y1 = LCELL16(qa + 384);
            uint32_t kt;
            LCELL(pa) = lane_pair_step<true>(w0, q0, rw, rwn, f, left, (uint32_t)vzero, kt, rel0, rel1, vzero, 7, 7, 1, 1);
            LCELL(pa + 256) = lane_pair_step<true>(w1, q1, rw, rwn, f, left, kt, kt, rel2, rel3, vzero, 7, 7, 1, 1);
            w0 = LCELL(pa + 1024); q0 = LCELL16(qa + 512); w1 = LCELL(pa + 1280); q1 = LCELL16(qa + 640);
            LCELL(pa + 512) = lane_pair_step<true>(x0, y0, rw, rwn, f, left, kt, kt, rel4, rel5, vzero, 7, 7, 1, 1);
            LCELL(pa + 768) = lane_pair_step<true>(x1, y1, rw, rwn, f, left, kt, kt, rel6, rel7, vzero, 7, 7, 1, 1);
            key = max(key, kt + (uint32_t)pa);
        }
    }
    if (key == 0xdeadbeefu) sink[lane] = key;
}

// one pair per lane with FOUR-BIT query codes: a byte of codes per column pair instead of a halfword - 2.5 instead
// of 3 bytes of LDS per column and lane, i.e. 7 instead of 6 wavefronts of the 100..135 class per CU - expanded into the v_perm_b32
// selector by three more instructions per column pair (shift, and, shift-or)
__global__ void __launch_bounds__(64) step_one4(int rows, int cols, unsigned *sink, unsigned rw, unsigned rwn)
{
    extern __shared__ uint32_t lcell[];
    const int lane = threadIdx.x, cb = lane * 4;
    const int qb = (cols >> 1) * 256 + lane;                    // one byte per column pair and lane
    for (int k = lane; k < (cols >> 1) * 80 + 512; k += 64) lcell[k] = 0x01020304u * (unsigned)((k & 3) + 1) & 0x07070707u & 0x43434343u;
    __syncthreads();
#define LQ8(a) (*((uint8_t *)lcell + (a)))
#define SEL4(q) ((((q) >> 4) << 8) | ((q) & 15u))
    int rel0 = 0, rel1 = 2, rel2 = 256, rel3 = 258, rel4 = 512, rel5 = 514, rel6 = 768, rel7 = 770, vzero = 0;
    asm volatile("" : "+v"(rel0), "+v"(rel1), "+v"(rel2), "+v"(rel3), "+v"(rel4), "+v"(rel5), "+v"(rel6), "+v"(rel7), "+v"(vzero));
    uint32_t key = 0;
    for (int i = 0; i < rows; ++i) {
        int f = 0, left = i & 7;
        int pa = cb, qa = qb;
        uint32_t w0 = LCELL(pa), q0 = LQ8(qa), w1 = LCELL(pa + 256), q1 = LQ8(qa + 64);
        for (int j = 0; j + 7 < cols; j += 8, pa += 1024, qa += 256) {
            const uint32_t x0 = LCELL(pa + 512), y0 = LQ8(qa + 128), x1 = LCELL(pa + 768), y1 = LQ8(qa + 192);
            uint32_t kt;
            LCELL(pa) = lane_pair_step<true>(w0, SEL4(q0), rw, rwn, f, left, (uint32_t)vzero, kt, rel0, rel1, vzero, 7, 7, 1, 1);
            LCELL(pa + 256) = lane_pair_step<true>(w1, SEL4(q1), rw, rwn, f, left, kt, kt, rel2, rel3, vzero, 7, 7, 1, 1);
            w0 = LCELL(pa + 1024); q0 = LQ8(qa + 256); w1 = LCELL(pa + 1280); q1 = LQ8(qa + 320);
            LCELL(pa + 512) = lane_pair_step<true>(x0, SEL4(y0), rw, rwn, f, left, kt, kt, rel4, rel5, vzero, 7, 7, 1, 1);
            LCELL(pa + 768) = lane_pair_step<true>(x1, SEL4(y1), rw, rwn, f, left, kt, kt, rel6, rel7, vzero, 7, 7, 1, 1);
            key = max(key, kt + (uint32_t)pa);
        }
    }
    if (key == 0xdeadbeefu) sink[lane] = key;
}

// two pairs per lane: cell dword of column j = bytes {eA, hA, eB, hB}; codes halfword = {cA, cB}
__device__ __forceinline__ unsigned pk_step(unsigned w, unsigned q, unsigned rwA, unsigned rwnA, unsigned rwB, unsigned rwnB, v2s &f, v2s &left, v2s &key, unsigned col2,
                                            v2s oe_del, v2s e_del, v2s e_ins, v2s bias)
{
    const unsigned sA = __builtin_amdgcn_perm(rwnA, rwA, q & 0xffu), sB = __builtin_amdgcn_perm(rwnB, rwB, q >> 8);      // biased scores (0..) in byte 0
    const v2s s = pk(__builtin_amdgcn_perm(sB, sA, 0x0c040c00u));                       // {sA, sB} zero-extended
    const v2s h = pk(__builtin_amdgcn_perm(0u, w, 0x0c030c01u));                         // {hA, hB}
    const v2s e = pk(__builtin_amdgcn_perm(0u, w, 0x0c020c00u));                         // {eA, eB}
    v2s m = h + s - bias;
    const v2s one = {1, 1}, zero = {0, 0};
    m = pk(bits(m) & bits(zero - pkminu(h, one)));                                       // diag ? diag + s : 0
    const v2s hh = pkmax(pkmax(m, e), f);
    const v2s td = m - oe_del;
    const v2s en = pkmax(pkmax(e - e_del, td), zero);
    f = pkmax(pkmax(f - e_ins, td), zero);
    key = __builtin_bit_cast(v2s, __builtin_elementwise_max(__builtin_bit_cast(v2u16, key), __builtin_bit_cast(v2u16, pk((bits(hh) << 8) | col2))));
    const unsigned cell = __builtin_amdgcn_perm(bits(left), bits(en), 0x06020400u);      // {enA, leftA, enB, leftB}
    left = hh;
    return cell;
}
__global__ void __launch_bounds__(64) step_two(int rows, int cols, unsigned *sink, unsigned rw, unsigned rwn)
{
    extern __shared__ uint32_t lcell[];
    const int lane = threadIdx.x, cb = lane * 4;
    const int qb = cols * 256 + lane * 2;
    for (int k = lane; k < cols * 96 + 512; k += 64) lcell[k] = 0x01020304u * (unsigned)((k & 3) + 1) & 0x07070707u;
    __syncthreads();
    const v2s oe_del = {7, 7}, e_del = {1, 1}, e_ins = {1, 1}, bias = {4, 4};
    v2s key = {0, 0};
    for (int i = 0; i < rows; ++i) {
        v2s f = {0, 0}, left = {(short)(i & 7), (short)(i & 3)};
        int pa = cb, qa = qb;
        uint32_t w0 = LCELL(pa), q0 = LCELL16(qa), w1 = LCELL(pa + 256), q1 = LCELL16(qa + 128);
        for (int j = 0; j + 3 < cols; j += 4, pa += 1024, qa += 512) {
            const uint32_t x0 = LCELL(pa + 512), y0 = LCELL16(qa + 256), x1 = LCELL(pa + 768), y1 = LCELL16(qa + 384);
            const unsigned c2 = (unsigned)j * 0x00010001u;
            LCELL(pa) = pk_step(w0, q0, rw, rwn, rw ^ 0x01010101u, rwn, f, left, key, c2, oe_del, e_del, e_ins, bias);
            LCELL(pa + 256) = pk_step(w1, q1, rw, rwn, rw ^ 0x01010101u, rwn, f, left, key, c2 + 0x00010001u, oe_del, e_del, e_ins, bias);
            w0 = LCELL(pa + 1024); q0 = LCELL16(qa + 512); w1 = LCELL(pa + 1280); q1 = LCELL16(qa + 640);
            LCELL(pa + 512) = pk_step(x0, y0, rw, rwn, rw ^ 0x01010101u, rwn, f, left, key, c2 + 0x00020002u, oe_del, e_del, e_ins, bias);
            LCELL(pa + 768) = pk_step(x1, y1, rw, rwn, rw ^ 0x01010101u, rwn, f, left, key, c2 + 0x00030003u, oe_del, e_del, e_ins, bias);
        }
    }
    if (bits(key) == 0xdeadbeefu) sink[lane] = bits(key);
}

int main()
{
    unsigned *sink; (void)hipMalloc(&sink, 4096);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const int rows = 4000, cols = 136;
    // LDS per wavefront as the kernel's long-query class has it: one pair per lane 3 B x 64 x 138 = 27 KB (6 per CU);
    // two pairs per lane 6 B x 64 x 138 = 53 KB (3 per CU): the same pairs per CU
    const size_t lds1 = (size_t)(cols >> 1) * 384 + 2048, lds2 = (size_t)cols * 384 + 2048;
    auto timed = [&](auto launch) { float ms = 0; (void)hipEventRecord(a); launch(); (void)hipEventRecord(b); (void)hipEventSynchronize(b); (void)hipEventElapsedTime(&ms, a, b); return ms; };
    const size_t lds4 = (size_t)(cols >> 1) * 320 + 2048;
    float best1 = 1e9f, best2 = 1e9f, best4 = 1e9f, best4at6 = 1e9f;
    for (int rep = 0; rep < 8; ++rep) {                        // interleaved, best of eight (the chip's clock wanders between launches)
        best1 = std::min(best1, timed([&] { hipLaunchKernelGGL(step_one, dim3(256 * 6), dim3(64), lds1, 0, rows, cols, sink, 0x01fcfcfcu, 0xffffffffu); }));
        best4 = std::min(best4, timed([&] { hipLaunchKernelGGL(step_one4, dim3(256 * 7), dim3(64), lds4, 0, rows, cols, sink, 0x01fcfcfcu, 0xffffffffu); }));
        best4at6 = std::min(best4at6, timed([&] { hipLaunchKernelGGL(step_one4, dim3(256 * 6), dim3(64), lds1, 0, rows, cols, sink, 0x01fcfcfcu, 0xffffffffu); }));
        best2 = std::min(best2, timed([&] { hipLaunchKernelGGL(step_two, dim3(256 * 3), dim3(64), lds2, 0, rows, cols, sink, 0x05000000u, 0x03030303u); }));
    }
    const double c6 = 256.0 * 6 * 64 * rows * cols, c7 = 256.0 * 7 * 64 * rows * cols;
    printf("one pair per lane, 28 instructions per column pair, 6 wavefronts per CU:            %.2f ms  %.0f G cells/s\n", best1, c6 / best1 / 1e6);
    printf("  ... four-bit query codes (31 per column pair), 7 wavefronts per CU (2.5 B / column):  %.2f ms  %.0f G cells/s\n", best4, c7 / best4 / 1e6);
    printf("  ... four-bit query codes at the SAME 6 wavefronts per CU (the instructions' cost):    %.2f ms  %.0f G cells/s\n", best4at6, c6 / best4at6 / 1e6);
    printf("two pairs per lane, packed int16, 3 wavefronts per CU (the same pairs per CU):          %.2f ms  %.0f G cells/s\n", best2, c6 / best2 / 1e6);
    return 0;
}
