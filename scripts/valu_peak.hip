// valu_peak — measured VALU issue rates of gfx950, per instruction FORM, for the forms the gbx kernels live on.
//
//   hipcc -O2 --offload-arch=gfx950 scripts/valu_peak.hip -o scripts/valu_peak && ./scripts/valu_peak > profiles/valu_peak.json
//
// Every form runs as long unrolled streams of independent instructions (8 accumulators per wavefront, 64 instructions
// per loop trip) on all CUs at 1, 2, 4 and 8 wavefronts per SIMD.  Reported per form and occupancy:
//   lane_ops_per_s  = wave instructions x 64 lanes / wall time (HIP events)           -> the roof bench.py divides by
//   cyc_per_inst_simd = SIMD cycles between two issues of the form (s_memtime ticks of a wavefront / its instructions
//                       / wavefronts sharing the SIMD)                                 -> 2 = full rate on a SIMD-32
// Nothing here is linked into the product; it is a measurement tool.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

// one instruction on each of the eight accumulators; B, C are loop-invariant operands
#define REP8(INS)                                                                                                      \
    asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)                                               \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)                      \
                 : "v"(b), "v"(c)                                                                                      \
                 : "vcc", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47");
#define REP64(INS) REP8(INS) REP8(INS) REP8(INS) REP8(INS) REP8(INS) REP8(INS) REP8(INS) REP8(INS)

#define KERNEL32(NAME, INS)                                                                                            \
    __global__ void __launch_bounds__(256) NAME(int iters, unsigned *out, unsigned long long *ticks)                   \
    {                                                                                                                  \
        unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6,       \
                 a7 = a0 + 7, b = blockIdx.x | 0x01020304u, c = 0x3f800001u + threadIdx.x;                             \
        asm volatile("s_mov_b32 vcc_lo, 0x55555555\ns_mov_b32 vcc_hi, 0x55555555\ns_mov_b32 s40, 0x33333333\ns_mov_b32 s41, 0x33333333" ::: "vcc", "s40", "s41");              \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                          \
        for (int i = 0; i < iters; ++i) { REP64(INS) }                                                                 \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                          \
        out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;                                   \
        if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;                            \
    }

// 64-bit accumulators (register pairs)
#define REP8D(INS)                                                                                                     \
    asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)                                               \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)                      \
                 : "v"(b), "v"(c)                                                                                      \
                 : "vcc");
#define REP64D(INS) REP8D(INS) REP8D(INS) REP8D(INS) REP8D(INS) REP8D(INS) REP8D(INS) REP8D(INS) REP8D(INS)
#define KERNEL64(NAME, INS)                                                                                            \
    __global__ void __launch_bounds__(256) NAME(int iters, unsigned *out, unsigned long long *ticks)                   \
    {                                                                                                                  \
        double a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6,         \
               a7 = a0 + 7, b = 1.0 + blockIdx.x * 1e-9, c = 1.0000001;                                                \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                          \
        for (int i = 0; i < iters; ++i) { REP64D(INS) }                                                                \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                          \
        out[blockIdx.x * 256 + threadIdx.x] = (unsigned)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);                       \
        if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;                            \
    }

#define I_ADD(k)      "v_add_u32 %" #k ", %" #k ", %8\n"
#define I_SUB(k)      "v_sub_u32 %" #k ", %" #k ", %8\n"
#define I_MAX(k)      "v_max_i32 %" #k ", %" #k ", %8\n"
#define I_MINU(k)     "v_min_u32 %" #k ", %" #k ", %8\n"
#define I_MAX3(k)     "v_max3_i32 %" #k ", %" #k ", %8, %9\n"
#define I_ADD3(k)     "v_add3_u32 %" #k ", %" #k ", %8, %9\n"
#define I_LSHLADD(k)  "v_lshl_add_u32 %" #k ", %" #k ", 1, %8\n"
#define I_ANDOR(k)    "v_and_or_b32 %" #k ", %" #k ", %8, %9\n"
#define I_AND(k)      "v_and_b32 %" #k ", %" #k ", %8\n"
#define I_LSHL(k)     "v_lshlrev_b32 %" #k ", 1, %" #k "\n"
#define I_ASHR(k)     "v_ashrrev_i32 %" #k ", 1, %" #k "\n"
#define I_BFE(k)      "v_bfe_i32 %" #k ", %" #k ", 3, 9\n"
#define I_BFI(k)      "v_bfi_b32 %" #k ", %8, %" #k ", %9\n"
#define I_PERM(k)     "v_perm_b32 %" #k ", %" #k ", %8, %9\n"
#define I_MOV(k)      "v_mov_b32 %" #k ", %8\n"
#define I_CNDMASK(k)  "v_cndmask_b32 %" #k ", %" #k ", %8, vcc\n"
#define I_CNDMASK64(k) "v_cndmask_b32_e64 %" #k ", %" #k ", %8, s[40:41]\n"
#define I_CNDMASK_C(k) "v_cndmask_b32 %" #k ", 0, %" #k ", vcc\n"
#define I_OR(k)       "v_or_b32 %" #k ", %" #k ", %8\n"
#define I_XOR(k)      "v_xor_b32 %" #k ", %" #k ", %8\n"
#define I_LSHR(k)     "v_lshrrev_b32 %" #k ", 1, %" #k "\n"
#define I_MINI(k)     "v_min_i32 %" #k ", %" #k ", %8\n"
#define I_MAXU(k)     "v_max_u32 %" #k ", %" #k ", %8\n"
#define I_MED3(k)     "v_med3_i32 %" #k ", %" #k ", %8, %9\n"
#define I_MIN3(k)     "v_min3_i32 %" #k ", %" #k ", %8, %9\n"
#define I_ADDCO(k)    "v_add_co_u32 %" #k ", vcc, %" #k ", %8\n"
#define I_SUBREV(k)   "v_subrev_u32 %" #k ", %" #k ", %8\n"
#define I_MUL24(k)    "v_mul_u32_u24 %" #k ", %" #k ", %8\n"
#define I_ALIGNBIT(k) "v_alignbit_b32 %" #k ", %" #k ", %8, 16\n"
#define I_MAXI16(k)   "v_max_i16 %" #k ", %" #k ", %8\n"
#define I_ADDU16(k)   "v_add_u16 %" #k ", %" #k ", %8\n"
#define I_PKMAD16(k)  "v_pk_mad_i16 %" #k ", %" #k ", %8, %9\n"
#define I_PKLSHL16(k) "v_pk_lshlrev_b16 %" #k ", 1, %" #k "\n"
#define I_PKASHR16(k) "v_pk_ashrrev_i16 %" #k ", 1, %" #k "\n"
#define I_PKMAXU16(k) "v_pk_max_u16 %" #k ", %" #k ", %8\n"
#define I_PKSUBI16(k) "v_pk_sub_i16 %" #k ", %" #k ", %8\n"
#define I_SAD(k)      "v_sad_u32 %" #k ", %" #k ", %8, %9\n"
#define I_ADD_DPP(k)  "v_add_u32_dpp %" #k ", %" #k ", %" #k " row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
#define I_AND_SDWA(k) "v_and_b32_sdwa %" #k ", %" #k ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
#define I_SUBF(k)     "v_sub_f32 %" #k ", %" #k ", %9\n"
#define I_MACF(k)     "v_fmac_f32 %" #k ", %8, %9\n"
#define I_PAIR_VCC(k)  "v_cmp_gt_i32 vcc, %" #k ", %8\nv_cndmask_b32 %" #k ", %" #k ", %9, vcc\n"
#define I_PAIR_SGPR(k) "v_cmp_gt_i32 s[40:41], %" #k ", %8\nv_cndmask_b32_e64 %" #k ", %" #k ", %9, s[40:41]\n"
#define I_PAIR_VCC_FAR(k)  "v_cmp_gt_i32 vcc, %" #k ", %8\nv_add_u32 %" #k ", %" #k ", %8\nv_add_u32 %" #k ", %" #k ", %8\nv_cndmask_b32 %" #k ", %" #k ", %9, vcc\n"
#define I_CMP(k)      "v_cmp_gt_i32 vcc, %" #k ", %8\n"
#define I_CMP_S(k)    "v_cmp_gt_i32 s[40:41], %" #k ", %8\n"
#define I_MULLO(k)    "v_mul_lo_u32 %" #k ", %" #k ", %8\n"
#define I_MAD24(k)    "v_mad_u32_u24 %" #k ", %" #k ", %8, %9\n"
#define I_MBCNT(k)    "v_mbcnt_lo_u32_b32 %" #k ", %8, %" #k "\n"
#define I_MAX_QP(k)   "v_max_i32_dpp %" #k ", %" #k ", %" #k " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
#define I_MAX_SHR(k)  "v_max_i32_dpp %" #k ", %" #k ", %" #k " row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
#define I_MAX_ROR(k)  "v_max_i32_dpp %" #k ", %" #k ", %" #k " row_ror:4 row_mask:0xf bank_mask:0xf\n"
#define I_MAX_HM(k)   "v_max_i32_dpp %" #k ", %" #k ", %" #k " row_half_mirror row_mask:0xf bank_mask:0xf\n"
#define I_MAX_WSHR(k) "v_max_i32_dpp %" #k ", %" #k ", %" #k " wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
#define I_MAX_BC15(k) "v_max_i32_dpp %" #k ", %" #k ", %" #k " row_bcast:15 row_mask:0xa bank_mask:0xf\n"
#define I_MOV_DPP(k)  "v_mov_b32_dpp %" #k ", %" #k " row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
#define I_ADD_SDWA(k) "v_add_u32_sdwa %" #k ", %" #k ", sext(%8) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
#define I_MAX_SDWA(k) "v_max_i32_sdwa %" #k ", %" #k ", sext(%8) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
#define I_PKMAX16(k)  "v_pk_max_i16 %" #k ", %" #k ", %8\n"
#define I_PKADD16(k)  "v_pk_add_i16 %" #k ", %" #k ", %8\n"
#define I_PKADD16C(k) "v_pk_add_i16 %" #k ", %" #k ", %8 clamp\n"
#define I_PKSUBU16(k) "v_pk_sub_u16 %" #k ", %" #k ", %8\n"
#define I_PKMINU16(k) "v_pk_min_u16 %" #k ", %" #k ", %8\n"
#define I_ADDF(k)     "v_add_f32 %" #k ", %" #k ", %9\n"
#define I_MULF(k)     "v_mul_f32 %" #k ", %" #k ", %9\n"
#define I_FMAF(k)     "v_fma_f32 %" #k ", %" #k ", %9, %9\n"
#define I_MAXF(k)     "v_max_f32 %" #k ", %" #k ", %9\n"
#define I_MULF_DPP(k) "v_mul_f32_dpp %" #k ", %" #k ", %" #k " wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
#define I_ADDF_DPP(k) "v_add_f32_dpp %" #k ", %" #k ", %" #k " row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
#define I_READLANE(k) "v_readlane_b32 s4" #k ", %" #k ", 17\n"
#define I_READFIRST(k) "v_readfirstlane_b32 s4" #k ", %" #k "\n"
#define I_CVT_F32_I32(k) "v_cvt_f32_i32 %" #k ", %" #k "\n"
#define I_EXP(k)      "v_exp_f32 %" #k ", %" #k "\n"
#define I_RCP(k)      "v_rcp_f32 %" #k ", %" #k "\n"

#define D_ADD(k)      "v_add_f64 %" #k ", %" #k ", %8\n"
#define D_MUL(k)      "v_mul_f64 %" #k ", %" #k ", %9\n"
#define D_FMA(k)      "v_fma_f64 %" #k ", %" #k ", %9, %8\n"
#define D_PKFMA(k)    "v_pk_fma_f32 %" #k ", %" #k ", %9, %8\n"
#define D_PKADD(k)    "v_pk_add_f32 %" #k ", %" #k ", %8\n"
#define D_PKMUL(k)    "v_pk_mul_f32 %" #k ", %" #k ", %9\n"
#define D_LSHL64(k)   "v_lshlrev_b64 %" #k ", 1, %" #k "\n"

KERNEL32(k_add_u32, I_ADD) KERNEL32(k_sub_u32, I_SUB) KERNEL32(k_max_i32, I_MAX) KERNEL32(k_min_u32, I_MINU)
KERNEL32(k_max3_i32, I_MAX3) KERNEL32(k_add3_u32, I_ADD3) KERNEL32(k_lshl_add_u32, I_LSHLADD) KERNEL32(k_and_or_b32, I_ANDOR)
KERNEL32(k_and_b32, I_AND) KERNEL32(k_lshlrev_b32, I_LSHL) KERNEL32(k_ashrrev_i32, I_ASHR) KERNEL32(k_bfe_i32, I_BFE)
KERNEL32(k_bfi_b32, I_BFI) KERNEL32(k_perm_b32, I_PERM) KERNEL32(k_mov_b32, I_MOV) KERNEL32(k_cndmask_b32, I_CNDMASK)
KERNEL32(k_cndmask_b32_e64_sgpr, I_CNDMASK64) KERNEL32(k_cndmask_b32_const, I_CNDMASK_C) KERNEL32(k_or_b32, I_OR) KERNEL32(k_xor_b32, I_XOR)
KERNEL32(k_lshrrev_b32, I_LSHR) KERNEL32(k_min_i32, I_MINI) KERNEL32(k_max_u32, I_MAXU) KERNEL32(k_med3_i32, I_MED3) KERNEL32(k_min3_i32, I_MIN3)
KERNEL32(k_add_co_u32, I_ADDCO) KERNEL32(k_subrev_u32, I_SUBREV) KERNEL32(k_mul_u32_u24, I_MUL24) KERNEL32(k_alignbit_b32, I_ALIGNBIT)
KERNEL32(k_max_i16, I_MAXI16) KERNEL32(k_add_u16, I_ADDU16) KERNEL32(k_pk_mad_i16, I_PKMAD16) KERNEL32(k_pk_lshlrev_b16, I_PKLSHL16)
KERNEL32(k_pk_ashrrev_i16, I_PKASHR16) KERNEL32(k_pk_max_u16, I_PKMAXU16) KERNEL32(k_pk_sub_i16, I_PKSUBI16) KERNEL32(k_sad_u32, I_SAD)
KERNEL32(k_add_u32_dpp_row_shr, I_ADD_DPP) KERNEL32(k_and_b32_sdwa_byte, I_AND_SDWA) KERNEL32(k_sub_f32, I_SUBF) KERNEL32(k_fmac_f32, I_MACF)
KERNEL32(k_pair_cmp_cndmask_vcc, I_PAIR_VCC) KERNEL32(k_pair_cmp_cndmask_sgpr, I_PAIR_SGPR) KERNEL32(k_quad_cmp_add_add_cndmask_vcc, I_PAIR_VCC_FAR)
KERNEL32(k_cmp_gt_i32_vcc, I_CMP) KERNEL32(k_cmp_gt_i32_sgpr, I_CMP_S) KERNEL32(k_mul_lo_u32, I_MULLO) KERNEL32(k_mad_u32_u24, I_MAD24)
KERNEL32(k_mbcnt_lo, I_MBCNT)
KERNEL32(k_max_i32_dpp_quad_perm, I_MAX_QP) KERNEL32(k_max_i32_dpp_row_shr, I_MAX_SHR) KERNEL32(k_max_i32_dpp_row_ror, I_MAX_ROR)
KERNEL32(k_max_i32_dpp_row_half_mirror, I_MAX_HM) KERNEL32(k_max_i32_dpp_wave_shr, I_MAX_WSHR) KERNEL32(k_max_i32_dpp_row_bcast15, I_MAX_BC15)
KERNEL32(k_mov_b32_dpp_row_shr, I_MOV_DPP)
KERNEL32(k_add_u32_sdwa_sext_byte, I_ADD_SDWA) KERNEL32(k_max_i32_sdwa_sext_word, I_MAX_SDWA)
KERNEL32(k_pk_max_i16, I_PKMAX16) KERNEL32(k_pk_add_i16, I_PKADD16) KERNEL32(k_pk_add_i16_clamp, I_PKADD16C)
KERNEL32(k_pk_sub_u16, I_PKSUBU16) KERNEL32(k_pk_min_u16, I_PKMINU16)
KERNEL32(k_add_f32, I_ADDF) KERNEL32(k_mul_f32, I_MULF) KERNEL32(k_fma_f32, I_FMAF) KERNEL32(k_max_f32, I_MAXF)
KERNEL32(k_mul_f32_dpp_wave_shr, I_MULF_DPP) KERNEL32(k_add_f32_dpp_row_shr, I_ADDF_DPP)
KERNEL32(k_readlane_b32, I_READLANE) KERNEL32(k_readfirstlane_b32, I_READFIRST)
KERNEL32(k_cvt_f32_i32, I_CVT_F32_I32) KERNEL32(k_exp_f32, I_EXP) KERNEL32(k_rcp_f32, I_RCP)
KERNEL64(k_add_f64, D_ADD) KERNEL64(k_mul_f64, D_MUL) KERNEL64(k_fma_f64, D_FMA)
KERNEL64(k_pk_fma_f32, D_PKFMA) KERNEL64(k_pk_add_f32, D_PKADD) KERNEL64(k_pk_mul_f32, D_PKMUL)
KERNEL64(k_lshlrev_b64, D_LSHL64)

// conversions between the two widths: eight float and eight double registers, every instruction independent
#define KERNEL_CVT(NAME, INS, OUTC, INC)                                                                               \
    __global__ void __launch_bounds__(256) NAME(int iters, unsigned *out, unsigned long long *ticks)                   \
    {                                                                                                                  \
        float f0 = threadIdx.x, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3, f4 = f0 + 4, f5 = f0 + 5, f6 = f0 + 6, f7 = f0 + 7; \
        double d0 = f0 * 1.5, d1 = d0 + 1, d2 = d0 + 2, d3 = d0 + 3, d4 = d0 + 4, d5 = d0 + 5, d6 = d0 + 6, d7 = d0 + 7;   \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                          \
        for (int i = 0; i < iters; ++i) {                                                                              \
            _Pragma("unroll") for (int u = 0; u < 8; ++u)                                                              \
                asm volatile(INS(0, 8) INS(1, 9) INS(2, 10) INS(3, 11) INS(4, 12) INS(5, 13) INS(6, 14) INS(7, 15)      \
                             : OUTC(0), OUTC(1), OUTC(2), OUTC(3), OUTC(4), OUTC(5), OUTC(6), OUTC(7)                  \
                             : INC(0), INC(1), INC(2), INC(3), INC(4), INC(5), INC(6), INC(7));                        \
        }                                                                                                              \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                          \
        out[blockIdx.x * 256 + threadIdx.x] = (unsigned)(f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7 + d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7); \
        if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;                            \
    }
#define C_DF(a, b) "v_cvt_f32_f64 %" #a ", %" #b "\n"
#define C_FD(a, b) "v_cvt_f64_f32 %" #a ", %" #b "\n"
#define OUT_F(k) "+v"(f##k)
#define OUT_D(k) "+v"(d##k)
#define IN_F(k) "v"(f##k)
#define IN_D(k) "v"(d##k)
KERNEL_CVT(k_cvt_f32_f64, C_DF, OUT_F, IN_D)
KERNEL_CVT(k_cvt_f64_f32, C_FD, OUT_D, IN_F)

typedef void (*kern_t)(int, unsigned *, unsigned long long *);
struct Form { const char *name; kern_t k; int lanes_per_inst_mult; };   // packed forms: 2 element operations per lane

#define F(n) {#n, k_##n, 1}
#define F2(n) {#n, k_##n, 2}
static const Form forms[] = {
    F(add_u32), F(sub_u32), F(max_i32), F(min_u32), F(max3_i32), F(add3_u32), F(lshl_add_u32), F(and_or_b32), F(and_b32),
    F(lshlrev_b32), F(ashrrev_i32), F(bfe_i32), F(bfi_b32), F(perm_b32), F(mov_b32), F(cndmask_b32), F(cmp_gt_i32_vcc),
    F(cndmask_b32_e64_sgpr), F(cndmask_b32_const), F(or_b32), F(xor_b32), F(lshrrev_b32), F(min_i32), F(max_u32), F(med3_i32), F(min3_i32),
    F(add_co_u32), F(subrev_u32), F(mul_u32_u24), F(alignbit_b32), F(max_i16), F(add_u16), F2(pk_mad_i16), F2(pk_lshlrev_b16), F2(pk_ashrrev_i16),
    F2(pk_max_u16), F2(pk_sub_i16), F(sad_u32), F(add_u32_dpp_row_shr), F(and_b32_sdwa_byte), F(sub_f32), F(fmac_f32),
    F(pair_cmp_cndmask_vcc), F(pair_cmp_cndmask_sgpr), F(quad_cmp_add_add_cndmask_vcc),
    F(cmp_gt_i32_sgpr), F(mul_lo_u32), F(mad_u32_u24), F(mbcnt_lo),
    F(max_i32_dpp_quad_perm), F(max_i32_dpp_row_shr), F(max_i32_dpp_row_ror), F(max_i32_dpp_row_half_mirror),
    F(max_i32_dpp_wave_shr), F(max_i32_dpp_row_bcast15), F(mov_b32_dpp_row_shr),
    F(add_u32_sdwa_sext_byte), F(max_i32_sdwa_sext_word),
    F2(pk_max_i16), F2(pk_add_i16), F2(pk_add_i16_clamp), F2(pk_sub_u16), F2(pk_min_u16),
    F(add_f32), F(mul_f32), F(fma_f32), F(max_f32), F(mul_f32_dpp_wave_shr), F(add_f32_dpp_row_shr),
    F(readlane_b32), F(readfirstlane_b32), F(cvt_f32_i32), F(exp_f32), F(rcp_f32),
    F(add_f64), F(mul_f64), F(fma_f64), F2(pk_fma_f32), F2(pk_add_f32), F2(pk_mul_f32), F(cvt_f32_f64), F(cvt_f64_f32), F(lshlrev_b64),
};

int main(int argc, char **argv)
{
    int iters = argc > 1 ? atoi(argv[1]) : 4000;
    const char *only = argc > 2 ? argv[2] : nullptr;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const int max_blocks = cus * 8;
    unsigned *out;
    unsigned long long *ticks;
    CHECK(hipMalloc(&out, (size_t)max_blocks * 256 * 4));
    CHECK(hipMalloc(&ticks, (size_t)max_blocks * 4 * 8));
    std::vector<unsigned long long> h((size_t)max_blocks * 4);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    printf("{\"device\": \"%s\", \"gcn_arch\": \"%s\", \"cus\": %d, \"clock_khz\": %d, \"iters\": %d, \"insts_per_wave\": %lld,\n"
           " \"command\": \"./scripts/valu_peak %d\",\n \"note\": \"blocks of 4 wavefronts (one per SIMD); W wavefronts per SIMD = W blocks per CU; "
           "lane_ops_per_s = wave instructions x 64 / wall time; cyc_per_inst_simd = s_memtime ticks per instruction of one wavefront / W "
           "forms named pair_* / quad_* are 2 / 4 instructions per counted instruction: divide their cycles accordingly\",\n \"forms\": {\n",
           prop.name, prop.gcnArchName, cus, prop.clockRate, iters, (long long)iters * 64, iters);
    bool first = true;
    for (const Form &f : forms) {
        if (only && !strstr(f.name, only)) continue;
        printf("%s  \"%s\": {", first ? "" : ",\n", f.name);
        first = false;
        const int occ[4] = {1, 2, 4, 8};
        for (int o = 0; o < 4; ++o) {
            const int blocks = cus * occ[o];
            hipLaunchKernelGGL(f.k, dim3(blocks), dim3(256), 0, 0, 64, out, ticks);       // warm-up
            CHECK(hipDeviceSynchronize());
            float best = 1e30f;
            double tick_mean = 0;
            for (int rep = 0; rep < 3; ++rep) {
                CHECK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL(f.k, dim3(blocks), dim3(256), 0, 0, iters, out, ticks);
                CHECK(hipEventRecord(e1, 0));
                CHECK(hipEventSynchronize(e1));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) {
                    best = ms;
                    CHECK(hipMemcpy(h.data(), ticks, (size_t)blocks * 4 * 8, hipMemcpyDeviceToHost));
                    double s = 0;
                    for (int i = 0; i < blocks * 4; ++i) s += (double)h[i];
                    tick_mean = s / (blocks * 4);
                }
            }
            const double insts = (double)blocks * 4 * iters * 64;
            const double lane_ops = insts * 64 * f.lanes_per_inst_mult / (best * 1e-3);
            // wall-clock cycles per instruction per SIMD at the nominal clock, and the same from the wavefront's own ticks
            const double cyc_wall = (best * 1e-3) * prop.clockRate * 1e3 / ((double)iters * 64 * occ[o]);
            const double ticks_per_inst = tick_mean / ((double)iters * 64) / occ[o];
            printf("%s\"w%d\": {\"ms\": %.4f, \"lane_ops_per_s\": %.4e, \"cyc_per_inst_simd_wall\": %.3f, \"ticks_per_inst_simd\": %.4f}",
                   o ? ", " : "", occ[o], best, lane_ops, cyc_wall, ticks_per_inst);
        }
        printf("}");
        fflush(stdout);
    }
    printf("\n }\n}\n");
    return 0;
}
