// poa_kernels.hip — partial-order-alignment consensus (spoa) for gfx950 (MI355X).
//
// Replaces, at whole-window granularity, the driver's loop over
// AlignmentEngine::align / Graph::add_alignment / Graph::generate_consensus
// (R/benchmarks/poa/msa_spoa_omp.cpp:237-252); algorithm = spoa v3 kNW with
// convex (two-piece affine) gaps, restated in poa_graph.h and below
// (spoa itself is an empty submodule in the reference).
//
// Design: one window per wavefront, thousands of windows in flight.
//   * The graph (SoA, bounded fan-in lists), the five DP matrices of the current
//     alignment and all scratch live in a per-wave workspace slot in HBM — this
//     is the one kernel of the four whose DP state cannot stay on chip
//     ((nodes+1) x (len+1) x {H,F,E,O,Q}).
//   * DP: graph rows in topological order, sequence columns across the lanes
//     (CPL consecutive columns per lane).  F/O/diagonal-H of a row come from
//     predecessor rows (coalesced row loads); the horizontal pieces E and Q are
//     a 2-state max-plus linear recurrence along the row,
//         (E,Q)[j+1] = T (x) (E,Q)[j] (+) (H[j]+g, H[j]+q),   T = [[e,g],[q,c]],
//     solved exactly with a lane-local pass, a Kogge-Stone DPP scan over the
//     lanes with constant 2x2 max-plus matrices T^(CPL*2^s), and a second local
//     pass (bit-identical E, Q, H to the sequential definition).
//   * Traceback, add_alignment, topological sort and the heaviest-bundle
//     consensus are inherently serial; every lane of the wave executes them
//     with identical (wave-uniform) data, so no election or broadcast is needed.
#include <algorithm>
#include <mutex>
#include <cstdlib>
#include "gbx_internal.h"
#include "poa_graph.h"

namespace gbx {
namespace {

constexpr int SNEG = -(1 << 29);          // identity of the max-plus scan (no overflow when path weights are added)

struct Mat2 { int a, b, c, d; };           // [[a,b],[c,d]] in max-plus
__host__ __device__ inline Mat2 mp_mul(const Mat2 &x, const Mat2 &y)
{
    Mat2 r;
    r.a = max(x.a + y.a, x.b + y.c); r.b = max(x.a + y.b, x.b + y.d);
    r.c = max(x.c + y.a, x.d + y.c); r.d = max(x.c + y.b, x.d + y.d);
    return r;
}
__host__ __device__ inline Mat2 mp_identity() { Mat2 r = {0, SNEG, SNEG, 0}; return r; }
__host__ __device__ inline Mat2 mp_pow(Mat2 base, int k)
{
    Mat2 r = mp_identity();
    while (k > 0) { if (k & 1) r = mp_mul(r, base); base = mp_mul(base, base); k >>= 1; }
    return r;
}
__device__ inline void mp_apply(const Mat2 &m, int E, int Q, int &oe, int &oq)
{
    oe = max(m.a + E, m.b + Q); oq = max(m.c + E, m.d + Q);
}

template <int CTRL, int ROWMASK = 0xf>
__device__ inline int dpp_i(int old, int x) { return __builtin_amdgcn_update_dpp(old, x, CTRL, ROWMASK, 0xf, false); }

struct PoaArgs {
    int64_t n_windows;
    const int64_t *win_first_seq, *seq_off;
    const int32_t *seq_len;
    const uint8_t *arena;
    uint8_t *cons; int32_t *cons_len; int32_t *status; int64_t cons_stride;
    char *work; int64_t slot_bytes;
    unsigned long long *cells;        // counter block: [0] DP cells (graph nodes x sequence length, summed over alignments), [1..13] phase
                                      // statistics, [14] windows of the main launch, [15] its cursor, [16] long windows, [17] their cursor
    const int32_t *wlist;             // this launch's windows, heaviest class first
    int cnt_idx, cur_idx;             // which counters of the block are this launch's
    int ncap, deg, lmax;
    int lds_marks;                    // 1: mark/check/DFS stack live in LDS (dynamic shared memory)
    int lds_stack;                    // entries of the DFS stack in LDS (poa_lds_plan)
    int lds_ncap;                     // nodes the sort's per-node LDS arrays hold (<= ncap): a window that outgrows it sorts in global memory
    PoaScore S;
    Mat2 Tc[2][4];                    // [CPL 8 | CPL 16][(T^CPL)^(1,2,4,8)]: uniform factors of the row_shr scan steps
};

// Row stride of the pipelined DP: 64 lanes x 8 columns always exist in memory, so every lane issues every
// load and store of a row (a fixed number of memory instructions per iteration lets the compiler wait for
// exactly the loads it needs instead of for everything outstanding, stores included).
constexpr int POA_PIPE_STRIDE = 8 + 512;
__host__ __device__ inline int poa_cap_stride(int lmax) { const int s = poa_row_stride(lmax); return s > POA_PIPE_STRIDE ? s : POA_PIPE_STRIDE; }

// byte offsets of the arrays inside one workspace slot
struct SlotLayout {
    int64_t code, in_cnt, out_cnt, aln_cnt, out_slot, out_slot_x, mark, check, decoder, coder;
    int64_t in_src, in_wt, out_dst, in_src_x, in_wt_x, out_dst_x, aln, r2n, n2r, stack, score, pred, path_node, path_pos, mat, total;
    int64_t hdr, st8save;      // lock-step form: the window's scalars and the sort's state bytes between launches
    int stk_cap, path_cap;
};

__host__ __device__ inline int64_t align_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }

__host__ __device__ inline SlotLayout make_layout(int ncap, int deg, int lmax, bool long_slot, int cell_bytes = (int)sizeof(poa_cell_t))
{
    SlotLayout L;
    int64_t o = 0;
    auto take = [&](int64_t bytes) { int64_t r = o; o = align_up(o + bytes, 16); return r; };
    L.stk_cap = ncap * 4 + 64;
    L.path_cap = ncap + lmax + 8;
    L.code = take(ncap); L.in_cnt = take(ncap); L.out_cnt = take(ncap); L.aln_cnt = take(ncap);
    L.out_slot = take((int64_t)ncap * 4); L.out_slot_x = take((int64_t)ncap * (deg - 4) + 16); L.mark = take(ncap); L.check = take(ncap);
    L.decoder = take(256); L.coder = take(512);
    L.hdr = take(64); L.st8save = take(ncap);
    L.in_src = take((int64_t)ncap * 16); L.in_wt = take((int64_t)ncap * 16); L.out_dst = take((int64_t)ncap * 16);
    L.in_src_x = take((int64_t)ncap * (deg - 4) * 4 + 16); L.in_wt_x = take((int64_t)ncap * (deg - 4) * 4 + 16);
    L.out_dst_x = take((int64_t)ncap * (deg - 4) * 4 + 16); L.aln = take((int64_t)ncap * POA_ALN_STRIDE * 4 + 64);
    L.r2n = take((int64_t)ncap * 4); L.n2r = take((int64_t)ncap * 4);
    L.stack = take((int64_t)L.stk_cap * 4); L.score = take((int64_t)ncap * 4); L.pred = take((int64_t)ncap * 4);
    L.path_node = take((int64_t)L.path_cap * 4); L.path_pos = take((int64_t)L.path_cap * 4);
#ifndef GBX_POA_DP_WAVES
#define GBX_POA_DP_WAVES 5          // wavefronts per SIMD of the lock-step DP kernel's default instance
#endif
#ifndef GBX_POA_SERIAL_WAVES
#define GBX_POA_SERIAL_WAVES 3      // ... and the serial-phase kernel (its LDS admits twelve windows per CU)
#endif
#ifndef GBX_POA_LONG_WAVES
#define GBX_POA_LONG_WAVES 1        // the long-window instance: wavefronts per SIMD it is compiled for,
#endif
#ifndef GBX_POA_LONG_RING
#define GBX_POA_LONG_RING 1         // ... whether its pipelined DP (sequences up to 512 columns) uses the row ring,
#endif
#ifndef GBX_POA_LONG_INC
#define GBX_POA_LONG_INC 1          // ... and whether its topological sort is the incremental one
#endif
#ifndef GBX_POA_TEAM_WAVES
#define GBX_POA_TEAM_WAVES 3        // the team kernel (a window per workgroup of four wavefronts): wavefronts per SIMD of the main-list instance
#endif
#ifndef GBX_POA_TEAM_LONG_WAVES
#define GBX_POA_TEAM_LONG_WAVES 1   // ... and of the long-window instance (column-block DP for sequences over 512 bases)
#endif
#ifndef GBX_POA_PLANES
#define GBX_POA_PLANES 2
#endif
    // pipelined DP (sequences up to 512 columns): the H plane + one plane of (H-F, H-O) byte pairs.  Only the slots of the
    // second launch (windows that hold a longer sequence: column-block DP) keep spoa's five int16 planes.
    L.mat = take((int64_t)(ncap + 1) * (long_slot ? poa_cap_stride(lmax) : POA_PIPE_STRIDE) * (long_slot ? 5 : GBX_POA_PLANES) * (int64_t)cell_bytes + 64);
    L.total = align_up(o, 256);
    return L;
}

// ---- DP of one sequence against the graph: fills M, returns the best sink cell -------------
// Row descriptors (first predecessor's row, in-degree, letter, sink flag) are built by a parallel
// pre-pass into g.score / g.pred (free until the consensus), so the row loop starts without a
// pointer chase.  Cells are int16 and a lane's CPL columns are one (CPL=8) or two (CPL=16) aligned
// 16-byte vectors, so a predecessor row is read with 3 wide loads per lane and a row is written with 5.
typedef short v8s __attribute__((ext_vector_type(8)));

template <int CPL>
__device__ inline void load_cells(const poa_cell_t *p, int *out)
{
#pragma unroll
    for (int k = 0; k < CPL / 8; ++k) {
        const v8s t = *(const v8s *)(p + 8 * k);
#pragma unroll
        for (int e = 0; e < 8; ++e) out[8 * k + e] = t[e];
    }
}
template <int CPL>
__device__ inline void store_cells(poa_cell_t *p, const int *in)
{
#pragma unroll
    for (int k = 0; k < CPL / 8; ++k) {
        v8s t;
#pragma unroll
        for (int e = 0; e < 8; ++e) t[e] = (short)in[8 * k + e];
        *(v8s *)(p + 8 * k) = t;
    }
}

// int32 cells (the wide path): the lane's 8 columns are two 16-byte vectors
typedef int v4i __attribute__((ext_vector_type(4)));
template <int CPL>
__device__ inline void load_cells(const int32_t *p, int *out)
{
#pragma unroll
    for (int k = 0; k < CPL / 4; ++k) {
        const v4i t = *(const v4i *)(p + 4 * k);
#pragma unroll
        for (int e = 0; e < 4; ++e) out[4 * k + e] = t[e];
    }
}
template <int CPL>
__device__ inline void store_cells(int32_t *p, const int *in)
{
#pragma unroll
    for (int k = 0; k < CPL / 4; ++k) {
        v4i t;
#pragma unroll
        for (int e = 0; e < 4; ++e) t[e] = in[4 * k + e];
        *(v4i *)(p + 4 * k) = t;
    }
}

// (the team's words in LDS: poa_dp_team below, and the column-block DP when a team runs it)
struct PoaTeamSync {
    int prog[4];                 // last DP row (1-based) wavefront w has completed
    int best[4], best_i[4];      // best sink score of wavefront w's rows and its row
    int n_nodes, err, widx, pad_;    // the team's view of the window: graph size and error bits after the last add_alignment, work item
};
typedef __attribute__((address_space(3))) PoaTeamSync lds_team;
typedef __attribute__((address_space(3))) int lds_i32;

// NW > 1 (round 5, the long windows of a team launch): the rows go to the NW wavefronts of the workgroup in turn, a row starts when its
// predecessor rows are complete (sy->prog, as in poa_dp_team) and is complete once its stores are acknowledged - this DP already ends
// every row waiting for them.  n_team = the graph's size (only wavefront 0's PoaGraph knows it).
template <int CPL, int NW = 1, class MT = PoaMatrices>
__device__ void poa_dp(const PoaGraph &g, const MT &M, const PoaArgs &A, const uint8_t *seq, int len,
                       int &max_i, int &max_j, lds_team *sy = nullptr, int wave = 0, int n_team = 0)
{
    static_assert(CPL == 8 || CPL == 16, "a lane owns one or two 16-byte vectors of cells");
    typedef typename MT::cell cell_t;
    constexpr int NEG = MT::neg;              // -infinity of the cell type (poa_graph.h)
    const int lane = threadIdx.x & 63;
    const int tid = NW > 1 ? (int)threadIdx.x : lane;
    constexpr int NT = 64 * NW;
    const PoaScore S = A.S;
    const int Wp = M.Wp;
    const int n = NW > 1 ? n_team : g.n_nodes;
    constexpr int BLK = 64 * CPL;
    const Mat2 *Tc = A.Tc[CPL == 8 ? 0 : 1];
    // lane-dependent max-plus matrices: Tc^(lane&15 + 1), Tc^(lane&31 + 1), Tc^lane
    const Mat2 P16 = mp_pow(Tc[0], (lane & 15) + 1);
    const Mat2 P32 = mp_pow(Tc[0], (lane & 31) + 1);
    const Mat2 PC = mp_pow(Tc[0], lane);

    // row 0 (sisd_alignment_engine `initialize`) and the row descriptors
    for (int j = tid; j <= len; j += NT) {
        const int e0 = j == 0 ? 0 : S.g + (j - 1) * S.e, q0 = j == 0 ? 0 : S.q + (j - 1) * S.c;
        M.E[j + POA_COL0] = (cell_t)e0; M.Q[j + POA_COL0] = (cell_t)q0;
        M.F[j + POA_COL0] = (cell_t)(j == 0 ? 0 : NEG); M.O[j + POA_COL0] = (cell_t)(j == 0 ? 0 : NEG);
        M.H[j + POA_COL0] = (cell_t)(j == 0 ? 0 : max(q0, e0));
    }
    int32_t *d_pred = g.score, *d_info = g.pred;       // [rank] first predecessor row | letter, in-degree, sink
    {
        PoaGraph &gm = const_cast<PoaGraph &>(g);
        for (int r = tid; r < n; r += NT) poa_rowdesc_one(gm, r);
    }
    if (NW > 1) {
        if (wave == 0 && lane < NW) sy->prog[lane] = 0;
        __syncthreads();
    } else {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }

    const bool single = len <= BLK;
    int sq[CPL];
    auto load_seq = [&](int base) {
#pragma unroll
        for (int c = 0; c < CPL; ++c) { const int j = base + lane * CPL + 1 + c; sq[c] = j <= len ? seq[j - 1] : -1; }
    };
    if (single) load_seq(0);

    int best = NEG;
    max_i = -1; max_j = -1;
    const int32_t *d_pred1 = g.path_node, *d_pred2 = g.path_pos;   // 2nd / 3rd predecessor rows (free until the traceback)
    const int r0 = NW > 1 ? wave : 0;
    int nx_pred = n > r0 ? d_pred[r0] : 0, nx_info = n > r0 ? d_info[r0] : 0, nx_p1 = n > r0 ? d_pred1[r0] : 0, nx_p2 = n > r0 ? d_pred2[r0] : 0;
    for (int r = r0; r < n; r += NW) {
        const int i = r + 1;
        const int p0 = nx_pred, info = nx_info, p1 = nx_p1, p2 = nx_p2;
        if (r + NW < n) { nx_pred = d_pred[r + NW]; nx_info = d_info[r + NW]; nx_p1 = d_pred1[r + NW]; nx_p2 = d_pred2[r + NW]; }   // prefetch
        const int letter = info & 0xff, ic = (info >> 8) & 0xff;
        const bool sink = (info >> 16) & 1;
        const int64_t ro = (int64_t)i * Wp + POA_COL0;                             // index of (i, 0)
        // predecessor rows: the first three from the descriptors, any further ones from the in-edge list
        const int node = ic > 3 ? g.r2n[r] : 0;
        auto pred_row = [&](int k) { return k == 0 ? p0 : k == 1 ? p1 : k == 2 ? p2 : g.n2r[PG_IN_SRC(g, node, k)] + 1; };
        if (NW > 1) {
            // wait for the predecessor rows: lane l (mod NW) for those wavefront l mod NW owns
            const int own = lane & (NW - 1);
            auto need_of = [&](int x) { return ((x - 1) & (NW - 1)) == own ? x : 0; };
            int need = 0;
            for (int k = 0; k < ic; ++k) need = max(need, need_of(__builtin_amdgcn_readfirstlane(pred_row(k))));
            lds_i32 *const pm = (lds_i32 *)&sy->prog[own];
            GBX_GUARD(gd_wait, 1 << 24);
            while (__ballot(*(volatile lds_i32 *)pm < need) != 0) {
                if (GBX_GUARD_TRIP(gd_wait, GBX_GK_POA, 6, i)) break;
                __builtin_amdgcn_s_sleep(1);
            }
            asm volatile("" ::: "memory");                       // nothing of the rows waited for is read before this point
        }
        int po = ic == 0 ? S.q - S.c : NEG, pf = ic == 0 ? S.g - S.e : NEG;
        {
            // column 0 of up to three predecessor rows: independent loads, one round trip
            const int64_t i0 = (int64_t)p0 * Wp + POA_COL0, i1 = (int64_t)p1 * Wp + POA_COL0, i2 = (int64_t)p2 * Wp + POA_COL0;
            const int o0 = M.O[i0], f0 = M.F[i0], o1 = M.O[i1], f1 = M.F[i1], o2 = M.O[i2], f2 = M.F[i2];
            if (ic > 0) { po = max(po, o0); pf = max(pf, f0); }
            if (ic > 1) { po = max(po, o1); pf = max(pf, f1); }
            if (ic > 2) { po = max(po, o2); pf = max(pf, f2); }
            for (int k = 3; k < ic; ++k) {
                const int64_t pi = (int64_t)pred_row(k) * Wp + POA_COL0;
                po = max(po, (int)M.O[pi]); pf = max(pf, (int)M.F[pi]);
            }
        }
        const int O0 = po + S.c, F0 = pf + S.e, H0 = max(O0, F0);
        if (lane == 0) {
            M.O[ro] = (cell_t)O0; M.F[ro] = (cell_t)F0; M.H[ro] = (cell_t)H0;
            M.E[ro] = (cell_t)NEG; M.Q[ro] = (cell_t)NEG;
        }
        // (E,Q) at the first column of the block; column 1: E = H0+g, Q = H0+q (E[0] = Q[0] = -inf)
        int cE = H0 + S.g, cQ = H0 + S.q;
        int hlast = 0;                                             // H(i, len) for the sink test
        for (int base = 0; base < len; base += BLK) {
            const int j0 = base + lane * CPL + 1;                  // lane's first column
            const bool mine = j0 <= len;                           // the lane owns at least one real column
            if (!single) load_seq(base);
            int Fa[CPL], Oa[CPL], Ha[CPL];
            for (int p = 0; p < (ic ? ic : 1); ++p) {
                const int64_t po_ = (ic ? (int64_t)pred_row(p) : 0) * Wp + POA_COL0;
                int hp[CPL], fp[CPL], op[CPL];
                if (mine) {
                    load_cells<CPL>(M.H + po_ + j0, hp); load_cells<CPL>(M.F + po_ + j0, fp); load_cells<CPL>(M.O + po_ + j0, op);
                } else {
#pragma unroll
                    for (int c = 0; c < CPL; ++c) { hp[c] = 0; fp[c] = 0; op[c] = 0; }
                }
                const int hfirst = M.H[po_ + base];                // H(pred, base): left neighbour of the block
                int hl = __builtin_amdgcn_update_dpp(hfirst, hp[CPL - 1], 0x138, 0xf, 0xf, false);
                if (lane == 0) hl = hfirst;
#pragma unroll
                for (int c = 0; c < CPL; ++c) {
                    const int sc = sq[c] == letter ? S.m : S.n;
                    const int f = max(hp[c] + S.g, fp[c] + S.e);
                    const int o = max(hp[c] + S.q, op[c] + S.c);
                    const int h = hl + sc;
                    hl = hp[c];
                    if (p == 0) { Fa[c] = f; Oa[c] = o; Ha[c] = h; }
                    else { Fa[c] = max(Fa[c], f); Oa[c] = max(Oa[c], o); Ha[c] = max(Ha[c], h); }
                }
            }
            int Aa[CPL];
#pragma unroll
            for (int c = 0; c < CPL; ++c) Aa[c] = mine ? max(Ha[c], max(Fa[c], Oa[c])) : SNEG;
            // ---- pass 1: lane-local recurrence from the identity -> this lane's contribution b
            int bE = SNEG, bQ = SNEG;
#pragma unroll
            for (int c = 0; c < CPL; ++c) {
                const int h = max(Aa[c], max(bE, bQ));
                const int ne = max(h + S.g, bE + S.e), nq = max(h + S.q, bQ + S.c);
                bE = ne; bQ = nq;
            }
            // ---- inclusive scan over lanes: x[l] = max_k<=l Tc^(l-k) (x) b[k]
            int xE = bE, xQ = bQ, tE, tQ;
            mp_apply(Tc[0], dpp_i<0x111>(SNEG, xE), dpp_i<0x111>(SNEG, xQ), tE, tQ); xE = max(xE, tE); xQ = max(xQ, tQ);
            mp_apply(Tc[1], dpp_i<0x112>(SNEG, xE), dpp_i<0x112>(SNEG, xQ), tE, tQ); xE = max(xE, tE); xQ = max(xQ, tQ);
            mp_apply(Tc[2], dpp_i<0x114>(SNEG, xE), dpp_i<0x114>(SNEG, xQ), tE, tQ); xE = max(xE, tE); xQ = max(xQ, tQ);
            mp_apply(Tc[3], dpp_i<0x118>(SNEG, xE), dpp_i<0x118>(SNEG, xQ), tE, tQ); xE = max(xE, tE); xQ = max(xQ, tQ);
            mp_apply(P16, dpp_i<0x142, 0xa>(SNEG, xE), dpp_i<0x142, 0xa>(SNEG, xQ), tE, tQ); xE = max(xE, tE); xQ = max(xQ, tQ);
            mp_apply(P32, dpp_i<0x143, 0xc>(SNEG, xE), dpp_i<0x143, 0xc>(SNEG, xQ), tE, tQ); xE = max(xE, tE); xQ = max(xQ, tQ);
            // (E,Q) entering this lane = previous lane's inclusive value (+) Tc^lane (x) block carry
            int vE = dpp_i<0x138>(SNEG, xE), vQ = dpp_i<0x138>(SNEG, xQ);
            if (lane == 0) { vE = SNEG; vQ = SNEG; }
            mp_apply(PC, cE, cQ, tE, tQ);
            vE = max(vE, tE); vQ = max(vQ, tQ);
            // carry for the next block: (E,Q) leaving lane 63
            if (!single) {
                int oE, oQ;
                mp_apply(Tc[0], tE, tQ, oE, oQ);               // Tc^(lane+1) (x) carry
                const int lE = max(xE, oE), lQ = max(xQ, oQ);
                cE = __builtin_amdgcn_readlane(lE, 63); cQ = __builtin_amdgcn_readlane(lQ, 63);
            }
            // ---- pass 2: exact E, Q, H of the lane's columns; store the row
            int Hn[CPL], En[CPL], Qn[CPL];
#pragma unroll
            for (int c = 0; c < CPL; ++c) {
                const int h = max(Aa[c], max(vE, vQ));
                Hn[c] = h; En[c] = vE; Qn[c] = vQ;
                const int ne = max(h + S.g, vE + S.e), nq = max(h + S.q, vQ + S.c);
                vE = ne; vQ = nq;
            }
            if (mine) {
                store_cells<CPL>(M.H + ro + j0, Hn); store_cells<CPL>(M.F + ro + j0, Fa); store_cells<CPL>(M.O + ro + j0, Oa);
                store_cells<CPL>(M.E + ro + j0, En); store_cells<CPL>(M.Q + ro + j0, Qn);
            }
            if (sink && base + BLK >= len) {                       // H(i, len) sits in this block
                const int cl = (len - 1 - base) % CPL;
                int hv = Hn[0];
#pragma unroll
                for (int c = 1; c < CPL; ++c) hv = c == cl ? Hn[c] : hv;
                hlast = __builtin_amdgcn_readlane(hv, (len - 1 - base) / CPL);
            }
        }
        if (NW > 1) {
            __builtin_amdgcn_s_waitcnt(0x0F70);                  // vmcnt(0): the row is in memory for the other wavefronts
            asm volatile("" ::: "memory");
            if (lane == 0) *(volatile lds_i32 *)&sy->prog[wave] = i;
        } else {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        if (sink && best < hlast) { best = hlast; max_i = i; max_j = len; }       // NW: best sink at the last column
    }
    if (NW > 1) {
        // the team's best sink: ties go to the first row in topological order (the serial loop's strict `<`)
        if (lane == 0) { sy->best[wave] = best; sy->best_i[wave] = max_i; }
        __syncthreads();
        int b = NEG, bi = -1;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const int v = *(volatile lds_i32 *)&sy->best[w], vi = *(volatile lds_i32 *)&sy->best_i[w];
            if (vi != -1 && (bi == -1 || v > b || (v == b && vi < bi))) { b = v; bi = vi; }
        }
        max_i = bi; max_j = bi == -1 ? -1 : len;
    }
}

// ---- the same DP with the row loop software-pipelined (sequences of at most 512 columns) ------------
// A row costs three dependent memory round trips in poa_dp (column 0 of the predecessors, their rows, and
// the wait for its own stores before the next row may read them), far more than its arithmetic.  Here the
// inputs of row r+1 (rows of its first two predecessors, packed int16, plus their column 0) are requested
// while row r is still being computed and *before* row r is stored, so they are not queued behind those
// stores; a predecessor that is row r itself is handed over in registers.  Vector memory operations of one
// wavefront are performed in order, so a row stored in an earlier iteration is visible to these loads
// without a fence.

// ---- packed int16 arithmetic of the pipelined DP --------------------------------------------------
// Cells are int16 in memory; the row arithmetic stays in packed int16 (two columns, or the (E,Q) pair,
// per register; v_pk_add_i16 with clamp, v_pk_max_i16, op_sel for swaps and broadcasts): half the
// instructions and registers of the int32 form, and no unpack / pack.  -32768 is the identity of the
// max-plus scan (saturating adds keep it there); real scores stay above -30000 (host plan), so the
// results are the integers of the int32 formulation.
typedef short v2s __attribute__((ext_vector_type(2)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));
// F and O live in memory as gap deficits: dF = H - F, dO = H - O, one byte each.  They are small by construction:
// H(i,j) <= max_p H(p,j) + (m - g) over the row's predecessors p (diagonal, vertical and horizontal moves alike, by
// induction over the columns), and F(i,j) >= max_p H(p,j) + g, O(i,j) >= max_p H(p,j) + q, hence 0 <= H - F <= m - 2g
// and 0 <= H - O <= m - g - q (14 and 33 with the driver's scores; the host plan checks they fit a byte).  Row 0 is
// the exception (F = O = -infinity there): its deficits are stored as 255, which is as good as -infinity in every
// use: a successor's F = max(H + g, F + e) never takes the second term once H - F >= e - g, and no equality test of
// the traceback against a value that low can hold (each is bounded below by H + g or H + q of the same cell).
// 4 instead of 6 bytes per cell written, and read, per predecessor row.
struct PoaPredIn { v8s h; v4u fo; int h0, o0, f0; };      // fo = the lane's 8 bytes of H - F (x, y) and 8 of H - O (z, w), as stored
constexpr int POA_C0_F = -3, POA_C0_O = -2;       // column-0 F and O of a row sit in H-plane pad cells before column 0: {F0, O0, -, H0} is one aligned 8-byte group
constexpr short PK_NEG = -32768;
__device__ inline v2s pk2(int lo, int hi) { v2s r; r.x = (short)lo; r.y = (short)hi; return r; }
__device__ inline v2s pk_sat(int lo, int hi) { return pk2(max(lo, -32768), max(hi, -32768)); }      // clamp of -inf entries
__device__ inline v2s pk_add(v2s a, v2s b) { return __builtin_elementwise_add_sat(a, b); }
__device__ inline v2s pk_max(v2s a, v2s b) { return __builtin_elementwise_max(a, b); }
__device__ inline v2s pk_swap(v2s a) { return __builtin_shufflevector(a, a, 1, 0); }
__device__ inline v2s pk_lo(v2s a) { return __builtin_shufflevector(a, a, 0, 0); }
__device__ inline v2s pk_hi(v2s a) { return __builtin_shufflevector(a, a, 1, 1); }
__device__ inline unsigned pk_bits(v2s a) { return __builtin_bit_cast(unsigned, a); }
__device__ inline v2s pk_from(unsigned u) { return __builtin_bit_cast(v2s, u); }
template <int K> __device__ inline v2s pk_pair(const v8s &x) { return __builtin_shufflevector(x, x, 2 * K, 2 * K + 1); }
__device__ inline v8s pk_join(v2s a, v2s b, v2s c, v2s d)
{
    typedef short v4s __attribute__((ext_vector_type(4)));
    const v4s lo = __builtin_shufflevector(a, b, 0, 1, 2, 3), hi = __builtin_shufflevector(c, d, 0, 1, 2, 3);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
template <int CTRL, int ROWMASK = 0xf>
__device__ inline v2s pk_dpp(v2s old, v2s x)
{
    return pk_from((unsigned)__builtin_amdgcn_update_dpp((int)pk_bits(old), (int)pk_bits(x), CTRL, ROWMASK, 0xf, false));
}
// 2x2 max-plus matrix as two packed columns: AC = (a, c), BD = (b, d);  m (x) (E,Q) = max(AC + (E,E), BD + (Q,Q))
struct PkMat { v2s ac, bd; };
__device__ inline PkMat pk_mat(const Mat2 &m) { PkMat r; r.ac = pk_sat(m.a, m.c); r.bd = pk_sat(m.b, m.d); return r; }
__device__ inline v2s pk_apply(const PkMat &m, v2s eq) { return pk_max(pk_add(m.ac, pk_lo(eq)), pk_add(m.bd, pk_hi(eq))); }

// RING (round 4): the last POA_RING_ROWS rows of the matrix also stay in LDS.  A partial-order graph of thirty noisy reads is
// wide: the topological order keeps the 2-7 letters of a column side by side, so a row's first predecessor is the row
// before it in only 22 % of the cases - but within six rows in 97 % (second predecessors, 38 % of the rows have one: 95 %;
// third ones, 18 %: 94 %; measured on the 'large' generator with the host model of the graph).  Without the ring every
// such row came back from HBM: 1.15 row reads per row written, 4.6 of the kernel's 8.8 B per cell, and a DP phase bound
// by HBM traffic (a launch of nothing but DP rows moves 3.7 TB/s, as much as the mixed kernel) with a third predecessor
// costing a synchronous round trip in the middle of its row.  The ring shares the window's LDS with the topological sort's
// arrays - the two are never live together; the sort's state bytes are parked in the slot meanwhile (st8save), its
// previous ranks are n2r - so the twelve windows per CU stay.
constexpr int POA_RING_SLOT = 64 * 16 * 2 + 16;            // H vectors, deficit vectors (16 B per lane each), {F0, O0, -, H0}
constexpr int POA_RING_DEFAULT = 6;                        // rows: 12.4 KB, what the sort's arrays take for 'large'
constexpr int POA_RING_BYTES = POA_RING_DEFAULT * POA_RING_SLOT;
typedef __attribute__((address_space(3))) v8s lds_v8s;
typedef __attribute__((address_space(3))) v4u lds_v4u;
typedef __attribute__((address_space(3))) v2u lds_v2u;

template <int POA_RING_ROWS, bool DESC_DONE = false>        // POA_RING_ROWS 0: no ring; DESC_DONE: the row descriptors are in place (poa_serial_call)
__device__ __attribute__((always_inline)) void poa_dp_pipelined(const PoaGraph &g, const PoaMatrices &M, const PoaArgs &A, const uint8_t *seq, int len,
                                 int &max_i, int &max_j, char *lds_ring = nullptr)
{
    constexpr int CPL = 8;
    constexpr bool RING = POA_RING_ROWS > 0;
    constexpr int RROWS = RING ? POA_RING_ROWS : 1;
    const int lane = threadIdx.x & 63;
    const PoaScore S = A.S;
    const int Wp = M.Wp;
    const int n = g.n_nodes;
    const Mat2 *Tc = A.Tc[0];
    const PkMat T0 = pk_mat(Tc[0]), T1 = pk_mat(Tc[1]), T2 = pk_mat(Tc[2]), T3 = pk_mat(Tc[3]);
    const PkMat P16 = pk_mat(mp_pow(Tc[0], (lane & 15) + 1));
    const PkMat P32 = pk_mat(mp_pow(Tc[0], (lane & 31) + 1));
    const PkMat PC = pk_mat(mp_pow(Tc[0], lane));
    const v2s G2 = pk2(S.g, S.g), E2 = pk2(S.e, S.e), Q2 = pk2(S.q, S.q), C2 = pk2(S.c, S.c);
    const v2s GQ = pk2(S.g, S.q), EC = pk2(S.e, S.c), NM = pk2(S.n - S.m, S.n - S.m), MM = pk2(S.m, S.m);
    const v2s NEG2 = pk2(PK_NEG, PK_NEG);

    for (int j = lane; j <= len; j += 64) {                    // row 0 (`initialize`)
        const int e0 = j == 0 ? 0 : S.g + (j - 1) * S.e, q0 = j == 0 ? 0 : S.q + (j - 1) * S.c;
        M.H[j + POA_COL0] = (poa_cell_t)(j == 0 ? 0 : max(q0, e0));
    }
    uint8_t *const FO = (uint8_t *)M.F;                        // the second plane: per lane 8 bytes of H-F, then 8 of H-O
    for (int j = lane; j < POA_PIPE_STRIDE * 2; j += 64) FO[j] = 255;          // row 0: F = O = -infinity
    if (lane == 0) { M.H[POA_COL0 + POA_C0_F] = 0; M.H[POA_COL0 + POA_C0_O] = 0; }
    int32_t *d_pred = g.score, *d_info = g.pred;
    const int32_t *d_pred1 = g.path_node, *d_pred2 = g.path_pos;
    int32_t *d_pred3 = g.stack;                                // RING: the 4th predecessor's row (the sort's order buffer is free during the DP)
    if (!DESC_DONE) {
        PoaGraph &gm = const_cast<PoaGraph &>(g);
        for (int r = lane; r < n; r += 64) {
            poa_rowdesc_one(gm, r);
            if (RING) {
                const int node = g.r2n[r];
                d_pred3[r] = g.in_cnt[node] > 3 ? g.n2r[PG_IN_SRC(g, node, 3)] + 1 : 0;
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    const int j0 = lane * CPL + 1;                             // the lane's first column
    const bool mine = j0 <= len;                               // rows are POA_PIPE_STRIDE wide: columns beyond len exist, hold garbage
    v2s sqp[4];                                                // sequence letters of the lane's columns, two per register
    {
        int sq[CPL];
#pragma unroll
        for (int c = 0; c < CPL; ++c) { const int j = j0 + c; sq[c] = j <= len ? seq[j - 1] : 0x7fff; }
#pragma unroll
        for (int k = 0; k < 4; ++k) sqp[k] = pk2(sq[2 * k], sq[2 * k + 1]);
    }

    auto wait_vm = [] { __builtin_amdgcn_s_waitcnt(0x0F70); };   // vmcnt(0) only (lgkmcnt 15, expcnt 7: not waited for)
    auto fetch = [&](int prow, PoaPredIn &x) {
        const int64_t b = (int64_t)prow * Wp + POA_COL0;
        x.h = *(const v8s *)(M.H + b + j0);
        x.fo = *(const v4u *)(M.F + b + j0);
        const v2u c0 = *(const v2u *)(M.H + b + POA_C0_F);      // {F0, O0 | pad, H0}
        x.f0 = (int)(short)(c0.x & 0xffffu); x.o0 = (int)(short)(c0.x >> 16); x.h0 = (int)(short)(c0.y >> 16);
    };
    // row `prow` out of the ring (it was written at most POA_RING_ROWS rows ago and not overwritten since)
    char *const ring_lane = lds_ring + lane * 16;
    auto ring_fetch = [&](int prow, PoaPredIn &x) {
        char *const sl = ring_lane + (prow % RROWS) * POA_RING_SLOT;
        x.h = *(const lds_v8s *)sl;
        x.fo = *(const lds_v4u *)(sl + 1024);
        const v2u c0 = *(const lds_v2u *)(lds_ring + (prow % RROWS) * POA_RING_SLOT + 2048);
        x.f0 = (int)(short)(c0.x & 0xffffu); x.o0 = (int)(short)(c0.x >> 16); x.h0 = (int)(short)(c0.y >> 16);
    };
    // a predecessor row for the row `target` (prow < target, and not the row being computed): ring or memory
    auto fetch_pred = [&](int prow_v, int computing, PoaPredIn &x) {
        const int prow = __builtin_amdgcn_readfirstlane(prow_v);       // the same in every lane: a scalar branch, scalar slot arithmetic
        if (RING && prow >= 1 && computing - prow <= RROWS) ring_fetch(prow, x);
        else {
            fetch(prow, x);
            if (RING) { wait_vm(); asm volatile("" : "+v"(x.h), "+v"(x.fo), "+v"(x.h0), "+v"(x.o0), "+v"(x.f0)); }
        }
    };
    // RING: the row loop issues NO vector-memory load in its steady state.  gfx9 counts loads and stores in one in-order
    // counter, so a wait for any load is also a wait for the acknowledgement of every store issued before it: with the
    // descriptors of rows r+2 requested every iteration, and a wait for them every iteration, each row also waited for the
    // HBM acknowledgement of the row stored before it (the ISA had four `s_waitcnt vmcnt(0)` per row).  The descriptors of
    // 64 consecutive rows now sit in five registers (lane k = row qbase + k; a row's values are v_readlane'd, i.e. scalar),
    // refilled once per 62 rows; a predecessor row is the previous row's registers, an LDS read of the ring, or - 3 % - a
    // load that is waited for on the spot, inside its own branch, so that the merged code carries no wait.
    int q_p0 = 0, q_p1 = 0, q_p2 = 0, q_p3 = 0, q_info = 0, qbase = 0;
    auto refill = [&](int base) {
        qbase = base;
        const int rr = min(base + lane, n - 1);
        q_p0 = d_pred[rr]; q_p1 = d_pred1[rr]; q_p2 = d_pred2[rr]; q_p3 = d_pred3[rr]; q_info = d_info[rr];
        wait_vm();
        asm volatile("" : "+v"(q_p0), "+v"(q_p1), "+v"(q_p2), "+v"(q_p3), "+v"(q_info));
    };
    auto desc = [&](int r, int &p0, int &p1, int &p2, int &info) {
        const int rr = min(r, n - 1);
        if (RING) {
            const int l = rr - qbase;
            p0 = __builtin_amdgcn_readlane(q_p0, l); p1 = __builtin_amdgcn_readlane(q_p1, l); p2 = __builtin_amdgcn_readlane(q_p2, l);
            info = __builtin_amdgcn_readlane(q_info, l);
        } else {
            p0 = d_pred[rr]; p1 = d_pred1[rr]; p2 = d_pred2[rr]; info = d_info[rr];
        }
    };
    auto settle_i = [](int &v) { asm volatile("" : "+v"(v)); };
    auto settle_in = [&](PoaPredIn &x) {
        asm volatile("" : "+v"(x.h), "+v"(x.fo));
        settle_i(x.h0); settle_i(x.o0); settle_i(x.f0);
    };
    // F, O and diagonal-H contributions of one predecessor row to the lane's columns (packed pairs)
    auto pred_terms = [&](const PoaPredIn &x, const v2s (&sc)[4], v2s (&F)[4], v2s (&O)[4], v2s (&H)[4]) {
        const v2s hp[4] = {pk_pair<0>(x.h), pk_pair<1>(x.h), pk_pair<2>(x.h), pk_pair<3>(x.h)};
        // byte k of a dword -> zero-extended int16, two per register: one v_perm_b32 per pair of columns
        const v2s dfp[4] = {pk_from(__builtin_amdgcn_perm(0u, x.fo.x, 0x0c010c00u)), pk_from(__builtin_amdgcn_perm(0u, x.fo.x, 0x0c030c02u)),
                            pk_from(__builtin_amdgcn_perm(0u, x.fo.y, 0x0c010c00u)), pk_from(__builtin_amdgcn_perm(0u, x.fo.y, 0x0c030c02u))};
        const v2s dqp[4] = {pk_from(__builtin_amdgcn_perm(0u, x.fo.z, 0x0c010c00u)), pk_from(__builtin_amdgcn_perm(0u, x.fo.z, 0x0c030c02u)),
                            pk_from(__builtin_amdgcn_perm(0u, x.fo.w, 0x0c010c00u)), pk_from(__builtin_amdgcn_perm(0u, x.fo.w, 0x0c030c02u))};
        // H(pred, j0-1): high half of the previous lane's last pair, or column 0 for lane 0
        unsigned left = (unsigned)__builtin_amdgcn_update_dpp(0, (int)pk_bits(hp[3]), 0x138, 0xf, 0xf, false);
        if (lane == 0) left = (unsigned)x.h0 << 16;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            // max(H + g, F + e) = H + max(g, e - (H - F)), likewise for O: the deficits are used as they are stored
            F[k] = pk_add(hp[k], pk_max(G2, E2 - dfp[k]));
            O[k] = pk_add(hp[k], pk_max(Q2, C2 - dqp[k]));
            const unsigned prev = k ? pk_bits(hp[k - 1]) : left;
            const v2s hs = pk_from(__builtin_amdgcn_alignbit(pk_bits(hp[k]), prev, 16));      // (H[j-1]) of the pair's columns
            H[k] = pk_add(hs, sc[k]);
        }
    };

    int best = POA_NEG_INF;
    max_i = -1; max_j = -1;
    if (n == 0) return;
    // descriptors: row r (a*), row r+1 (b*), row r+2 is requested inside the loop
    // (RING: none of this pipeline - a row reads its own descriptor out of the registers and its predecessors out of the ring at
    // its top; the previous row is in the ring like any other, so the eleven-register hand-over of `last` and the 22 selects per
    // row that chose between it and the prefetch registers are gone, and nothing but the ring lives across the iterations)
    int ap0 = 0, ap1 = 0, ap2 = 0, ainfo = 0, bp0 = 0, bp1 = 0, bp2 = 0, binfo = 0;
    if (RING) refill(0);
    else {
        desc(0, ap0, ap1, ap2, ainfo);
        desc(1, bp0, bp1, bp2, binfo);
    }
    PoaPredIn in0, in1, last;
    last.h = (v8s)0; last.fo = (v4u)0;
    last.h0 = last.o0 = last.f0 = 0;
    in0 = last; in1 = last;
    bool reg0 = false, reg1 = false;                           // predecessor k of the row at hand is the previous row
    if (!RING) {
        const int ic = (ainfo >> 8) & 0xff;
        fetch(ic ? ap0 : 0, in0);
        fetch(ic > 1 ? ap1 : 0, in1);
        // nothing is in flight when the loop is entered: the loop's own issue order is then the only one the
        // compiler has to reason about
        settle_i(ap0); settle_i(ap1); settle_i(ap2); settle_i(ainfo); settle_i(bp0); settle_i(bp1); settle_i(bp2); settle_i(binfo);
        settle_in(in0); settle_in(in1);
    }
    for (int r = 0; r < n; ++r) {
        const int i = r + 1;
        int p2, info;
        if (RING) {
            if (r >= qbase + 64) refill(r);
            const int l = r - qbase;
            const int sp0 = __builtin_amdgcn_readlane(q_p0, l), sp1 = __builtin_amdgcn_readlane(q_p1, l);
            p2 = __builtin_amdgcn_readlane(q_p2, l); info = __builtin_amdgcn_readlane(q_info, l);
            const int ric = (info >> 8) & 0xff;
            fetch_pred(ric >= 1 ? sp0 : 0, i, in0);           // (a row without predecessors hangs off the start row)
            if (ric >= 2) fetch_pred(sp1, i, in1);
        } else {
            p2 = ap2; info = __builtin_amdgcn_readfirstlane(ainfo);      // the row's descriptor is the same in every lane: scalar, so that what follows branches
        }
        const int letter = info & 0xff, ic = (info >> 8) & 0xff;
        const bool sink = (info >> 16) & 1;
        const int64_t ro = (int64_t)i * Wp + POA_COL0;
        int cp0 = 0, cp1 = 0, cp2 = 0, cinfo = 0;              // row r+2
        if (!RING) desc(r + 2, cp0, cp1, cp2, cinfo);

        // match / mismatch score of the lane's columns against this row's letter: m + (n-m) * (seq != letter)
        v2s sc[4];
        {
            const v2s LL = pk2(letter, letter);
#pragma unroll
            for (int k = 0; k < 4; ++k) sc[k] = __builtin_elementwise_min(sqp[k] ^ LL, pk2(1, 1)) * NM + MM;
        }
        // ---- inputs of this row out of the prefetch registers (or the previous row's registers)
        v2s Fa[4], Oa[4], Ha[4];
        int po = ic == 0 ? S.q - S.c : POA_NEG_INF, pf = ic == 0 ? S.g - S.e : POA_NEG_INF;
        {
            const PoaPredIn &x = RING ? in0 : reg0 ? last : in0;
            if (ic > 0) { po = max(po, x.o0); pf = max(pf, x.f0); }
            pred_terms(x, sc, Fa, Oa, Ha);
        }
        if (ic > 1) {
            // second predecessor: its registers are requested for every row (a fixed number of loads per iteration; the
            // start row stands in when there is none) and used only here, under a scalar branch (258 instead of 265 ms on
            // 'large'; branching on "the predecessor is the previous row" as well, instead of selecting, gave part of that back)
            const PoaPredIn &x = RING ? in1 : reg1 ? last : in1;
            po = max(po, x.o0); pf = max(pf, x.f0);
            v2s F2[4], O2[4], H2[4];
            pred_terms(x, sc, F2, O2, H2);
#pragma unroll
            for (int k = 0; k < 4; ++k) { Fa[k] = pk_max(Fa[k], F2[k]); Oa[k] = pk_max(Oa[k], O2[k]); Ha[k] = pk_max(Ha[k], H2[k]); }
        }
        // ---- request the inputs of row r+1 now: before this row's stores, behind those of the rows before.
        // Always both predecessors (the start row stands in for a missing one; a predecessor that is this very
        // row is read as well and ignored): a fixed number of loads per iteration.
        const int nic = (binfo >> 8) & 0xff;
        const bool nreg0 = r + 1 < n && nic >= 1 && bp0 == i, nreg1 = r + 1 < n && nic >= 2 && bp1 == i;
        if (RING) {
            // (nothing: the next row fetches its own inputs at its top)
        } else {
            fetch(nic >= 1 && !nreg0 ? bp0 : 0, in0);
            fetch(nic >= 2 && !nreg1 ? bp1 : 0, in1);
        }
        // ---- predecessors beyond the second: read in place (rare)
        if (ic > 2) {
            // (RING: the third and fourth come from the descriptor registers; a fifth and further ones - rarer still - walk the
            // in-edge list in memory and wait for it here, inside the branch)
            const int p3 = RING ? __builtin_amdgcn_readlane(q_p3, r - qbase) : 0;
            int node = 0;
            if (!RING || ic > 4) { node = g.r2n[r]; if (RING) { wait_vm(); asm volatile("" : "+v"(node)); } }
            for (int k = 2; k < ic; ++k) {
                int prow;
                if (k == 2) prow = p2;
                else if (RING && k == 3) prow = p3;
                else { prow = g.n2r[PG_IN_SRC(g, node, k)] + 1; if (RING) { wait_vm(); asm volatile("" : "+v"(prow)); } }
                PoaPredIn x;
                fetch_pred(prow, i, x);
                po = max(po, x.o0); pf = max(pf, x.f0);
                v2s F2[4], O2[4], H2[4];
                pred_terms(x, sc, F2, O2, H2);
#pragma unroll
                for (int q = 0; q < 4; ++q) { Fa[q] = pk_max(Fa[q], F2[q]); Oa[q] = pk_max(Oa[q], O2[q]); Ha[q] = pk_max(Ha[q], H2[q]); }
            }
        }
        const int O0 = po + S.c, F0 = pf + S.e, H0 = max(O0, F0);
        // column 0: every lane stores the same values to the same cells
        if (lane == 0) {                                       // column 0: one 8-byte store
            v2u c0;
            c0.x = ((unsigned)F0 & 0xffffu) | ((unsigned)O0 << 16); c0.y = (unsigned)H0 << 16;
            *(v2u *)(M.H + ro + POA_C0_F) = c0;
        }
        const v2s cEQ = pk2(H0 + S.g, H0 + S.q);               // (E,Q) entering column 1
        v2s Aa[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) Aa[k] = mine ? pk_max(Ha[k], pk_max(Fa[k], Oa[k])) : NEG2;
        // pass 1: lane-local (E,Q) recurrence from the identity; column c sits in half c&1 of pair c>>1
        v2s bEQ = NEG2;
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            const v2s a = (c & 1) ? pk_hi(Aa[c >> 1]) : pk_lo(Aa[c >> 1]);
            const v2s h = pk_max(a, pk_max(bEQ, pk_swap(bEQ)));
            bEQ = pk_max(pk_add(h, GQ), pk_add(bEQ, EC));
        }
        // inclusive scan over the lanes: x[l] = max_k<=l Tc^(l-k) (x) b[k]
        v2s x = bEQ;
        x = pk_max(x, pk_apply(T0, pk_dpp<0x111>(NEG2, x)));
        x = pk_max(x, pk_apply(T1, pk_dpp<0x112>(NEG2, x)));
        x = pk_max(x, pk_apply(T2, pk_dpp<0x114>(NEG2, x)));
        x = pk_max(x, pk_apply(T3, pk_dpp<0x118>(NEG2, x)));
        x = pk_max(x, pk_apply(P16, pk_dpp<0x142, 0xa>(NEG2, x)));
        x = pk_max(x, pk_apply(P32, pk_dpp<0x143, 0xc>(NEG2, x)));
        v2s vEQ = pk_dpp<0x138>(NEG2, x);                      // (E,Q) entering this lane from the lanes before
        if (lane == 0) vEQ = NEG2;
        vEQ = pk_max(vEQ, pk_apply(PC, cEQ));                  // (+) Tc^lane (x) the (E,Q) entering column 1
        // pass 2: exact H (E, Q are rebuilt by the traceback on demand)
        v2s hcol[CPL];                                         // both halves = H of column c
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            const v2s a = (c & 1) ? pk_hi(Aa[c >> 1]) : pk_lo(Aa[c >> 1]);
            const v2s h = pk_max(a, pk_max(vEQ, pk_swap(vEQ)));
            hcol[c] = h;
            vEQ = pk_max(pk_add(h, GQ), pk_add(vEQ, EC));
        }
        v2s Hn[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) Hn[k] = __builtin_shufflevector(hcol[2 * k], hcol[2 * k + 1], 0, 3);
        last.h = pk_join(Hn[0], Hn[1], Hn[2], Hn[3]);
        const v2s d0 = Hn[0] - Fa[0], d1 = Hn[1] - Fa[1], d2 = Hn[2] - Fa[2], d3 = Hn[3] - Fa[3];      // deficits of this row
        const v2s q0 = Hn[0] - Oa[0], q1 = Hn[1] - Oa[1], q2 = Hn[2] - Oa[2], q3 = Hn[3] - Oa[3];
        last.h0 = H0; last.o0 = O0; last.f0 = F0;
        // low bytes of four int16 pairs -> one dword each (v_perm_b32: bytes 0 and 2 of either source)
        last.fo.x = __builtin_amdgcn_perm(pk_bits(d1), pk_bits(d0), 0x06040200u);
        last.fo.y = __builtin_amdgcn_perm(pk_bits(d3), pk_bits(d2), 0x06040200u);
        last.fo.z = __builtin_amdgcn_perm(pk_bits(q1), pk_bits(q0), 0x06040200u);
        last.fo.w = __builtin_amdgcn_perm(pk_bits(q3), pk_bits(q2), 0x06040200u);
        *(v8s *)(M.H + ro + j0) = last.h; *(v4u *)(M.F + ro + j0) = last.fo;
        if (RING) {
            char *const sl = ring_lane + (i % RROWS) * POA_RING_SLOT;
            *(lds_v8s *)sl = last.h;
            *(lds_v4u *)(sl + 1024) = last.fo;
            if (lane == 0) {
                v2u c0;
                c0.x = ((unsigned)F0 & 0xffffu) | ((unsigned)O0 << 16); c0.y = (unsigned)H0 << 16;
                *(lds_v2u *)(lds_ring + (i % RROWS) * POA_RING_SLOT + 2048) = c0;
            }
        }
        if (sink) {                                            // H(i, len)
            const int cl = (len - 1) % CPL;
            v2s hv = hcol[0];
#pragma unroll
            for (int c = 1; c < CPL; ++c) hv = c == cl ? hcol[c] : hv;
            const int hlast = (int)(short)__builtin_amdgcn_readlane((int)pk_bits(hv), (len - 1) / CPL);
            if (best < hlast) { best = hlast; max_i = i; max_j = len; }
        }
        reg0 = nreg0; reg1 = nreg1;
        ap2 = bp2; ainfo = binfo;
        bp0 = cp0; bp1 = cp1; bp2 = cp2; binfo = cinfo;
    }
}


// ---- the TEAM form of the pipelined DP (round 5): one window on NW wavefronts of a workgroup -------------------------------------
// A window that has a SIMD to itself - every window of BASELINE config 4's 8-GPU leg: 750 windows per GPU on 1 024 SIMDs - is a
// serial job: a lone wavefront issues one instruction per ~8 clocks whatever its kind (scripts/issue_cost.hip), 2 400 clocks per DP
// row, 77 ms per window, of which the DP is 60 % (profiles/r05a_poa_lone_windows.txt).  The rows of a partial-order graph are only
// partially ordered: a row needs its predecessor rows and nothing else, and the graph of thirty noisy reads is several letters wide
// (a row's first predecessor is the row before it in 22 % of the cases).  So the NW wavefronts of a workgroup take the rows of the
// topological order in turn (row i belongs to wavefront (i - 1) % NW) and run them as a DATAFLOW: a row starts when its predecessor
// rows are complete.  The row arithmetic is poa_dp_pipelined's, unchanged; what is new is the hand-over between the wavefronts:
//   * the last RR rows live in an LDS ring shared by the team (slot i % RR; one more slot holds row 0); a predecessor at most K rows
//     back is read from there, behind its owner's progress word prog[w] = last row wavefront w has completed (LDS: written after the
//     ring slot, and the LDS serves a wavefront's operations in order);
//   * a row further back comes from HBM with NO wait: row i starts only when every row <= i - BOUND is complete (BOUND = RR - K: that
//     is also what keeps a ring slot from being overwritten while a reader may still want it), and a wavefront completes a row only
//     after `s_waitcnt vmcnt(3)` behind that row's three stores, i.e. after the stores of its PREVIOUS row are acknowledged.  With
//     K >= BOUND + NW - 1 the owner of a far row p has completed row p + NW by then: the far row's stores are visible by construction
//     (all wavefronts of a workgroup share the CU's vector L1, which is write-through).
// A waiting wavefront sleeps (s_sleep) and polls one LDS word per owner; the smallest incomplete row never waits for anything, so the
// team always makes progress.

template <int NW, int RR, int K>
__device__ __attribute__((always_inline)) void poa_dp_team(const PoaGraph &g, const PoaMatrices &M, const PoaArgs &A, const uint8_t *seq, int len, int n,
                                                           char *lds_ring, lds_team *sy, int wave, int &max_i, int &max_j)
{
    static_assert(NW == 2 || NW == 4, "row owner = (i - 1) & (NW - 1)");
    static_assert((RR & (RR - 1)) == 0, "ring slot = i & (RR - 1)");
    constexpr int CPL = 8, NT = 64 * NW, BOUND = RR - K;
    static_assert(BOUND >= NW && K >= BOUND + NW - 1, "far rows must be acknowledged by construction");
    const int lane = threadIdx.x & 63, tid = threadIdx.x;
    const PoaScore S = A.S;
    const int Wp = M.Wp;
    const Mat2 *Tc = A.Tc[0];
    PkMat T0 = pk_mat(Tc[0]), T1 = pk_mat(Tc[1]), T2 = pk_mat(Tc[2]), T3 = pk_mat(Tc[3]);
    // the scan's uniform factors in VECTOR registers: as scalars they were spilled and reloaded one v_readlane per use (eight per
    // row, each an issue slot of a wavefront that has the SIMD to itself); the team's kernel has the vector registers to spare
    asm volatile("" : "+v"(T0.ac), "+v"(T0.bd), "+v"(T1.ac), "+v"(T1.bd), "+v"(T2.ac), "+v"(T2.bd), "+v"(T3.ac), "+v"(T3.bd));
    const PkMat P16 = pk_mat(mp_pow(Tc[0], (lane & 15) + 1));
    const PkMat P32 = pk_mat(mp_pow(Tc[0], (lane & 31) + 1));
    const PkMat PC = pk_mat(mp_pow(Tc[0], lane));
    const v2s G2 = pk2(S.g, S.g), E2 = pk2(S.e, S.e), Q2 = pk2(S.q, S.q), C2 = pk2(S.c, S.c);
    const v2s GQ = pk2(S.g, S.q), EC = pk2(S.e, S.c), NM = pk2(S.n - S.m, S.n - S.m), MM = pk2(S.m, S.m);
    const v2s NEG2 = pk2(PK_NEG, PK_NEG);
    const int j0 = lane * CPL + 1;
    const bool mine = j0 <= len;

    // ---- set-up by all NT threads: row 0 (`initialize`) in memory and in the ring's extra slot, the row descriptors
    for (int j = tid; j <= len; j += NT) {
        const int e0 = j == 0 ? 0 : S.g + (j - 1) * S.e, q0 = j == 0 ? 0 : S.q + (j - 1) * S.c;
        M.H[j + POA_COL0] = (poa_cell_t)(j == 0 ? 0 : max(q0, e0));
    }
    uint8_t *const FO = (uint8_t *)M.F;
    for (int j = tid; j < POA_PIPE_STRIDE * 2; j += NT) FO[j] = 255;
    if (tid == 0) { M.H[POA_COL0 + POA_C0_F] = 0; M.H[POA_COL0 + POA_C0_O] = 0; }
    int32_t *d_pred = g.score, *d_info = g.pred;
    const int32_t *d_pred1 = g.path_node, *d_pred2 = g.path_pos;
    int32_t *d_pred3 = g.stack;
    {
        PoaGraph &gm = const_cast<PoaGraph &>(g);
        for (int r = tid; r < n; r += NT) {
            poa_rowdesc_one(gm, r);
            const int node = g.r2n[r];
            d_pred3[r] = g.in_cnt[node] > 3 ? g.n2r[PG_IN_SRC(g, node, 3)] + 1 : 0;
        }
    }
    char *const ring_lane = lds_ring + lane * 16;
    if (wave == 0) {
        v8s h0v;
#pragma unroll
        for (int c = 0; c < CPL; ++c) { const int j = j0 + c; h0v[c] = (short)max(max(S.g + (j - 1) * S.e, S.q + (j - 1) * S.c), -32768); }
        char *const sl = ring_lane + RR * POA_RING_SLOT;
        *(lds_v8s *)sl = h0v;
        v4u ff; ff.x = ff.y = ff.z = ff.w = 0xffffffffu;
        *(lds_v4u *)(sl + 1024) = ff;
        if (lane == 0) { v2u c0; c0.x = 0; c0.y = 0; *(lds_v2u *)(lds_ring + RR * POA_RING_SLOT + 2048) = c0; }
        if (lane < NW) sy->prog[lane] = 0;
    }
    __syncthreads();

    v2s sqp[4];
    {
        int sq[CPL];
#pragma unroll
        for (int c = 0; c < CPL; ++c) { const int j = j0 + c; sq[c] = j <= len ? seq[j - 1] : 0x7fff; }
#pragma unroll
        for (int k = 0; k < 4; ++k) sqp[k] = pk2(sq[2 * k], sq[2 * k + 1]);
    }
    auto wait_vm = [] { __builtin_amdgcn_s_waitcnt(0x0F70); };
    auto fetch = [&](int prow, PoaPredIn &x) {
        const int64_t b = (int64_t)prow * Wp + POA_COL0;
        x.h = *(const v8s *)(M.H + b + j0);
        x.fo = *(const v4u *)(M.F + b + j0);
        const v2u c0 = *(const v2u *)(M.H + b + POA_C0_F);
        x.f0 = (int)(short)(c0.x & 0xffffu); x.o0 = (int)(short)(c0.x >> 16); x.h0 = (int)(short)(c0.y >> 16);
    };
    auto ring_fetch = [&](int slot, PoaPredIn &x) {
        char *const sl = ring_lane + slot * POA_RING_SLOT;
        x.h = *(const lds_v8s *)sl;
        x.fo = *(const lds_v4u *)(sl + 1024);
        const v2u c0 = *(const lds_v2u *)(lds_ring + slot * POA_RING_SLOT + 2048);
        x.f0 = (int)(short)(c0.x & 0xffffu); x.o0 = (int)(short)(c0.x >> 16); x.h0 = (int)(short)(c0.y >> 16);
    };
    // a predecessor row of row `computing` (complete: the caller has waited for it)
    auto fetch_pred = [&](int prow_v, int computing, PoaPredIn &x) {
        const int prow = __builtin_amdgcn_readfirstlane(prow_v);
        if (prow == 0) ring_fetch(RR, x);
        else if (computing - prow <= K) ring_fetch(prow & (RR - 1), x);
        else {
            fetch(prow, x);
            wait_vm(); asm volatile("" : "+v"(x.h), "+v"(x.fo), "+v"(x.h0), "+v"(x.o0), "+v"(x.f0));
        }
    };
    // the descriptors of 64 of this wavefront's rows in five registers: lane k = row qbase + k * NW
    int q_p0 = 0, q_p1 = 0, q_p2 = 0, q_p3 = 0, q_info = 0;
    auto refill = [&](int base) {
        const int rr = min(base + lane * NW, n - 1);
        q_p0 = d_pred[rr]; q_p1 = d_pred1[rr]; q_p2 = d_pred2[rr]; q_p3 = d_pred3[rr]; q_info = d_info[rr];
        wait_vm();
        asm volatile("" : "+v"(q_p0), "+v"(q_p1), "+v"(q_p2), "+v"(q_p3), "+v"(q_info));
    };
    auto pred_terms = [&](const PoaPredIn &x, const v2s (&sc)[4], v2s (&F)[4], v2s (&O)[4], v2s (&H)[4]) {
        const v2s hp[4] = {pk_pair<0>(x.h), pk_pair<1>(x.h), pk_pair<2>(x.h), pk_pair<3>(x.h)};
        const v2s dfp[4] = {pk_from(__builtin_amdgcn_perm(0u, x.fo.x, 0x0c010c00u)), pk_from(__builtin_amdgcn_perm(0u, x.fo.x, 0x0c030c02u)),
                            pk_from(__builtin_amdgcn_perm(0u, x.fo.y, 0x0c010c00u)), pk_from(__builtin_amdgcn_perm(0u, x.fo.y, 0x0c030c02u))};
        const v2s dqp[4] = {pk_from(__builtin_amdgcn_perm(0u, x.fo.z, 0x0c010c00u)), pk_from(__builtin_amdgcn_perm(0u, x.fo.z, 0x0c030c02u)),
                            pk_from(__builtin_amdgcn_perm(0u, x.fo.w, 0x0c010c00u)), pk_from(__builtin_amdgcn_perm(0u, x.fo.w, 0x0c030c02u))};
        unsigned left = (unsigned)__builtin_amdgcn_update_dpp(0, (int)pk_bits(hp[3]), 0x138, 0xf, 0xf, false);
        if (lane == 0) left = (unsigned)x.h0 << 16;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            F[k] = pk_add(hp[k], pk_max(G2, E2 - dfp[k]));
            O[k] = pk_add(hp[k], pk_max(Q2, C2 - dqp[k]));
            const unsigned prev = k ? pk_bits(hp[k - 1]) : left;
            const v2s hs = pk_from(__builtin_amdgcn_alignbit(pk_bits(hp[k]), prev, 16));
            H[k] = pk_add(hs, sc[k]);
        }
    };
    lds_i32 *const prog_mine = (lds_i32 *)&sy->prog[lane & (NW - 1)];
    // row x (1-based; x <= 0: nothing to wait for) is complete according to the snapshot pv (lane l = prog[l & (NW - 1)])
    auto done = [&](int pv, int x) -> bool { return __builtin_amdgcn_readlane(pv, (x - 1) & (NW - 1)) >= x; };      // (progress words are never negative)

    int best = POA_NEG_INF, best_i = -1;
    int t = 0;                                                     // this wavefront's row counter
    for (int r = wave; r < n; r += NW, ++t) {
        const int i = r + 1;
        if ((t & 63) == 0) refill(r);
        const int l = t & 63;
        const int sp0 = __builtin_amdgcn_readlane(q_p0, l), sp1 = __builtin_amdgcn_readlane(q_p1, l);
        const int p2 = __builtin_amdgcn_readlane(q_p2, l), p3 = __builtin_amdgcn_readlane(q_p3, l);
        const int info = __builtin_amdgcn_readlane(q_info, l);
        const int letter = info & 0xff, ic = (info >> 8) & 0xff;
        const bool sink = (info >> 16) & 1;
        // ---- wait: every row up to i - BOUND, and this row's predecessors.  Lane l (mod NW) works out the row wavefront l must have
        // completed - its latest row at or below i - BOUND, or a predecessor of this row that it owns - and compares it with that
        // wavefront's progress word: one LDS read and one ballot per poll, no scalar branch per row waited for.
        {
            const int own = lane & (NW - 1);
            auto need_of = [&](int x) { return ((x - 1) & (NW - 1)) == own ? x : 0; };
            const int X = i - BOUND;
            int need = X - ((X - 1 - own) & (NW - 1));
            need = max(need, max(max(need_of(ic >= 1 ? sp0 : 0), need_of(ic >= 2 ? sp1 : 0)), max(need_of(ic >= 3 ? p2 : 0), need_of(ic >= 4 ? p3 : 0))));
            GBX_GUARD(gd_wait, 1 << 24);                           // polls: seconds, against microseconds of legitimate waiting
            while (__ballot(*(volatile lds_i32 *)prog_mine < need) != 0) {
                if (GBX_GUARD_TRIP(gd_wait, GBX_GK_POA, 4, i)) break;
                __builtin_amdgcn_s_sleep(1);
            }
            asm volatile("" ::: "memory");                       // nothing of the rows waited for is read before this point
        }
        PoaPredIn in0, in1;
        fetch_pred(ic >= 1 ? sp0 : 0, i, in0);
        if (ic >= 2) fetch_pred(sp1, i, in1);
        const int64_t ro = (int64_t)i * Wp + POA_COL0;
        v2s sc[4];
        {
            const v2s LL = pk2(letter, letter);
#pragma unroll
            for (int k = 0; k < 4; ++k) sc[k] = __builtin_elementwise_min(sqp[k] ^ LL, pk2(1, 1)) * NM + MM;
        }
        v2s Fa[4], Oa[4], Ha[4];
        int po = ic == 0 ? S.q - S.c : POA_NEG_INF, pf = ic == 0 ? S.g - S.e : POA_NEG_INF;
        if (ic > 0) { po = max(po, in0.o0); pf = max(pf, in0.f0); }
        pred_terms(in0, sc, Fa, Oa, Ha);
        if (ic > 1) {
            po = max(po, in1.o0); pf = max(pf, in1.f0);
            v2s F2[4], O2[4], H2[4];
            pred_terms(in1, sc, F2, O2, H2);
#pragma unroll
            for (int k = 0; k < 4; ++k) { Fa[k] = pk_max(Fa[k], F2[k]); Oa[k] = pk_max(Oa[k], O2[k]); Ha[k] = pk_max(Ha[k], H2[k]); }
        }
        if (ic > 2) {
            int node = 0;
            if (ic > 4) { node = g.r2n[r]; wait_vm(); asm volatile("" : "+v"(node)); }
            for (int k = 2; k < ic; ++k) {
                int prow;
                if (k == 2) prow = p2;
                else if (k == 3) prow = p3;
                else {
                    prow = g.n2r[PG_IN_SRC(g, node, k)] + 1; wait_vm(); asm volatile("" : "+v"(prow));
                    prow = __builtin_amdgcn_readfirstlane(prow);
                    GBX_GUARD(gd_wait5, 1 << 24);
                    for (;;) {                                     // (a fifth predecessor: rare; its row may be one of the last few)
                        const int pv = *(volatile lds_i32 *)prog_mine;
                        if (done(pv, prow) || GBX_GUARD_TRIP(gd_wait5, GBX_GK_POA, 5, i)) break;
                        __builtin_amdgcn_s_sleep(1);
                    }
                    asm volatile("" ::: "memory");
                }
                PoaPredIn x;
                fetch_pred(prow, i, x);
                po = max(po, x.o0); pf = max(pf, x.f0);
                v2s F2[4], O2[4], H2[4];
                pred_terms(x, sc, F2, O2, H2);
#pragma unroll
                for (int q = 0; q < 4; ++q) { Fa[q] = pk_max(Fa[q], F2[q]); Oa[q] = pk_max(Oa[q], O2[q]); Ha[q] = pk_max(Ha[q], H2[q]); }
            }
        }
        const int O0 = po + S.c, F0 = pf + S.e, H0 = max(O0, F0);
        const v2s cEQ = pk2(H0 + S.g, H0 + S.q);
        v2s Aa[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) Aa[k] = mine ? pk_max(Ha[k], pk_max(Fa[k], Oa[k])) : NEG2;
        v2s bEQ = NEG2;
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            const v2s a = (c & 1) ? pk_hi(Aa[c >> 1]) : pk_lo(Aa[c >> 1]);
            const v2s h = pk_max(a, pk_max(bEQ, pk_swap(bEQ)));
            bEQ = pk_max(pk_add(h, GQ), pk_add(bEQ, EC));
        }
        v2s x = bEQ;
        x = pk_max(x, pk_apply(T0, pk_dpp<0x111>(NEG2, x)));
        x = pk_max(x, pk_apply(T1, pk_dpp<0x112>(NEG2, x)));
        x = pk_max(x, pk_apply(T2, pk_dpp<0x114>(NEG2, x)));
        x = pk_max(x, pk_apply(T3, pk_dpp<0x118>(NEG2, x)));
        x = pk_max(x, pk_apply(P16, pk_dpp<0x142, 0xa>(NEG2, x)));
        x = pk_max(x, pk_apply(P32, pk_dpp<0x143, 0xc>(NEG2, x)));
        v2s vEQ = pk_dpp<0x138>(NEG2, x);
        if (lane == 0) vEQ = NEG2;
        vEQ = pk_max(vEQ, pk_apply(PC, cEQ));
        v2s hcol[CPL];
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            const v2s a = (c & 1) ? pk_hi(Aa[c >> 1]) : pk_lo(Aa[c >> 1]);
            const v2s h = pk_max(a, pk_max(vEQ, pk_swap(vEQ)));
            hcol[c] = h;
            vEQ = pk_max(pk_add(h, GQ), pk_add(vEQ, EC));
        }
        v2s Hn[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) Hn[k] = __builtin_shufflevector(hcol[2 * k], hcol[2 * k + 1], 0, 3);
        const v8s hrow = pk_join(Hn[0], Hn[1], Hn[2], Hn[3]);
        const v2s d0 = Hn[0] - Fa[0], d1 = Hn[1] - Fa[1], d2 = Hn[2] - Fa[2], d3 = Hn[3] - Fa[3];
        const v2s q0 = Hn[0] - Oa[0], q1 = Hn[1] - Oa[1], q2 = Hn[2] - Oa[2], q3 = Hn[3] - Oa[3];
        v4u forow;
        forow.x = __builtin_amdgcn_perm(pk_bits(d1), pk_bits(d0), 0x06040200u);
        forow.y = __builtin_amdgcn_perm(pk_bits(d3), pk_bits(d2), 0x06040200u);
        forow.z = __builtin_amdgcn_perm(pk_bits(q1), pk_bits(q0), 0x06040200u);
        forow.w = __builtin_amdgcn_perm(pk_bits(q3), pk_bits(q2), 0x06040200u);
        v2u c0;
        c0.x = ((unsigned)F0 & 0xffffu) | ((unsigned)O0 << 16); c0.y = (unsigned)H0 << 16;
        // ring first (what the next rows are waiting for), then the three stores of the row and a wait for everything issued before
        // them - the previous row's stores: `complete` then also means "my row before this one is acknowledged"
        {
            char *const sl = ring_lane + (i & (RR - 1)) * POA_RING_SLOT;
            *(lds_v8s *)sl = hrow;
            *(lds_v4u *)(sl + 1024) = forow;
            if (lane == 0) *(lds_v2u *)(lds_ring + (i & (RR - 1)) * POA_RING_SLOT + 2048) = c0;
        }
        {
            poa_cell_t *const pH = M.H + ro + j0, *const pF = M.F + ro + j0, *const pC = M.H + ro + POA_C0_F;     // (column 0: every lane stores the same 8 bytes)
            asm volatile("global_store_dwordx4 %0, %3, off\n\t"
                         "global_store_dwordx4 %1, %4, off\n\t"
                         "global_store_dwordx2 %2, %5, off\n\t"
                         "s_waitcnt vmcnt(3) lgkmcnt(0)"
                         :: "v"(pH), "v"(pF), "v"(pC), "v"(hrow), "v"(forow), "v"(c0) : "memory");
        }
        if (lane == 0) *(volatile lds_i32 *)&sy->prog[wave] = i;
        if (sink) {
            const int cl = (len - 1) % CPL;
            v2s hv = hcol[0];
#pragma unroll
            for (int c = 1; c < CPL; ++c) hv = c == cl ? hcol[c] : hv;
            const int hlast = (int)(short)__builtin_amdgcn_readlane((int)pk_bits(hv), (len - 1) / CPL);
            if (best < hlast) { best = hlast; best_i = i; }
        }
    }
    if (lane == 0) { sy->best[wave] = best; sy->best_i[wave] = best_i; }
    // (the row stores were issued from an asm block: the compiler's wait-count bookkeeping has not seen them, so the barrier's own
    // release fence may have been relaxed - the traceback reads what the other wavefronts stored)
    wait_vm();
    __syncthreads();
    // NW: the best sink at the last column; ties go to the first row in topological order (the serial loop's strict `<`)
    int b = POA_NEG_INF, bi = -1;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        const int v = *(volatile lds_i32 *)&sy->best[w], vi = *(volatile lds_i32 *)&sy->best_i[w];
        if (vi != -1 && (bi == -1 || v > b || (v == b && vi < bi))) { b = v; bi = vi; }
    }
    max_i = bi; max_j = bi == -1 ? -1 : len;
}

#ifdef GBX_POA_PHASE_STATS
__device__ unsigned long long g_tb_fast, g_tb_slow, g_tb_fill;
#endif
// ---- traceback for the pipelined DP: E and Q are not stored ---------------------------------------
// The pipelined DP writes H, F, O only (3 of the 5 planes: -40 % of the stores).  The traceback needs E and Q
// of a row only where the path moves left (an insertion), a few per cent of its steps; there the row's E, Q
// are rebuilt from its H by two tilted prefix-max scans, E(i,j) = max_{k<j} H(i,k) + g + (j-1-k) e (same
// for Q with q, c) - the same integers the DP had, since H is final - one row load and ~60 instructions,
// cached per row.  Otherwise identical to poa_traceback (poa_graph.h), which serves the stored-E/Q path.
// LD (the team kernel): the row descriptors - first four predecessor rows, letter and in-degree, node - are read from a copy in LDS
// (ldesc: six arrays of n 16-bit entries; the team's row ring is free once the DP is done).  A step of the general path begins with
// its row's descriptor and the rows of its further predecessors (two dependent loads each, through the in-edge list): on a lone
// wavefront every one of those was a round trip to memory.
typedef __attribute__((address_space(3))) const unsigned short lds_cu16;
template <int LD = 0>        // 0: descriptors from memory; 1: six arrays in LDS (team); 2: first predecessor, info and node in LDS (one-wavefront kernel)
__device__ __attribute__((always_inline)) void poa_traceback_wave(PoaGraph &g, const PoaMatrices &M, const PoaScore &S, const uint8_t *seq, int len,
                                   int max_i, int max_j, lds_cu16 *ldesc = nullptr, int ldn = 0)
{
    g.n_path = 0;
    if (max_i == -1 && max_j == -1) return;
    constexpr int CPL = 8;
    const int lane = threadIdx.x & 63;
    const int Wp = M.Wp;
    const int32_t *rd_pred = g.score, *rd_info = g.pred;
    auto desc_p0 = [&](int r) -> int { return LD ? (int)ldesc[r] : rd_pred[r]; };
    auto d_info_of = [&](int r) -> int { return LD ? (int)ldesc[(LD == 1 ? 4 : 1) * ldn + r] : rd_info[r]; };      // (letter | in-degree << 8: the sink bit is not needed here)
    auto d_node_of = [&](int r) -> int { return LD ? (int)ldesc[(LD == 1 ? 5 : 2) * ldn + r] : g.r2n[r]; };
    // DP row of in-edge source p >= 1 of the node of rank r
    auto pred_row = [&](int r, int node, int p) -> int { return LD == 1 && p < 4 ? (int)ldesc[p * ldn + r] : g.n2r[PG_IN_SRC(g, node, p)] + 1; };
    const int j0 = lane * CPL + 1;
    (void)len;
    int eq_row = -1;
    int Ec[CPL], Qc[CPL];                                      // E(eq_row, j0+c), Q(eq_row, j0+c)
#pragma unroll
    for (int c = 0; c < CPL; ++c) { Ec[c] = 0; Qc[c] = 0; }
    auto fill_eq = [&](int i) {
        if (i == eq_row) return;
        eq_row = i;
        const int64_t b = (int64_t)i * Wp + POA_COL0;
        int h[CPL];
        load_cells<CPL>(M.H + b + j0, h);
        const int h0 = M.H[b];
        int left = __builtin_amdgcn_update_dpp(h0, h[CPL - 1], 0x138, 0xf, 0xf, false);   // H(i, j0-1)
        if (lane == 0) left = h0;
        // lane-local pass from the identity
        int le = SNEG, lq = SNEG, u = left;
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            le = max(u + S.g, le + S.e); lq = max(u + S.q, lq + S.c);
            Ec[c] = le; Qc[c] = lq;
            u = h[c];
        }
        // inclusive scan over the lanes with decay: x[l] = max_k<=l  last[k] + (l-k) * 8 * step
        int xe = le, xq = lq;
        const int de = CPL * S.e, dq = CPL * S.c;
        xe = max(xe, dpp_i<0x111>(SNEG, xe) + de);      xq = max(xq, dpp_i<0x111>(SNEG, xq) + dq);
        xe = max(xe, dpp_i<0x112>(SNEG, xe) + 2 * de);  xq = max(xq, dpp_i<0x112>(SNEG, xq) + 2 * dq);
        xe = max(xe, dpp_i<0x114>(SNEG, xe) + 4 * de);  xq = max(xq, dpp_i<0x114>(SNEG, xq) + 4 * dq);
        xe = max(xe, dpp_i<0x118>(SNEG, xe) + 8 * de);  xq = max(xq, dpp_i<0x118>(SNEG, xq) + 8 * dq);
        xe = max(xe, dpp_i<0x142, 0xa>(SNEG, xe) + ((lane & 15) + 1) * de);  xq = max(xq, dpp_i<0x142, 0xa>(SNEG, xq) + ((lane & 15) + 1) * dq);
        xe = max(xe, dpp_i<0x143, 0xc>(SNEG, xe) + ((lane & 31) + 1) * de);  xq = max(xq, dpp_i<0x143, 0xc>(SNEG, xq) + ((lane & 31) + 1) * dq);
        int ie = dpp_i<0x138>(SNEG, xe), iq = dpp_i<0x138>(SNEG, xq);          // E, Q of column j0-1
        if (lane == 0) { ie = SNEG; iq = SNEG; }
#pragma unroll
        for (int c = 0; c < CPL; ++c) { Ec[c] = max(Ec[c], ie + (c + 1) * S.e); Qc[c] = max(Qc[c], iq + (c + 1) * S.c); }
    };
    auto pick = [&](const int (&a)[CPL], int j) -> int {       // a = Ec or Qc of the cached row, column j >= 1
        const int idx = (j - 1) & 7;
        int v = a[0];
#pragma unroll
        for (int c = 1; c < CPL; ++c) v = c == idx ? a[c] : v;
        return __builtin_amdgcn_readlane(v, (j - 1) >> 3);
    };
    auto E_at = [&](int i, int j) -> int { if (j == 0) return POA_NEG_INF; fill_eq(i); return pick(Ec, j); };
    auto Q_at = [&](int i, int j) -> int { if (j == 0) return POA_NEG_INF; fill_eq(i); return pick(Qc, j); };

    int i = max_i, j = max_j, prev_i = 0, prev_j = 0, np = 0;
#define PG_AT(A, a, b) ((int)(A)[(int64_t)(a) * Wp + (b) + POA_COL0])
    // F and O of a cell from the deficit plane (PoaPredIn): column j >= 1 is byte (j-1)&7 of the 16-byte group of lane
    // (j-1)>>3, H-F in the group's first 8 bytes and H-O in the last 8; column 0 keeps both as int16 in the row's pad cells
    const uint8_t *const FO = (const uint8_t *)M.F;
    auto fo_at = [&](int i_, int j_, int hv, int which) -> int {
        if (j_ == 0) return (int)M.H[(int64_t)i_ * Wp + POA_COL0 + (which ? POA_C0_O : POA_C0_F)];
        const int64_t byte = ((int64_t)i_ * Wp + POA_COL0 + 1) * 2 + (int64_t)((j_ - 1) >> 3) * 16 + ((j_ - 1) & 7) + (which ? 8 : 0);
        return hv - (int)FO[byte];
    };
    int plo = -1, phi = -1;                      // positions come out in descending order
    // The path collects in registers (entry np in lane np & 63) and leaves 64 entries at a time: a store per step
    // would sit in front of every gather below (gfx9 counts loads and stores in one in-order counter, so the wait
    // for a gather is also a wait for the acknowledgement of every store issued before it).
    int pbn = 0, pbp = 0;
    auto path_flush = [&](int first, int count) {
        if (lane < count && first + lane < g.aln_path_cap) { g.path_node[first + lane] = pbn; g.path_pos[first + lane] = pbp; }
    };
#define PG_PUSH(nd, ps) do { const int ps_ = (ps), nd_ = (nd); const bool me_ = lane == (np & 63); \
                             pbn = me_ ? nd_ : pbn; pbp = me_ ? ps_ : pbp; \
                             if (ps_ != -1) { if (phi < 0) phi = ps_; plo = ps_; } ++np; \
                             if ((np & 63) == 0) path_flush(np - 64, 64); } while (0)
    // A step is two dependent memory round trips: the row descriptor (first predecessor's row, letter, in-degree,
    // node) and then the predecessor's cell.  Most steps go to the first predecessor, so that row's descriptor is
    // requested together with its cell, and the cell that decided the move is the next step's H(i,j): a step
    // that follows the first predecessor (or stays in the row) starts with everything but one cell in registers.
    int Hcur = 0, d_p0 = 0, d_info = 0, d_node = 0;
    bool h_known = false, d_known = false;
    // On top of that the lanes keep an 8x8 block of the H cells behind the position (rows bi-8..bi-1, columns
    // bj-8..bj-1, one cell per lane, with the row's descriptor and the column's letter): a run of diagonal moves
    // to first predecessors is served from registers, one gather per 4-5 steps instead of one round trip each.
    int bi = -(1 << 20), bj = -(1 << 20), bH = 0, bP0 = 0, bInfo = 0, bNode = 0, bS = 0;
    // LD == 1 (round 6): the block's rows also carry the rows of their SECOND and THIRD in-edge sources (the team's descriptor copy
    // in LDS has them), so a diagonal move to one of those - a read that follows a side branch of the graph: most of the steps the
    // block could not serve - is taken from the registers too when that row lies in the block (95 % of second sources are within
    // six rows).  The in-edge order of the reference's test is kept: source p is tried only after sources 0..p-1 failed,
    // and a source outside the block hands the step to the general path.
    int bP1 = 0, bP2 = 0, d_p1 = 0, d_p2 = 0;
#ifdef GBX_POA_PHASE_STATS
    unsigned long long tb_fast_ = 0, tb_slow_ = 0, tb_fill_ = 0;
#endif
    auto fill_block = [&](int oi, int oj) {
#ifdef GBX_POA_PHASE_STATS
        ++tb_fill_;
#endif
        bi = oi; bj = oj;
        const int row = max(oi - 1 - (lane >> 3), 0), col = max(oj - 1 - (lane & 7), 0);
        bH = PG_AT(M.H, row, col);
        const int dr = max(row - 1, 0);
        bP0 = desc_p0(dr); bInfo = d_info_of(dr); bNode = d_node_of(dr);
        if (LD == 1) { bP1 = (int)ldesc[ldn + dr]; bP2 = (int)ldesc[2 * ldn + dr]; }
        bS = seq[col];
    };
    while (!(i == 0 && j == 0)) {
        if (i != 0 && j != 0 && d_known && h_known) {
            const int p0f = d_p0;
            int rr = bi - 1 - p0f, cc = bj - j;
            if ((unsigned)rr >= 8u || (unsigned)cc >= 8u) { fill_block(i, j); rr = i - 1 - p0f; cc = 0; }
            if ((unsigned)rr < 8u) {
                const int L = rr * 8 + cc;
                const int hd = __builtin_amdgcn_readlane(bH, L);
                const int mc = (d_info & 0xff) == __builtin_amdgcn_readlane(bS, L) ? S.m : S.n;
                if (Hcur == hd + mc) {
#ifdef GBX_POA_PHASE_STATS
                    ++tb_fast_;
#endif
                    PG_PUSH(d_node, j - 1);
                    i = p0f; j = j - 1; Hcur = hd;
                    d_p0 = __builtin_amdgcn_readlane(bP0, L); d_info = __builtin_amdgcn_readlane(bInfo, L);
                    d_node = __builtin_amdgcn_readlane(bNode, L); d_known = p0f > 0;
                    if (LD == 1) { d_p1 = __builtin_amdgcn_readlane(bP1, L); d_p2 = __builtin_amdgcn_readlane(bP2, L); }
                    if (np > g.aln_path_cap) { g.err |= POA_ERR_NODES; break; }
                    continue;
                }
#ifndef GBX_POA_TB_BLOCK23
#define GBX_POA_TB_BLOCK23 1          // 0: the block serves first in-edge sources only (round 5), for A/B builds
#endif
                if (LD == 1 && GBX_POA_TB_BLOCK23) {
                    const int icf = (d_info >> 8) & 0xff;
                    int took = -1, Lt = 0, ht = 0;
#pragma unroll
                    for (int p = 1; p < 3; ++p) {
                        if (took >= 0 || p >= icf) break;
                        const int pp = p == 1 ? d_p1 : d_p2;
                        const int rp = bi - 1 - pp;
                        if ((unsigned)rp >= 8u) break;               // that source's row is not in the block: the general path goes on from here
                        const int Lp = rp * 8 + cc;
                        const int hp = __builtin_amdgcn_readlane(bH, Lp);
                        if (Hcur == hp + mc) { took = pp; Lt = Lp; ht = hp; }
                    }
                    if (took >= 0) {
#ifdef GBX_POA_PHASE_STATS
                        ++tb_fast_;
#endif
                        PG_PUSH(d_node, j - 1);
                        i = took; j = j - 1; Hcur = ht;
                        d_p0 = __builtin_amdgcn_readlane(bP0, Lt); d_info = __builtin_amdgcn_readlane(bInfo, Lt);
                        d_node = __builtin_amdgcn_readlane(bNode, Lt); d_known = took > 0;
                        d_p1 = __builtin_amdgcn_readlane(bP1, Lt); d_p2 = __builtin_amdgcn_readlane(bP2, Lt);
                        if (np > g.aln_path_cap) { g.err |= POA_ERR_NODES; break; }
                        continue;
                    }
                }
            }
        }
#ifdef GBX_POA_PHASE_STATS
        ++tb_slow_;
#endif
        const int Hij = h_known ? Hcur : PG_AT(M.H, i, j);
        bool found = false, ext_left = false, ext_up = false;
        int node = -1, ic = 0, p0 = 0;
        int n_p0 = 0, n_info = 0, n_node = 0, Hnext = 0, info = 0;
        int c_p1 = 0, c_p2 = 0, n_p1 = 0, n_p2 = 0;           // (LD == 1) rows of the second / third in-edge source: of this row, of the first source's row
        bool hn_known = false, dn_first = false;          // next step: H known / descriptor = the prefetched one
        if (i != 0) {
            if (d_known) { p0 = d_p0; info = d_info; node = d_node; c_p1 = d_p1; c_p2 = d_p2; }
            else {
                p0 = desc_p0(i - 1); info = d_info_of(i - 1); node = d_node_of(i - 1);
                if (LD == 1) { c_p1 = (int)ldesc[ldn + i - 1]; c_p2 = (int)ldesc[2 * ldn + i - 1]; }
            }
            ic = (info >> 8) & 0xff;
            const int pr = p0 > 0 ? p0 - 1 : 0;            // descriptor of the first predecessor's row, in flight with its cell
            n_p0 = desc_p0(pr); n_info = d_info_of(pr); n_node = d_node_of(pr);
            if (LD == 1) { n_p1 = (int)ldesc[ldn + pr]; n_p2 = (int)ldesc[2 * ldn + pr]; }
            if (j != 0) {
                const int mc = (info & 0xff) == seq[j - 1] ? S.m : S.n;
                int pfirst = 0;
#ifndef GBX_POA_TB_BATCH
#define GBX_POA_TB_BATCH 0            // measured (profiles/r05o_poa_team_tb_ab.txt): 56.8 against 55.3 ms for a lone window - most general steps end at the first predecessor; off
#endif
                if (LD == 1 && GBX_POA_TB_BATCH) {
                    // the diagonal cells of the first four predecessors in ONE round trip: their rows come out of LDS, so the loads
                    // do not depend on each other (through memory each row took two dependent loads first: the loop below)
                    const int icc = ic ? ic : 1;
                    int pis[4], hds[4];
#pragma unroll
                    for (int p = 0; p < 4; ++p) { pis[p] = p && p < icc ? pred_row(i - 1, node, p) : p0; hds[p] = PG_AT(M.H, pis[p], j - 1); }
#pragma unroll
                    for (int p = 0; p < 4; ++p)
                        if (!found && p < icc && Hij == hds[p] + mc) { prev_i = pis[p]; prev_j = j - 1; found = true; Hnext = hds[p]; hn_known = true; dn_first = p == 0 && p0 > 0; }
                    pfirst = 4;
                }
                for (int p = pfirst; p < (ic ? ic : 1) && !found; ++p) {
                    const int pi = p ? pred_row(i - 1, node, p) : p0;
                    const int hd = PG_AT(M.H, pi, j - 1);
                    if (Hij == hd + mc) { prev_i = pi; prev_j = j - 1; found = true; Hnext = hd; hn_known = true; dn_first = p == 0 && p0 > 0; }
                }
            }
            if (!found) {
                for (int p = 0; p < (ic ? ic : 1) && !found; ++p) {
                    const int pi = p ? pred_row(i - 1, node, p) : p0;
                    const int hv = PG_AT(M.H, pi, j), fv = fo_at(pi, j, hv, 0), ov = fo_at(pi, j, hv, 1);
                    const bool c1 = !S.linear && Hij == fv + S.e;
                    const bool c2 = !c1 && Hij == hv + S.g;
                    const bool c3 = !S.linear && !c1 && !c2 && Hij == ov + S.c;
                    const bool c4 = !c1 && !c2 && !c3 && Hij == hv + S.q;
                    ext_up = ext_up || c1 || c3;
                    if (c1 || c2 || c3 || c4) { prev_i = pi; prev_j = j; found = true; Hnext = hv; hn_known = true; dn_first = p == 0 && p0 > 0; }
                }
            }
        }
        bool same_row = false;
        if (!found && j != 0) {
            const int hv = PG_AT(M.H, i, j - 1);
            const int ev = S.linear ? 0 : E_at(i, j - 1), qv = S.linear ? 0 : Q_at(i, j - 1);      // (linear: no gap states to rebuild)
            const bool c1 = !S.linear && Hij == ev + S.e;
            const bool c2 = !c1 && Hij == hv + S.g;
            const bool c3 = !S.linear && !c1 && !c2 && Hij == qv + S.c;
            const bool c4 = !c1 && !c2 && !c3 && Hij == hv + S.q;
            ext_left = c1 || c3;
            if (c1 || c2 || c3 || c4) { prev_i = i; prev_j = j - 1; found = true; Hnext = hv; hn_known = true; same_row = i != 0; }
        }
        // state for the next step
        if (same_row) { d_p0 = p0; d_info = info; d_node = node; d_p1 = c_p1; d_p2 = c_p2; d_known = true; }
        else if (dn_first) { d_p0 = n_p0; d_info = n_info; d_node = n_node; d_p1 = n_p1; d_p2 = n_p2; d_known = true; }
        else d_known = false;
        Hcur = Hnext; h_known = hn_known && !ext_left && !ext_up;
        if (ext_left || ext_up) d_known = d_known && ext_left;      // the extension loops below move on; a left run stays in the row
        PG_PUSH(i == prev_i ? -1 : node, j == prev_j ? -1 : j - 1);
        i = prev_i; j = prev_j;
        if (ext_left) {
            for (;;) {
                PG_PUSH(-1, j - 1);
                --j;
                if (j <= 0) break;                          // (column 0 ends every horizontal run: E(i,0) = Q(i,0) = -infinity; the test below says so too on sound data)
                if (E_at(i, j) + S.e != E_at(i, j + 1) && Q_at(i, j) + S.c != Q_at(i, j + 1)) break;
            }
        } else if (ext_up) {
            GBX_GUARD(gd_up, g.n_nodes + 2);
            for (;;) {
                if (GBX_GUARD_TRIP(gd_up, GBX_GK_POA, 2, i)) { g.err |= POA_ERR_STACK; break; }
                bool stop = false;
                prev_i = 0;
                const int nd = d_node_of(i - 1);
                const int icu = (d_info_of(i - 1) >> 8) & 0xff;
                const int hij = PG_AT(M.H, i, j);
                const int fij = fo_at(i, j, hij, 0), oij = fo_at(i, j, hij, 1);
                for (int p = 0; p < icu; ++p) {
                    const int pi = p ? pred_row(i - 1, nd, p) : desc_p0(i - 1);
                    const int hv = PG_AT(M.H, pi, j);
                    const bool s1 = fij == hv + S.g;
                    const bool s2 = !s1 && fij == fo_at(pi, j, hv, 0) + S.e;
                    const bool s3 = !s1 && !s2 && oij == hv + S.q;
                    const bool s4 = !s1 && !s2 && !s3 && oij == fo_at(pi, j, hv, 1) + S.c;
                    if (s1) stop = true; else if (s2) stop = false; else stop = s3;
                    if (s1 || s2 || s3 || s4) { prev_i = pi; break; }
                }
                PG_PUSH(nd, -1);
                i = prev_i;
                if (stop || i == 0) break;
            }
        }
        if (np > g.aln_path_cap) { g.err |= POA_ERR_NODES; break; }
    }
#undef PG_AT
#undef PG_PUSH
#ifdef GBX_POA_PHASE_STATS
    if (lane == 0) { atomicAdd(&g_tb_fast, tb_fast_); atomicAdd(&g_tb_slow, tb_slow_); atomicAdd(&g_tb_fill, tb_fill_); }
#endif
    if (np & 63) path_flush(np & ~63, np & 63);
    g.n_path = np <= g.aln_path_cap ? np : 0;
    g.path_lo = plo; g.path_hi = phi;
}

// ---- Graph::add_alignment, wavefront version --------------------------------------------------
// Same result as poa_add_alignment (poa_graph.h), reorganised so that the serial part touches memory
// as little as possible: letter codes are assigned with ballots in order of first appearance, fresh
// chains are written by all lanes in parallel, and the path is consumed in chunks of 64 elements whose
// node codes / sequence codes are fetched by the lanes in parallel and handed to the (wave-uniform)
// serial loop with v_readlane.
__device__ inline int rl(int v, int k) { return __builtin_amdgcn_readlane(v, k); }
// runtime-indexed vector elements would live in scratch memory: select explicitly
__device__ inline int pick4(const PoaInt4 &v, int k) { return k == 0 ? v.v[0] : k == 1 ? v.v[1] : k == 2 ? v.v[2] : v.v[3]; }

// a fresh chain for seq[begin,end) written lane-parallel; returns its first node or -1
__device__ int poa_add_chain_wave(PoaGraph &g, const uint8_t *seq, int begin, int end)
{
    if (begin >= end) return -1;
    const int lane = threadIdx.x & 63;
    const int L = end - begin, n0 = g.n_nodes;
    if (n0 + L > g.ncap) { g.err |= POA_ERR_NODES; return g.ncap - 1; }
    for (int k = lane; k < L; k += 64) {
        const int id = n0 + k;
        g.code[id] = (uint8_t)g.coder[seq[begin + k]];
        g.in_cnt[id] = k > 0; g.out_cnt[id] = k < L - 1; g.aln_cnt[id] = 0;
        if (k > 0) { PG_IN_SRC(g, id, 0) = id - 1; PG_IN_WT(g, id, 0) = 2; }
        if (k < L - 1) { PG_OUT_DST(g, id, 0) = id + 1; PG_OUT_SLOT(g, id, 0) = 0; }
    }
    g.n_nodes = n0 + L;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    return n0;
}

// Graph::add_edge with the first four out-edges fetched in one load.  chg = the state bytes of the topological sort in
// LDS (null when the sort runs in global memory): a node that gets a NEW in-edge is flagged there (POA_ST_CHANGED)
typedef __attribute__((address_space(3))) unsigned char lds_u8;
constexpr int POA_ST_ROOT = 8, POA_ST_CHANGED = 16, POA_ST_MATE = 64;
__device__ void poa_add_edge_wave(PoaGraph &g, int b, int e, int w, lds_u8 *chg, bool use_chg)
{
    const int oc = g.out_cnt[b];
    const PoaInt4 d4 = *(const PoaInt4 *)(g.out_dst + (int64_t)b * 4);
    const int ic = g.in_cnt[e];
    for (int k = 0; k < oc; ++k) {
        const int d = k < 4 ? pick4(d4, k) : PG_OUT_DST(g, b, k);
        if (d == e) { PG_IN_WT(g, e, PG_OUT_SLOT(g, b, k)) += w; return; }
    }
    if (oc >= g.deg || ic >= g.deg) { g.err |= POA_ERR_DEGREE; return; }
    PG_OUT_DST(g, b, oc) = e; PG_OUT_SLOT(g, b, oc) = (uint8_t)ic; g.out_cnt[b] = (uint8_t)(oc + 1);
    PG_IN_SRC(g, e, ic) = b; PG_IN_WT(g, e, ic) = w; g.in_cnt[e] = (uint8_t)(ic + 1);
    if (use_chg && (threadIdx.x & 63) == 0) chg[e] = (unsigned char)(chg[e] | POA_ST_CHANGED);
}

// Graph::topological_sort with all of its mutable state in LDS: per node one byte (mark in bits 0-1,
// "check aligned nodes" in bit 2), the DFS stack and the order being built (16-bit node ids).  The loop
// issues no global stores (on gfx9 a load behind a store waits for the store's acknowledgement); the
// order is written to r2n / n2r by all lanes at the end.  Same order as poa_topo_sort (poa_graph.h).
constexpr int POA_LDS_STACK16 = 256;       // largest DFS stack kept in LDS (sweep: 64 entries 436 ms, 128: 356, 192-320: 331-349,
                                           // 376: 362); the launch may choose less (PoaTopoLds::stk_cap)

#ifdef GBX_POA_PHASE_STATS
__device__ unsigned long long g_topo_cycles, g_topo_iters, g_topo_visits, g_topo_blocks, g_topo_dfs_cycles, g_topo_roots, g_topo_trivial;
__device__ unsigned long long g_add_serial_cycles, g_add_unsettled, g_add_lanepar_cycles, g_add_head_cycles;
#endif
// LDS arrays of the topological sort, persistent per wavefront for the life of a window
constexpr int POA_REC_SHORTS = 13;          // record of a node in the block cache: 4 in-edge sources, 8 aligned slots, counts
constexpr int POA_OBUF_SHORTS = 80;         // staging of the order under construction: 64 entries + one emission (1 + 8 aligned)
constexpr int POA_LDS_FIXED = POA_LDS_STACK16 * 2 + 64 * POA_REC_SHORTS * 2 + POA_OBUF_SHORTS * 2;   // stack, block cache, staging (16-byte multiple)
static_assert(POA_LDS_FIXED % 16 == 0, "per-node LDS arrays start 16-byte aligned");
// The pointers carry the LDS address space in their type.  As plain (generic) pointers they depend on the compiler
// inferring the address space through the whole inlined window kernel; when it does not (it stopped after an unrelated
// change of the DP: SQ_INSTS_LDS fell from 2.1e9 to 2.8e6 per launch) every access below becomes a FLAT instruction,
// which works, and costs half as much again per visit.
typedef __attribute__((address_space(3))) short lds_s16;
struct PoaTopoLds {
    lds_u8 *st8;             // [ncp] per node: mark in bits 0-1, "check aligned nodes" in bit 2
    lds_s16 *old;            // [ncp] rank of the node in the previous sort (-1: node added since)
    lds_s16 *stk;            // [stk_cap] DFS stack
    int stk_cap;             // entries; a deeper walk falls back to the global-memory sort
    lds_s16 *rec;            // [64][POA_REC_SHORTS] records of 64 nodes that were consecutive in the previous order
    lds_s16 *obuf;           // [POA_OBUF_SHORTS] the newest entries of the order, flushed to global memory 64 at a time
    int n_sorted;            // nodes ranked by the previous sort
    int flags_ok;            // the root flags in st8 describe the previous order (not after a sort that ran in global memory)
    int cap;                 // nodes st8 / old hold (PoaArgs::lds_ncap)
    int use;                 // 0: the node capacity does not fit LDS, the global-memory sort runs instead (a null test
                             // will not do: LDS offset 0 is a valid address, and the LDS null pointer is not 0)
};

// Graph::topological_sort, same order as poa_topo_sort (poa_graph.h).  The walk itself is serial, but one
// visit is done by the lanes together: lane k < 4 owns in-edge source k, lane 4+k aligned node k; they read
// the marks of their nodes at once, a ballot says who must be pushed (in list order = lane order), the
// pushes / the emission of the node with its aligned nodes are single LDS instructions.  A visit is four
// dependent LDS round trips instead of one per list entry.  The walk follows the previous topological
// order closely (one sequence changes the graph little), so the records of 64 nodes that were consecutive
// in that order are cached in LDS and re-read when the walk has moved on (second miss in the same
// 64-rank region); nodes added since the previous sort are read in place.
// INCREMENTAL (round 3).  The order spoa's DFS produces is a concatenation of BLOCKS: what the walk that starts at one root
// (a node not yet done when its id comes up) emits - the nodes it pulls in, then the root, then the root's aligned nodes
// (a root is the first of its group to be examined, so it emits the group).  One more sequence changes few of them:
//   * a node is examined through its own in-edge and aligned lists and the done / not-done state of the nodes in them;
//   * new edges only run forward in the previous order (the alignment path visits the nodes by increasing rank, new nodes
//     sit between them), so every node a walk pulls in - through old or new edges or aligned links - lies, in the previous
//     order, in an EARLIER block (done already) or in the SAME block.  A walk never reaches into a later block, the roots
//     stay roots, and their order (by id) is the blocks' order; new nodes have the largest ids, so those of them that no
//     old node pulls in are the last roots.
// Hence a block none of whose nodes got a new in-edge or aligned node (add_alignment flags those, POA_ST_CHANGED) comes out
// exactly as before and is COPIED (lane-parallel, marked done); the walk is repeated only from the roots of the blocks that
// hold a flagged node, with everything before the block done and everything behind it untouched: 66 of the 478 blocks of
// a 30-sequence window per sequence.  Checked against the full sort on the host (tests/hostcheck/poa_topo_inc_check.cpp,
// tests/test_poa_cpu.py: 0 differences in 23 000 sorts) and on the device: -DGBX_POA_TOPO_CHECK runs the full
// global-memory sort behind every incremental one and counts the orders that differ (scripts/dbg_poa_phases.py).
// State bytes (LDS, persistent for a window): mark (bits 0-1), check (2), POA_ST_ROOT (the node started a walk),
// POA_ST_MATE (it was emitted as an aligned node of a root: the tail of that root's block), POA_ST_CHANGED.
#ifdef GBX_POA_TOPO_CHECK
__device__ unsigned long long g_topo_mismatch, g_topo_inc_sorts, g_topo_walked, g_topo_blocks_all;
#endif
// INC = false: every sort walks everything (the second launch's kernel, whose column-block DP leaves no registers for
// the block bookkeeping: with it the 25 long windows of 'large' took 292 ms instead of 252 and ended after the main launch)
template <bool INC>
__device__ __attribute__((always_inline)) inline void poa_topo_sort_lds(PoaGraph &g, PoaTopoLds &T)
{
    lds_u8 *st8 = T.st8; lds_s16 *stk = T.stk, *old = T.old, *rec = T.rec, *obuf = T.obuf;
    // The order under construction goes to global memory (the global DFS-stack area is free here; r2n still holds the
    // previous order) - 64 entries at a time through an LDS staging buffer.  One store per emitted node looked free
    // ("stores only, nothing waits for them") and was not: the compiler keeps an `s_waitcnt vmcnt(0)` at the head of
    // the loop over the roots (a rarely taken path of the body loads from global memory into a register the head
    // redefines), gfx9 counts stores in that counter, and nearly every node of a partial-order graph is a root whose
    // walk ends with an emission - so every node waited a full store round trip (2 900 clocks per node).
    int32_t *ord = g.stack;
    int nflush = 0;                                            // entries of the order already in global memory
    const int n = g.n_nodes;
    const int lane = threadIdx.x & 63;
    const int n_old = T.n_sorted;
    const bool inc = INC && T.flags_ok && n_old > 0;
    // mark 0, check 1; old nodes keep their root flag and what add_alignment flagged
    for (int i = lane; i < n; i += 64) {
        const int keep = inc && i < n_old ? st8[i] & (POA_ST_ROOT | POA_ST_MATE | POA_ST_CHANGED) : 0;
        st8[i] = (unsigned char)(4 | keep);
        if (i >= n_old) old[i] = -1;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    int rb = -64, miss_rb = -64;
    auto load_block = [&](int base) {
        rb = base;
        const int id = g.r2n[min(base + lane, n_old - 1)];
        const int cc = (int)g.in_cnt[id] | (int)g.aln_cnt[id] << 8;
        const PoaInt4 e = *(const PoaInt4 *)(g.in_src + (int64_t)id * 4);
        const PoaInt4 a = *(const PoaInt4 *)(g.aln + (int64_t)id * POA_ALN_STRIDE);
        const PoaInt4 b = *(const PoaInt4 *)(g.aln + (int64_t)id * POA_ALN_STRIDE + 4);
        lds_s16 *r = rec + lane * POA_REC_SHORTS;
#pragma unroll
        for (int k = 0; k < 4; ++k) { r[k] = (short)e.v[k]; r[4 + k] = (short)a.v[k]; r[8 + k] = (short)b.v[k]; }
        r[12] = (short)cc;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    int sp = 0, nr = 0;
#ifdef GBX_POA_PHASE_STATS
    unsigned long long tv0_ = __builtin_readcyclecounter(), nvis_ = 0, nblk_ = 0, nroot_ = 0, ntriv_ = 0;
#endif
    // The node on top of the stack and its state byte are carried in registers whenever they are known: after a
    // push the new top is the last candidate pushed and its state was just read for the push decision, and the
    // outer loop has read the state of the node it pushes.  Only a pop has to read them back.  The walk is
    // bound by instruction issue (every wavefront of the CU is in some serial phase), so the loop is kept
    // short: one overflow check per visit with a single exit, 32-bit ballots (lanes 0..11 hold the list),
    // v_mbcnt for a lane's push slot.
    bool overflow = false;
    GBX_GUARD(gd_walk, (long long)n * (2 * (g.deg + POA_ALN_STRIDE) + 4) + 64);      // visits of one sort: a node is pushed once per list that names it
    // the walk from root i (its state byte st_i: not done), emitting into obuf / ord
    auto walk = [&](int i, int st_i) {
        if (lane == 0) st8[i] = (unsigned char)(st_i | POA_ST_ROOT);
        st_i |= POA_ST_ROOT;
        stk[sp++] = (short)i;
        int top = i, top_st = st_i;
        bool top_known = true;
#ifdef GBX_POA_PHASE_STATS
        ++nroot_; const unsigned long long v0_ = nvis_;
#endif
        while (sp) {
#ifdef GBX_POA_PHASE_STATS
            ++nvis_;
#endif
            if (GBX_GUARD_TRIP(gd_walk, GBX_GK_POA, 3, i)) { g.err |= POA_ERR_STACK; sp = 0; break; }
            if (sp > T.stk_cap - 16) { overflow = true; break; }     // a visit pushes at most 4 + 8 (+ cold, checked there)
            const int id = top_known ? top : (int)stk[sp - 1];
            const int stv = top_known ? top_st : (int)st8[id];
            top_known = false;
            if ((stv & 3) == 2) { --sp; continue; }
            const int o = old[id];
            if (o >= 0 && (unsigned)(o - rb) >= 64u) {          // ranked before, outside the block
                const int mb = o & ~63;
                if (mb == miss_rb) {
                    load_block(mb);
#ifdef GBX_POA_PHASE_STATS
                    ++nblk_;
#endif
                } else miss_rb = mb;
            }
            // this lane's list entry and the counts
            int cand, cc;
            const int l = o - rb;
            if (o >= 0 && (unsigned)l < 64u && !(stv & POA_ST_CHANGED)) {       // (a flagged node's cached record may be stale)
                cand = rec[l * POA_REC_SHORTS + min(lane, 11)];
                cc = (unsigned short)rec[l * POA_REC_SHORTS + 12];
            } else {
                cand = lane < 4 ? g.in_src[(int64_t)id * 4 + lane] : g.aln[(int64_t)id * POA_ALN_STRIDE + min(lane, 11) - 4];
                cc = (int)g.in_cnt[id] | (int)g.aln_cnt[id] << 8;
            }
            const int ic = cc & 0xff, ac = cc >> 8;
            const bool chk = (stv & 4) != 0;
            const bool is_in = lane < min(ic, 4), is_al = chk && lane >= 4 && lane < 4 + ac;
            const int sa = (is_in || is_al) ? st8[cand] : 2;
            const bool need = (sa & 3) != 2;
            bool valid = true;
            // pushes in list order: hot in-edge sources, cold ones (rare, serial), aligned nodes
            const uint32_t pin = (uint32_t)__ballot(is_in && need);
            if (pin) {
                if (is_in && need) stk[sp + (int)__builtin_amdgcn_mbcnt_lo(pin, 0u)] = (short)cand;
                sp += __builtin_popcount(pin); valid = false;
                const int hi = 31 - __builtin_clz(pin);
                top = __builtin_amdgcn_readlane(cand, hi); top_st = __builtin_amdgcn_readlane(sa, hi);
                top_known = true;
            }
            for (int k = 4; k < ic; ++k) {
                const int b = PG_IN_SRC(g, id, k);
                if ((st8[b] & 3) != 2) {
                    if (sp >= T.stk_cap - 16) { overflow = true; break; }
                    stk[sp++] = (short)b; valid = false;
                    top_known = false;
                }
            }
            const uint32_t pal = (uint32_t)__ballot(is_al && need);
            if (pal) {
                if (is_al && need) { stk[sp + (int)__builtin_amdgcn_mbcnt_lo(pal, 0u)] = (short)cand; st8[cand] = (unsigned char)(sa & ~4); }
                sp += __builtin_popcount(pal); valid = false;
                const int hi = 31 - __builtin_clz(pal);
                top = __builtin_amdgcn_readlane(cand, hi); top_st = __builtin_amdgcn_readlane(sa, hi) & ~4;
                top_known = true;
            }
            if (valid) {
                if (lane == 0) st8[id] = (unsigned char)((stv & (4 | POA_ST_ROOT)) | 2);
                if (chk) {
                    const int ob = nr - nflush;                  // < 64
                    if (lane == 0) obuf[ob] = (short)id;
                    if (lane >= 4 && lane < 4 + ac) {
                        obuf[ob + 1 + lane - 4] = (short)cand;
                        // the aligned nodes of a root are the tail of its block
                        st8[cand] = (unsigned char)(((int)st8[cand] & ~POA_ST_MATE) | ((stv & POA_ST_ROOT) ? POA_ST_MATE : 0));
                    }
                    nr += 1 + ac;
                    if (nr - nflush >= 64) {
                        ord[nflush + lane] = ((const volatile lds_s16 *)obuf)[lane];
                        const int rem = nr - nflush - 64;        // <= 8 entries of the last emission
                        const short mv = ((const volatile lds_s16 *)obuf)[64 + min(lane, 15)];
                        if (lane < rem) obuf[lane] = mv;
                        nflush += 64;
                    }
                }
                --sp;
            } else if (lane == 0) st8[id] = (unsigned char)((stv & (4 | POA_ST_ROOT | POA_ST_CHANGED)) | 1);
        }
#ifdef GBX_POA_PHASE_STATS
        if (nvis_ - v0_ == 1) ++ntriv_;
#endif
    };
#ifdef GBX_POA_TOPO_CHECK
    unsigned long long walked_ = 0, blocks_ = 0;
#endif
    int first_new_root = 0;
    if (inc) {
        // The previous order in one pass, 64 ranks at a time: ballots of the root / tail / flagged bits, the blocks inside a
        // chunk delimited by bit operations (a block ends at a root or tail rank that no tail rank follows), blocks that
        // cross a chunk boundary carried along.  A block that ends untouched is left for a later copy; when a touched one
        // ends, everything before it that is still pending is copied and the walk restarts at its root.
        int q = 0;                                             // first rank not yet in the new order
        int bs = 0, root_rank = -1;                            // the open block: its first rank, its root's rank once seen
        bool chg = false, pend_end = false;                    // ... holds a flagged node; the previous chunk ended on a root / tail rank
        auto copy_upto = [&](int upto) {
            if (upto <= q) return;
            if (lane < nr - nflush) ord[nflush + lane] = ((const volatile lds_s16 *)obuf)[lane];      // what the walks have staged
            for (int t0 = q; t0 < upto; t0 += 64) {
                const int t = t0 + lane;
                if (t < upto) {
                    const int nd = g.r2n[t];
                    ord[nr + t - q] = nd;
                    st8[nd] = (unsigned char)((st8[nd] & (POA_ST_ROOT | POA_ST_MATE)) | 4 | 2);
                }
            }
            nr += upto - q; nflush = nr; q = upto;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        };
        auto close_block = [&](int e) {                        // the open block is [bs, e]
            if (chg && root_rank >= 0) {
                copy_upto(bs);
                const int root = g.r2n[root_rank];
                walk(root, (int)st8[root]);
#ifdef GBX_POA_TOPO_CHECK
                ++walked_;
#endif
                q = e + 1;
            }
            bs = e + 1; chg = false; root_rank = -1;
        };
        for (int t0 = 0; t0 < n_old && !overflow; t0 += 64) {
            const int t = t0 + lane;
            const int sv = t < n_old ? (int)st8[g.r2n[t]] : 0;
            const unsigned long long rm = __ballot((sv & POA_ST_ROOT) != 0), mm = __ballot((sv & POA_ST_MATE) != 0);
            const unsigned long long cm = __ballot((sv & POA_ST_CHANGED) != 0);
            if (pend_end && !(mm & 1)) close_block(t0 - 1);
            pend_end = false;
            unsigned long long em = (rm | mm) & ~(mm >> 1);
            if (em >> 63) { pend_end = true; em &= ~(1ull << 63); }    // whether rank t0 + 63 ends its block shows in the next chunk
            while (em && !overflow) {
                const int eb = __builtin_ctzll(em);
                em &= em - 1;
                const int lo = bs > t0 ? bs - t0 : 0;
                const unsigned long long seg = (eb == 63 ? ~0ull : (2ull << eb) - 1) & ~((1ull << lo) - 1);
                chg = chg || (cm & seg) != 0;
                if (rm & seg) root_rank = t0 + 63 - __builtin_clzll(rm & seg);
                close_block(t0 + eb);
            }
            const int lo = bs > t0 ? bs - t0 : 0;                       // what is left of the chunk belongs to the open block
            if (lo < 64) {
                const unsigned long long rest = ~((1ull << lo) - 1);
                chg = chg || (cm & rest) != 0;
                if (rm & rest) root_rank = t0 + 63 - __builtin_clzll(rm & rest);
            }
        }
        if (pend_end && !overflow) close_block(n_old - 1);
        if (!overflow) copy_upto(n_old);
        first_new_root = n_old;
    }
    // full walk (first sort of a window, or after a sort in global memory) / the roots among the new nodes
    for (int i = first_new_root; i < n && !overflow; ++i) {
        const int st_i = st8[i];
        if ((st_i & 3) != 0) continue;
        walk(i, st_i);
    }
    if (overflow) {
        // deeper than the LDS stack (a long fresh chain walked back node by node): redo this sort with the
        // global-memory version, same order, and refresh the ranks kept in LDS
        poa_topo_sort(g);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (g.err == 0) for (int r = lane; r < n; r += 64) old[g.r2n[r]] = (short)r;
        T.n_sorted = n; T.flags_ok = 0;                         // no root flags from that sort: the next one walks everything
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        return;
    }
#ifdef GBX_POA_PHASE_STATS
    if (lane == 0) { atomicAdd(&g_topo_dfs_cycles, __builtin_readcyclecounter() - tv0_); atomicAdd(&g_topo_visits, nvis_); atomicAdd(&g_topo_blocks, nblk_); atomicAdd(&g_topo_roots, nroot_); atomicAdd(&g_topo_trivial, ntriv_); }
#endif
    if (lane < nr - nflush) ord[nflush + lane] = ((const volatile lds_s16 *)obuf)[lane];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int r = lane; r < n; r += 64) { const int id = ord[r]; g.r2n[r] = id; g.n2r[id] = r; old[id] = (short)r; }
    T.n_sorted = n; T.flags_ok = 1;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#ifdef GBX_POA_TOPO_CHECK
    // the same sort once more, in full, in global memory: the two orders must agree
    {
        for (int r = lane; r < n; r += 64) g.score[r] = g.r2n[r];          // (cons_path is the DFS stack's memory; the row descriptors are free here)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        poa_topo_sort(g);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        int bad = 0;
        for (int r = lane; r < n; r += 64) bad += g.score[r] != g.r2n[r];
        const unsigned long long any = __ballot(bad != 0);
        for (int r = 0; r < n; r += 64) if (r + lane < n && (st8[r + lane] & POA_ST_ROOT)) ++blocks_;
        if (lane == 0) {
            if (any || nr != n) atomicAdd(&g_topo_mismatch, 1ull);
            if (inc) { atomicAdd(&g_topo_inc_sorts, 1ull); atomicAdd(&g_topo_walked, walked_); }
        }
        unsigned long long bsum = blocks_;
        for (int d = 32; d; d >>= 1) bsum += __shfl_xor(bsum, d);
        if (lane == 0 && inc) atomicAdd(&g_topo_blocks_all, bsum);
        for (int r = lane; r < n; r += 64) old[g.r2n[r]] = (short)r;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
#endif
}

#ifdef GBX_POA_PHASE_STATS
#define TOPO_TIMED(g) { unsigned long long t0_ = __builtin_readcyclecounter(); POA_TOPO(g); if ((threadIdx.x & 63) == 0) { atomicAdd(&g_topo_cycles, __builtin_readcyclecounter() - t0_); atomicAdd(&g_topo_iters, (unsigned long long)g.n_nodes); } }
#else
#define TOPO_TIMED(g) POA_TOPO(g);
#endif
// topological sort through LDS when the kernel was launched with the LDS layout, else the global-memory one
#define POA_TOPO(g) { if (T.use) poa_topo_sort_lds<INC>(g, T); else poa_topo_sort(g); }
template <bool INC>
__device__ __attribute__((always_inline)) void poa_add_alignment_wave(PoaGraph &g, const uint8_t *seq, int len, PoaTopoLds &T)
{
    if (len == 0) return;
    const int lane = threadIdx.x & 63;
    // (round 5) the sort's LDS arrays hold T.cap nodes - what lets twelve (or sixteen) windows share a CU, not the graph's capacity: a
    // window that may outgrow them with this sequence (every base a new node at worst) sorts in global memory from here on
    if (T.use && g.n_nodes + len > T.cap) { T.use = 0; T.flags_ok = 0; }
    // letter codes, in order of first appearance in the sequence
    for (int base = 0; base < len; base += 64) {
        const int i = base + lane;
        const int c = i < len ? seq[i] : -1;
        bool unk = c >= 0 && g.coder[c] < 0;
        unsigned long long m = __ballot(unk);
        while (m) {
            const int cf = rl(c, __builtin_ctzll(m));
            g.coder[cf] = (int16_t)g.n_codes; g.decoder[g.n_codes] = (uint8_t)cf; ++g.n_codes;
            unk = unk && c != cf;
            m = __ballot(unk);
        }
    }
#ifdef GBX_POA_PHASE_STATS
    unsigned long long ts_ = __builtin_readcyclecounter(), t_ser_ = 0, t_par_ = 0, n_uns_ = 0, t_head_ = 0;
#define ADD_LAP(x) { const unsigned long long tn_ = __builtin_readcyclecounter(); x += tn_ - ts_; ts_ = tn_; }
#else
#define ADD_LAP(x)
#endif
    const int np = g.n_path;
    if (np == 0) {
        poa_add_chain_wave(g, seq, 0, len);
        if (g.err == 0) TOPO_TIMED(g)
        return;
    }
    const int first_pos = g.path_lo, last_pos = g.path_hi;
    const int before = g.n_nodes;
    poa_add_chain_wave(g, seq, 0, first_pos);
    int head = before == g.n_nodes ? -1 : g.n_nodes - 1;
    const int tail = poa_add_chain_wave(g, seq, last_pos + 1, len);
    int prev_w = head == -1 ? 0 : 1;
    bool head_new = head != -1;                                // head was made during this alignment and has no out-edge yet (here: the end of the fresh chain)
    ADD_LAP(t_head_)
    // The path, 64 elements at a time, ALL of it by the lanes in parallel (a lane = a path element).  What makes that
    // possible: the nodes along a path are distinct, an element's edge appends to (or bumps a weight in) the out-list of
    // the element before and the in-list of its own node only, and the aligned-node lists an element extends are those
    // of its own column - no two elements touch the same list, so the order in which the serial Graph::add_alignment
    // walks them does not show in the result, except in the ids of new nodes, which are handed out in path order by a
    // prefix count.  A lane never reads back what another lane has just written: lists of nodes made here are known to
    // be empty.  (The loop over the elements a lane-parallel pre-pass could not settle - 37 % of them, 6 000 clocks each,
    // two dependent memory round trips - was 11.5 % of the kernel's wave clocks.)
    for (int t0 = np - 1; t0 >= 0; t0 -= 64) {                 // forward order = stored order reversed: lane 0 first
        const int t = t0 - lane;
        const int pos = t >= 0 ? g.path_pos[t] : -1;
        const int node = t >= 0 ? g.path_node[t] : -1;
        const bool valid = pos >= 0;
        const int scode = valid ? (int)g.coder[seq[pos]] : -1;
        const int ncode = node >= 0 ? (int)g.code[node] : -1;
        // ---- the element's node: the path's, an aligned one with the element's letter, or a new one
        int id = node;
        bool fresh = valid && node < 0;
        const bool mism = valid && node >= 0 && ncode != scode;
        int ac = 0;
        PoaInt4 m0, m1;
        m0.v[0] = m0.v[1] = m0.v[2] = m0.v[3] = m1.v[0] = m1.v[1] = m1.v[2] = m1.v[3] = 0;
        int mcnt[POA_ALN_STRIDE];                               // aln_cnt of the aligned nodes (their lists grow when a new node joins the column)
        if (mism) {
            ac = g.aln_cnt[node];
            m0 = *(const PoaInt4 *)(g.aln + (int64_t)node * POA_ALN_STRIDE);
            m1 = *(const PoaInt4 *)(g.aln + (int64_t)node * POA_ALN_STRIDE + 4);
            int mcode[POA_ALN_STRIDE];
#pragma unroll
            for (int z = 0; z < POA_ALN_STRIDE; ++z) {
                const int a = z < 4 ? m0.v[z] : m1.v[z - 4];
                mcode[z] = z < ac ? (int)g.code[a] : -1;
                mcnt[z] = z < ac ? (int)g.aln_cnt[a] : 0;
            }
            int found = -1;
#pragma unroll
            for (int z = POA_ALN_STRIDE - 1; z >= 0; --z) if (z < ac && mcode[z] == scode) found = z < 4 ? m0.v[z] : m1.v[z - 4];      // the first in list order
            if (found >= 0) id = found;
            else fresh = true;
        }
        const bool joins = mism && fresh;                       // a new node that takes a place in the path node's column
        // ---- ids of the new nodes, in path order
        const unsigned long long fm = __ballot(fresh);
        const int n_new = __builtin_popcountll(fm);
        if (g.n_nodes + n_new > g.ncap) { g.err |= POA_ERR_NODES; return; }
        if (__ballot(joins && ac + 1 > POA_ALN_CAP)) { g.err |= POA_ERR_LETTERS; return; }
        if (fresh) {
            id = g.n_nodes + __builtin_popcountll(fm & ((1ull << lane) - 1));
            g.code[id] = (uint8_t)scode;
            g.in_cnt[id] = 0; g.out_cnt[id] = 0; g.aln_cnt[id] = joins ? (uint8_t)(ac + 1) : (uint8_t)0;
        }
        g.n_nodes += n_new;
        if (joins) {
#pragma unroll
            for (int z = 0; z < POA_ALN_STRIDE; ++z) {
                if (z < ac) {
                    const int a = z < 4 ? m0.v[z] : m1.v[z - 4];
                    g.aln[(int64_t)id * POA_ALN_STRIDE + z] = a;
                    g.aln[(int64_t)a * POA_ALN_STRIDE + mcnt[z]] = id; g.aln_cnt[a] = (uint8_t)(mcnt[z] + 1);
                    if (T.use) T.st8[a] = (unsigned char)(T.st8[a] | POA_ST_CHANGED);
                }
            }
            g.aln[(int64_t)id * POA_ALN_STRIDE + ac] = node;
            g.aln[(int64_t)node * POA_ALN_STRIDE + ac] = id; g.aln_cnt[node] = (uint8_t)(ac + 1);
            if (T.use) T.st8[node] = (unsigned char)(T.st8[node] | POA_ST_CHANGED);
        }
#ifdef GBX_POA_PHASE_STATS
        n_uns_ += (unsigned long long)__builtin_popcountll(__ballot(valid));
#endif
        ADD_LAP(t_par_)
        // ---- the edge from the element before (the last element of the chunk before, or the fresh chain, for the first one)
        const unsigned long long vmask = __ballot(valid);
        const unsigned long long vb = vmask & ((1ull << lane) - 1);
        const int prev_l = vb ? 63 - __builtin_clzll(vb) : lane;
        int pid = __builtin_amdgcn_ds_bpermute(prev_l << 2, id);
        int pfresh = __builtin_amdgcn_ds_bpermute(prev_l << 2, (int)fresh);
        if (!vb) { pid = head; pfresh = (int)head_new; }
        bool deg_err = false;
        if (valid && pid != -1) {
            const int oc = pfresh ? 0 : (int)g.out_cnt[pid];
            const int ic = fresh ? 0 : (int)g.in_cnt[id];
            int kk = -1;
            if (oc > 0 && !fresh) {
                const PoaInt4 d4 = *(const PoaInt4 *)(g.out_dst + (int64_t)pid * 4);
#pragma unroll
                for (int z = 3; z >= 0; --z) if (z < oc && d4.v[z] == id) kk = z;
                for (int z = 4; kk < 0 && z < oc; ++z) if (g.out_dst_x[(int64_t)pid * (g.deg - 4) + (z - 4)] == id) kk = z;
            }
            if (kk >= 0) {
                const int sl = PG_OUT_SLOT(g, pid, kk);
                int32_t *wp = sl < 4 ? g.in_wt + (int64_t)id * 4 + sl : g.in_wt_x + (int64_t)id * (g.deg - 4) + (sl - 4);
                *wp += 2;
            } else if (oc >= g.deg || ic >= g.deg) {
                deg_err = true;
            } else {
                PG_OUT_DST(g, pid, oc) = id; PG_OUT_SLOT(g, pid, oc) = (uint8_t)ic; g.out_cnt[pid] = (uint8_t)(oc + 1);
                PG_IN_SRC(g, id, ic) = pid; PG_IN_WT(g, id, ic) = 2; g.in_cnt[id] = (uint8_t)(ic + 1);
                if (T.use) T.st8[id] = (unsigned char)(T.st8[id] | POA_ST_CHANGED);
            }
        }
        if (__ballot(deg_err)) { g.err |= POA_ERR_DEGREE; return; }
        if (vmask) {                                            // the chunk's last element is the next chunk's predecessor
            const int lv = 63 - __builtin_clzll(vmask);
            head = rl(id, lv);
            head_new = rl((int)fresh, lv) != 0;
            prev_w = 1;
        }
        ADD_LAP(t_ser_)
    }
    if (tail != -1) poa_add_edge_wave(g, head, tail, prev_w + 1, T.st8, T.use != 0);
#ifdef GBX_POA_PHASE_STATS
    if (lane == 0) { atomicAdd(&g_add_serial_cycles, t_ser_); atomicAdd(&g_add_unsettled, n_uns_); atomicAdd(&g_add_lanepar_cycles, t_par_); atomicAdd(&g_add_head_cycles, t_head_); }
#endif
    if (g.err) return;
    TOPO_TIMED(g)
}

// The main launch is held back until the long-window launch (queued beside it on a side stream) has its wavefronts on the
// chip.  A long window is a serial job of the whole kernel's length, so it must start at once; but its instance is compiled
// for fewer, larger wavefronts, and once the main grid has filled the SIMDs (three wavefronts of 168 VGPRs each) a larger
// wavefront finds no room until main wavefronts retire - tens of milliseconds later.  One wavefront polls a counter the
// long instance bumps on entry, for at most a couple of milliseconds (a launch that cannot get all its blocks resident
// must not hold the main one back for ever).
__global__ void poa_gate_kernel(const unsigned long long *started, unsigned target)
{
    for (int k = 0; k < 600; ++k) {
        const unsigned long long v = __hip_atomic_load(started, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v >= target) break;
        __builtin_amdgcn_s_sleep(127);
    }
}

constexpr int POA_SWEEPS = 8;
constexpr int POA_CNT_MAIN = 14, POA_CUR_MAIN = 15, POA_CNT_LONG = 16, POA_CUR_LONG = 17, POA_LONG_STARTED = 18, POA_NCOUNTERS = 32;
// Work lists (one block).  A window's cost grows with the square of its sequence count, so the main launch hands the
// windows out heaviest class first (class = sequences per window against the job's mean: the kernel's tail is then made of
// light windows); windows that hold a sequence of more than POA_PIPE_MAXLEN bases go to the second launch's list.
__global__ void __launch_bounds__(1024) poa_classify_kernel(PoaArgs A, int32_t *wlist, int32_t *llist, int llist_cap)
{
    __shared__ int cnt[POA_SWEEPS + 1], base[POA_SWEEPS + 1];
    const int tid = threadIdx.x;
    if (tid <= POA_SWEEPS) cnt[tid] = 0;
    __syncthreads();
    const long long nw = (long long)A.n_windows;
    const long long seq_sum = (long long)(A.win_first_seq[A.n_windows] - A.win_first_seq[0]);
    auto class_of = [&](int64_t w) {
        const int64_t s0 = A.win_first_seq[w], s1 = A.win_first_seq[w + 1];
        for (int64_t s = s0; s < s1; ++s) if (A.seq_len[s] > POA_PIPE_MAXLEN) return POA_SWEEPS;
        // class = floor((1.4 - sequences / mean) * 10) clamped to [0, POA_SWEEPS): 0 = 1.4 x the mean and more
        const long long ns10 = 10ll * (long long)(s1 - s0) * nw;
        const long long d = 14 * seq_sum - ns10;
        return d <= 0 ? 0 : (int)min((long long)(POA_SWEEPS - 1), d / max(seq_sum, 1ll));
    };
    for (int64_t w = tid; w < A.n_windows; w += 1024) atomicAdd(&cnt[class_of(w)], 1);
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int c = 0; c < POA_SWEEPS; ++c) { base[c] = run; run += cnt[c]; }
        base[POA_SWEEPS] = 0;
        A.cells[POA_CNT_MAIN] = (unsigned long long)run;
        A.cells[POA_CNT_LONG] = (unsigned long long)min(cnt[POA_SWEEPS], llist_cap);
    }
    __syncthreads();
    for (int64_t w = tid; w < A.n_windows; w += 1024) {
        const int c = class_of(w);
        const int at = atomicAdd(&base[c], 1);
        if (c < POA_SWEEPS) wlist[at] = (int32_t)w;
        else if (at < llist_cap) llist[at] = (int32_t)w;
        else if (A.status) A.status[w] = POA_ERR_NODES;       // (cannot happen with a plan made for these windows)
    }
}

__device__ __attribute__((always_inline)) inline void poa_bind_graph(PoaGraph &g, char *slot, const SlotLayout &L, const PoaArgs &A)
{
    g.ncap = A.ncap; g.deg = A.deg; g.stk_cap = L.stk_cap; g.aln_path_cap = L.path_cap;
    g.code = (uint8_t *)(slot + L.code); g.in_cnt = (uint8_t *)(slot + L.in_cnt);
    g.out_cnt = (uint8_t *)(slot + L.out_cnt); g.aln_cnt = (uint8_t *)(slot + L.aln_cnt);
    g.out_slot = (uint8_t *)(slot + L.out_slot); g.mark = (uint8_t *)(slot + L.mark); g.check = (uint8_t *)(slot + L.check);
    g.decoder = (uint8_t *)(slot + L.decoder); g.coder = (int16_t *)(slot + L.coder);
    g.in_src = (int32_t *)(slot + L.in_src); g.in_wt = (int32_t *)(slot + L.in_wt);
    g.in_src_x = (int32_t *)(slot + L.in_src_x); g.in_wt_x = (int32_t *)(slot + L.in_wt_x);
    g.out_dst_x = (int32_t *)(slot + L.out_dst_x); g.out_slot_x = (uint8_t *)(slot + L.out_slot_x);
    g.out_dst = (int32_t *)(slot + L.out_dst); g.aln = (int32_t *)(slot + L.aln);
    g.r2n = (int32_t *)(slot + L.r2n); g.n2r = (int32_t *)(slot + L.n2r);
    g.stack = (int32_t *)(slot + L.stack); g.score = (int32_t *)(slot + L.score); g.pred = (int32_t *)(slot + L.pred);
    g.cons_path = g.stack;                                    // the global DFS-stack area doubles as the consensus path
    g.path_node = (int32_t *)(slot + L.path_node); g.path_pos = (int32_t *)(slot + L.path_pos);
}
// serial DFS state on chip (LDS): state byte per node, order under construction, stack
__device__ __attribute__((always_inline)) inline void poa_bind_lds(PoaTopoLds &T, char *lds_raw, const PoaArgs &A)
{
    const int ncp = (A.lds_ncap + 15) & ~15;
    lds_u8 *const lds0 = (lds_u8 *)lds_raw;
    // fixed-size arrays first, at compile-time offsets (the per-node arrays behind them need the node capacity): the
    // bases then fold into the ds instructions' offset fields instead of living in (spilled) scalar registers
    T.stk = (lds_s16 *)lds0;
    T.rec = (lds_s16 *)(lds0 + POA_LDS_STACK16 * 2);
    T.obuf = T.rec + 64 * POA_REC_SHORTS;
    T.st8 = lds0 + POA_LDS_FIXED;
    T.use = A.lds_marks; T.cap = A.lds_ncap;
    T.old = (lds_s16 *)(lds0 + POA_LDS_FIXED + ncp);
    T.stk_cap = A.lds_stack;
    T.n_sorted = 0; T.flags_ok = 0;
}

// The phase functions above are always_inline: the window kernel is one function on purpose.  Left to its cost model the
// inliner stops at 1100 basic blocks (amdgpu-inline-max-bb) and the first phase it leaves out (add_alignment, after the
// DP grew by a dozen instructions) takes `PoaGraph &` by reference: the graph's pointers then live in scratch memory,
// lose their address space, and every access of the kernel becomes a FLAT instruction (993 of them, 397 instead of
// 330 ms, found through SQ_INSTS_LDS dropping to nothing).
// WAVES / RROWS: wavefronts per SIMD the instance is compiled for and rows of the DP's LDS ring (main launch only): <3, 6> is
// the default (168 VGPRs with spills, twelve windows per CU), <2, 9> the variant without spills (GBX_POA_OCC=2)
template <bool LONG, int WAVES = 3, int RROWS = POA_RING_DEFAULT>
__global__ void __launch_bounds__(64, WAVES) poa_kernel(PoaArgs A, SlotLayout L)
{
    char *slot = A.work + (int64_t)blockIdx.x * A.slot_bytes;
    PoaGraph g;
    poa_bind_graph(g, slot, L, A);
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    PoaTopoLds T;
    poa_bind_lds(T, lds_raw, A);
    poa_cell_t *mat = (poa_cell_t *)(slot + L.mat);
    if (LONG && (threadIdx.x & 63) == 0) atomicAdd(A.cells + POA_LONG_STARTED, 1ull);      // poa_gate_kernel waits for these

    unsigned long long cells = 0;
#ifdef GBX_POA_PHASE_STATS
    unsigned long long t_dp = 0, t_tb = 0, t_add = 0, t_cons = 0, n_rows = 0, n_steps = 0;
#define PH_T0 unsigned long long ph0_ = __builtin_readcyclecounter();
#define PH_ACC(x) { unsigned long long ph1_ = __builtin_readcyclecounter(); x += ph1_ - ph0_; ph0_ = ph1_; }
#else
#define PH_T0
#define PH_ACC(x)
#endif
    // Windows are handed out by an atomic cursor over the launch's work list (poa_classify_kernel): a slot that finishes early
    // takes the next one (a static stride left the slots with three windows running alone for a third of the kernel).
    const unsigned nwork = (unsigned)A.cells[A.cnt_idx];
    for (;;) {
        unsigned long long wq = 0;
        if ((threadIdx.x & 63) == 0) wq = atomicAdd(A.cells + A.cur_idx, 1ull);
        const unsigned q32 = (unsigned)__builtin_amdgcn_readfirstlane((int)wq);
        if (q32 >= nwork) break;
        const int64_t w = (int64_t)A.wlist[q32];
        poa_graph_reset(g);
        T.n_sorted = 0; T.flags_ok = 0; T.use = A.lds_marks;
        const int64_t s0 = A.win_first_seq[w], s1 = A.win_first_seq[w + 1];
        for (int64_t s = s0; s < s1; ++s) {
            const uint8_t *seq = A.arena + A.seq_off[s];
            const int len = A.seq_len[s];
            g.n_path = 0;
            bool ran_dp = false;
            if (g.n_nodes != 0 && len != 0 && g.err == 0) {
                ran_dp = true;
                const int wp = !LONG || len <= POA_PIPE_MAXLEN ? POA_PIPE_STRIDE : poa_row_stride(len);
                const int64_t plane = (int64_t)(g.n_nodes + 1) * wp;
                PoaMatrices M = {mat, mat + plane, mat + 2 * plane, mat + 3 * plane, mat + 4 * plane, wp};
                int mi, mj;
                cells += (unsigned long long)g.n_nodes * (unsigned long long)len;
                PH_T0
                // The serial phases (one useful lane) are latency chains that lose issue slots to the other wavefronts' DP rows;
                // the DP is throughput work that does not mind waiting.  Priority 3 for the former: 300.8 -> 294.7 ms.
                __builtin_amdgcn_s_setprio(0);
                if (!LONG || len <= POA_PIPE_MAXLEN) poa_dp_pipelined<(LONG && !GBX_POA_LONG_RING) ? 0 : RROWS>(g, M, A, seq, len, mi, mj, lds_raw);
                else poa_dp<8>(g, M, A, seq, len, mi, mj);     // longer sequences run as several column blocks
                PH_ACC(t_dp)
                __builtin_amdgcn_s_setprio(3);
#ifndef GBX_POA_TB_LDS3
#define GBX_POA_TB_LDS3 1           // (round 5, measured: profiles/r05z_poa_tb_lds3_ab.txt, 200.2 against 201.9 ms) the one-wavefront kernel's traceback with p0 / info / node of every row in the ring's LDS
#endif
                const bool ld3 = GBX_POA_TB_LDS3 && T.use && (!LONG || len <= POA_PIPE_MAXLEN) && g.n_nodes * 6 <= 3 * T.cap;      // (the sort's per-node arrays: 3 bytes per node, free until add_alignment restores them)
                if (ld3) {
                    typedef __attribute__((address_space(3))) unsigned short lds_u16;
                    lds_u16 *const ldw = (lds_u16 *)((lds_u8 *)lds_raw + POA_LDS_FIXED);
                    const int n_ = g.n_nodes;
                    for (int r = threadIdx.x & 63; r < n_; r += 64) { ldw[r] = (unsigned short)g.score[r]; ldw[n_ + r] = (unsigned short)g.pred[r]; ldw[2 * n_ + r] = (unsigned short)g.r2n[r]; }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    poa_traceback_wave<2>(g, M, A.S, seq, len, mi, mj, (lds_cu16 *)((lds_u8 *)lds_raw + POA_LDS_FIXED), n_);
                } else
                if (!LONG || len <= POA_PIPE_MAXLEN) poa_traceback_wave(g, M, A.S, seq, len, mi, mj);
                else poa_traceback(g, M, A.S, seq, mi, mj);
                PH_ACC(t_tb)
#ifdef GBX_POA_PHASE_STATS
                n_rows += (unsigned long long)g.n_nodes; n_steps += (unsigned long long)g.n_path;
#endif
            }
            {
                PH_T0
                if (T.use && ran_dp) {
                    // the DP's row ring has used the sort's LDS: state bytes back from the slot, previous ranks = n2r
                    const uint8_t *save = (const uint8_t *)(slot + L.st8save);
                    for (int i = threadIdx.x & 63; i < g.n_nodes; i += 64) { T.st8[i] = save[i]; T.old[i] = i < T.n_sorted ? (short)g.n2r[i] : (short)-1; }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                }
                if (g.err == 0) poa_add_alignment_wave<(!LONG || GBX_POA_LONG_INC)>(g, seq, len, T);
                if (T.use && s + 1 < s1) {
                    uint8_t *save = (uint8_t *)(slot + L.st8save);
                    for (int i = threadIdx.x & 63; i < g.n_nodes; i += 64) save[i] = T.st8[i];
                }
                PH_ACC(t_add)
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        int clen = 0;
        {
            PH_T0
            if (g.err == 0) clen = poa_consensus(g, A.cons + w * A.cons_stride, (int)A.cons_stride);
            PH_ACC(t_cons)
        }
        if ((threadIdx.x & 63) == 0) { A.cons_len[w] = clen; A.status[w] = g.err; }
    }
    if ((threadIdx.x & 63) == 0) atomicAdd(A.cells, cells);
#ifdef GBX_POA_PHASE_STATS
    if ((threadIdx.x & 63) == 0) { atomicAdd(A.cells + 1, t_dp); atomicAdd(A.cells + 2, t_tb); atomicAdd(A.cells + 3, t_add); atomicAdd(A.cells + 4, t_cons); atomicAdd(A.cells + 12, n_rows); atomicAdd(A.cells + 13, n_steps); }
    __syncthreads();
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) {
        A.cells[5] = g_topo_cycles; A.cells[6] = g_topo_iters; A.cells[7] = g_topo_visits; A.cells[8] = g_topo_blocks; A.cells[9] = g_topo_dfs_cycles; A.cells[10] = g_topo_roots; A.cells[11] = g_topo_trivial;
        A.cells[28] = g_tb_fast; A.cells[29] = g_tb_slow; A.cells[30] = g_tb_fill;
        A.cells[24] = g_add_serial_cycles; A.cells[25] = g_add_unsettled; A.cells[26] = g_add_lanepar_cycles; A.cells[27] = g_add_head_cycles;      // (14-17 are the work lists' counts and cursors)
#ifdef GBX_POA_TOPO_CHECK
        A.cells[20] = g_topo_mismatch; A.cells[21] = g_topo_inc_sorts; A.cells[22] = g_topo_walked; A.cells[23] = g_topo_blocks_all;
#endif
    }
#endif
}


// ---- the wide kernel (round 6): int32 cells ---------------------------------------------------------------------------------------
// spoa switches to 32-bit lanes when a window's scores may leave the int16 range (long reads: with c = -1 a graph of 25 000
// nodes and a 5 000-base read is already past -30 000 in the worst case the plan must assume).  Here such windows - and
// windows whose graph outgrew every capacity the int16 paths admit - run on this kernel: poa_kernel's window loop with the
// column-block DP and the five-plane traceback instantiated for int32 cells (PoaMatricesW), the topological sort in global
// memory (node ids need not fit 16 bits), one window per wavefront, slots handed out by a cursor.  A fallback: correct for
// every window spoa accepts and the device has memory for, not tuned (the fast paths serve everything that fits them).
__global__ void __launch_bounds__(64, 1) poa_wide_kernel(PoaArgs A, SlotLayout L)
{
    char *slot = A.work + (int64_t)blockIdx.x * A.slot_bytes;
    PoaGraph g;
    poa_bind_graph(g, slot, L, A);
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    PoaTopoLds T;
    poa_bind_lds(T, lds_raw, A);                               // (A.lds_marks = 0: T.use = 0, nothing of the sort lives in LDS)
    int32_t *mat = (int32_t *)(slot + L.mat);
    unsigned long long cells = 0;
    for (;;) {
        unsigned long long wq = 0;
        if ((threadIdx.x & 63) == 0) wq = atomicAdd(A.cells + A.cur_idx, 1ull);
        const unsigned q32 = (unsigned)__builtin_amdgcn_readfirstlane((int)wq);
        if ((int64_t)q32 >= A.n_windows) break;
        const int64_t w = (int64_t)q32;
        poa_graph_reset(g);
        T.n_sorted = 0; T.flags_ok = 0; T.use = 0;
        const int64_t s0 = A.win_first_seq[w], s1 = A.win_first_seq[w + 1];
        for (int64_t s = s0; s < s1; ++s) {
            const uint8_t *seq = A.arena + A.seq_off[s];
            const int len = A.seq_len[s];
            g.n_path = 0;
            if (g.n_nodes != 0 && len != 0 && g.err == 0) {
                const int wp = poa_row_stride(len);
                const int64_t plane = (int64_t)(g.n_nodes + 1) * wp;
                PoaMatricesW M = {mat, mat + plane, mat + 2 * plane, mat + 3 * plane, mat + 4 * plane, wp};
                int mi, mj;
                cells += (unsigned long long)g.n_nodes * (unsigned long long)len;
                poa_dp<8, 1, PoaMatricesW>(g, M, A, seq, len, mi, mj);
                poa_traceback(g, M, A.S, seq, mi, mj);
            }
            if (g.err == 0) poa_add_alignment_wave<false>(g, seq, len, T);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        int clen = 0;
        if (g.err == 0) clen = poa_consensus(g, A.cons + w * A.cons_stride, (int)A.cons_stride);
        if ((threadIdx.x & 63) == 0) { A.cons_len[w] = clen; A.status[w] = g.err; }
    }
    if ((threadIdx.x & 63) == 0) atomicAdd(A.cells, cells);
}

// ---- the team kernel (round 5): one window per WORKGROUP of NW wavefronts ---------------------------------------------------------
// poa_kernel's window loop with the DP of a sequence run by all NW wavefronts (poa_dp_team) and everything else - traceback,
// add_alignment with the sort, consensus - by wavefront 0 while the others wait at a barrier (a waiting wavefront takes no issue
// slots).  For jobs with fewer windows than the chip has SIMDs (a shard of BASELINE config 4: 750 windows on 1 024 SIMDs) and for the
// windows of the long launch, which are few by definition: there the job is as long as its slowest window, and a window on one
// wavefront is a serial program.  LONG: a sequence over 512 bases runs poa_dp<8> / poa_traceback on wavefront 0 (column blocks,
// five-plane slots).  TEAM_RR / TEAM_K: rows of the shared LDS ring and how far back a predecessor is read from it.
constexpr int POA_TEAM_NW = 4, POA_TEAM_RR = 16, POA_TEAM_K = 10;
constexpr int POA_TEAM_RING_BYTES = (POA_TEAM_RR + 1) * POA_RING_SLOT;

template <bool LONG, int WAVES>
__global__ void __launch_bounds__(64 * POA_TEAM_NW, WAVES) poa_team_kernel(PoaArgs A, SlotLayout L, int sync_off)
{
    constexpr int NW = POA_TEAM_NW;
    char *slot = A.work + (int64_t)blockIdx.x * A.slot_bytes;
    PoaGraph g;
    poa_bind_graph(g, slot, L, A);
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    PoaTopoLds T;
    poa_bind_lds(T, lds_raw, A);
    lds_team *const sy = (lds_team *)((lds_u8 *)lds_raw + sync_off);
    poa_cell_t *mat = (poa_cell_t *)(slot + L.mat);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    if (LONG && threadIdx.x == 0) atomicAdd(A.cells + POA_LONG_STARTED, 1ull);      // poa_gate_kernel waits for these

    unsigned long long cells = 0;
#ifdef GBX_POA_PHASE_STATS
    unsigned long long t_dp = 0, t_tb = 0, t_add = 0, t_cons = 0, n_rows = 0, n_steps = 0;
#endif
    const unsigned nwork = (unsigned)A.cells[A.cnt_idx];
    for (;;) {
        if (threadIdx.x == 0) {
            const unsigned long long wq = atomicAdd(A.cells + A.cur_idx, 1ull);
            *(volatile lds_i32 *)&sy->widx = (int)(wq < 0x7fffffffull ? wq : 0x7fffffffull);
        }
        __syncthreads();
        const unsigned q32 = (unsigned)__builtin_amdgcn_readfirstlane(*(volatile lds_i32 *)&sy->widx);
        if (q32 >= nwork) break;
        const int64_t w = (int64_t)A.wlist[q32];
        if (wave == 0) { poa_graph_reset(g); T.n_sorted = 0; T.flags_ok = 0; T.use = A.lds_marks; }
        int n_nodes = 0, err = 0;                                  // the team's view of the graph (wavefront 0 owns the PoaGraph registers)
        const int64_t s0 = A.win_first_seq[w], s1 = A.win_first_seq[w + 1];
        for (int64_t s = s0; s < s1; ++s) {
            const uint8_t *seq = A.arena + A.seq_off[s];
            const int len = A.seq_len[s];
            if (wave == 0) g.n_path = 0;
            bool ran_dp = false;
            if (n_nodes != 0 && len != 0 && err == 0) {
                ran_dp = true;
                const bool piped = !LONG || len <= POA_PIPE_MAXLEN;
                const int wp = piped ? POA_PIPE_STRIDE : poa_row_stride(len);
                const int64_t plane = (int64_t)(n_nodes + 1) * wp;
                PoaMatrices M = {mat, mat + plane, mat + 2 * plane, mat + 3 * plane, mat + 4 * plane, wp};
                int mi = -1, mj = -1;
                if (wave == 0) cells += (unsigned long long)n_nodes * (unsigned long long)len;
                PH_T0
                __builtin_amdgcn_s_setprio(0);
                if (piped) poa_dp_team<NW, POA_TEAM_RR, POA_TEAM_K>(g, M, A, seq, len, n_nodes, lds_raw, sy, wave, mi, mj);
                else poa_dp<8, NW>(g, M, A, seq, len, mi, mj, sy, wave, n_nodes);          // column blocks; the rows shared out like the team DP's
                // the row descriptors into the ring's LDS for the traceback (the path it writes shares memory with two of them)
#ifndef GBX_POA_TB_NOLDS
#define GBX_POA_TB_NOLDS 0            // tuning aid: 1 = the team's traceback reads its descriptors from memory, as the one-wavefront kernel's
#endif
                const bool ld = piped && !GBX_POA_TB_NOLDS && n_nodes * 12 <= POA_TEAM_RING_BYTES;
                if (ld) {
                    typedef __attribute__((address_space(3))) unsigned short lds_u16;
                    lds_u16 *const ldw = (lds_u16 *)(lds_u8 *)lds_raw;
                    for (int r = threadIdx.x; r < n_nodes; r += 64 * NW) {
                        ldw[r] = (unsigned short)g.score[r]; ldw[n_nodes + r] = (unsigned short)g.path_node[r];
                        ldw[2 * n_nodes + r] = (unsigned short)g.path_pos[r]; ldw[3 * n_nodes + r] = (unsigned short)g.stack[r];
                        ldw[4 * n_nodes + r] = (unsigned short)g.pred[r]; ldw[5 * n_nodes + r] = (unsigned short)g.r2n[r];
                    }
                    __syncthreads();
                }
                PH_ACC(t_dp)
                if (wave == 0) {
                    __builtin_amdgcn_s_setprio(3);
                    if (ld) poa_traceback_wave<1>(g, M, A.S, seq, len, mi, mj, (lds_cu16 *)(lds_u8 *)lds_raw, n_nodes);
                    else if (piped) poa_traceback_wave(g, M, A.S, seq, len, mi, mj);
                    else poa_traceback(g, M, A.S, seq, mi, mj);
                    PH_ACC(t_tb)
#ifdef GBX_POA_PHASE_STATS
                    n_rows += (unsigned long long)n_nodes; n_steps += (unsigned long long)g.n_path;
#endif
                }
            }
            if (wave == 0) {
                PH_T0
                if (T.use && ran_dp) {
                    // the DP's ring has used the sort's LDS: state bytes back from the slot, previous ranks = n2r
                    const uint8_t *save = (const uint8_t *)(slot + L.st8save);
                    for (int i = lane; i < g.n_nodes; i += 64) { T.st8[i] = save[i]; T.old[i] = i < T.n_sorted ? (short)g.n2r[i] : (short)-1; }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                }
                if (g.err == 0) poa_add_alignment_wave<true>(g, seq, len, T);
                if (T.use && s + 1 < s1) {
                    uint8_t *save = (uint8_t *)(slot + L.st8save);
                    for (int i = lane; i < g.n_nodes; i += 64) save[i] = T.st8[i];
                }
                if (lane == 0) { *(volatile lds_i32 *)&sy->n_nodes = g.n_nodes; *(volatile lds_i32 *)&sy->err = g.err; }
                PH_ACC(t_add)
            }
            __syncthreads();                                       // the graph as wavefront 0 left it, for everybody
            n_nodes = __builtin_amdgcn_readfirstlane(*(volatile lds_i32 *)&sy->n_nodes);
            err = __builtin_amdgcn_readfirstlane(*(volatile lds_i32 *)&sy->err);
        }
        if (wave == 0) {
            PH_T0
            int clen = 0;
            if (g.err == 0) clen = poa_consensus(g, A.cons + w * A.cons_stride, (int)A.cons_stride);
            if (lane == 0) { A.cons_len[w] = clen; A.status[w] = g.err; }
            PH_ACC(t_cons)
        }
        __syncthreads();                                           // (nobody is still reading this window's work item when the next one is written)
    }
    if (threadIdx.x == 0) atomicAdd(A.cells, cells);
#ifdef GBX_POA_PHASE_STATS
    // (wavefront 0's clocks: the DP's are the team's wall clock for it, the other phases are its own)
    if (threadIdx.x == 0) { atomicAdd(A.cells + 1, t_dp); atomicAdd(A.cells + 2, t_tb); atomicAdd(A.cells + 3, t_add); atomicAdd(A.cells + 4, t_cons); atomicAdd(A.cells + 12, n_rows); atomicAdd(A.cells + 13, n_steps); }
#endif
}


// ---- the serial phases as ONE out-of-line function (round 5) -----------------------------------------------------------------------
// poa_kernel and poa_team_kernel above are single functions: every phase is inlined, and the PoaGraph (25 pointers), the output
// pointers and the sort's LDS arrays are live from a window's first sequence to its consensus - through the DP's row loop too, which
// needs a dozen of them.  258-332 scalar registers spill; the allocator keeps some of the live-through values in registers and
// reloads loop constants instead (the scan's matrices, one v_readlane per use): 34 reloads in the row loop.
// Here the kernel's body is the DP and nothing else.  Traceback, add_alignment with the sort, the row descriptors of the next
// alignment and the consensus are poa_serial_call: NOT inlined, it binds the graph by itself - from the launch's arguments, which it
// reads out of the kernel-argument segment (scalar loads through the pointer the kernel passes; the workgroup id comes with the call), and the window's five words of
// state, which travel as arguments and return value.  One call per sequence; inside it the allocator starts from nothing, and in the
// kernel body nothing of the graph is live across the row loop.  (A first attempt at out-of-line phases, round 3, passed `PoaGraph &`:
// the struct went to scratch memory and every access became FLAT.  Nothing is passed by reference here.)
struct PoaKernArgs { PoaArgs A; SlotLayout L; };
static_assert(sizeof(PoaKernArgs) % 4 == 0, "copied word by word out of the kernel-argument segment");
struct PoaWinState { int n_nodes, n_codes, err, n_sorted, flags_ok; };
constexpr int POA_SF_FIRST = 1, POA_SF_RAN_DP = 2, POA_SF_LAST = 4, POA_SF_LONGSEQ = 8, POA_SF_EMPTY = 16;

__device__ inline int poa_uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ inline int64_t poa_uni64(int64_t v)
{
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v & 0xffffffff)), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32));
    return (int64_t)(((unsigned long long)hi << 32) | lo);
}

// WAVES: wavefronts per SIMD of the calling kernel (an instance per caller: the compiler hands the caller's register budget down
// to a function all of whose callers agree on it).  LONG: the caller's slots have five planes, sequences over 512 bases may come.
// DESC: also builds the row descriptors of the next alignment (the one-wavefront kernel; the team builds them with all its lanes).
template <int WAVES, bool LONG, bool INC, bool DESC>
__device__ __attribute__((noinline)) PoaWinState poa_serial_call(PoaWinState st, int64_t w, int64_t s, int mi, int mj, int flags, unsigned long long kernargs)
{
    // arguments arrive in vector registers: everything below must know that they are the same in every lane
    st.n_nodes = poa_uni(st.n_nodes); st.n_codes = poa_uni(st.n_codes); st.err = poa_uni(st.err);
    st.n_sorted = poa_uni(st.n_sorted); st.flags_ok = poa_uni(st.flags_ok);
    w = poa_uni64(w); s = poa_uni64(s); mi = poa_uni(mi); mj = poa_uni(mj); flags = poa_uni(flags);
    // (a struct cannot be copy-constructed out of the constant address space: its words are loaded and put together again; the
    // loads are scalar - s_load - and only those of fields that are used survive)
    typedef const __attribute__((address_space(4))) int kernarg_word_t;
    // (the kernel hands its kernel-argument pointer down: asked for inside a callee, __builtin_amdgcn_kernarg_segment_ptr() faults on
    // this toolchain - scripts/kernarg_probe.hip)
    kernarg_word_t *const kw = (kernarg_word_t *)(unsigned long long)poa_uni64((int64_t)kernargs);
    PoaKernArgs KK;
    {
        int words[sizeof(PoaKernArgs) / 4];
#pragma unroll
        for (unsigned k = 0; k < sizeof(PoaKernArgs) / 4; ++k) words[k] = kw[k];
        __builtin_memcpy(&KK, words, sizeof(PoaKernArgs));
    }
    const PoaArgs &A = KK.A;
    const SlotLayout &L = KK.L;
    char *slot = A.work + (int64_t)blockIdx.x * A.slot_bytes;
    PoaGraph g;
    poa_bind_graph(g, slot, L, A);
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    PoaTopoLds T;
    poa_bind_lds(T, lds_raw, A);
    g.n_nodes = st.n_nodes; g.n_codes = st.n_codes; g.err = st.err; g.n_path = 0; g.path_lo = g.path_hi = -1;
    T.n_sorted = st.n_sorted; T.flags_ok = st.flags_ok;
    if (st.n_sorted < 0) { T.use = 0; T.n_sorted = 0; }       // (the window has outgrown the sort's LDS arrays: see poa_add_alignment_wave)
    const int lane = threadIdx.x & 63;
    if (flags & POA_SF_FIRST) { poa_graph_reset(g); T.n_sorted = 0; T.flags_ok = 0; T.use = A.lds_marks; }
    if (!(flags & POA_SF_EMPTY)) {
        const uint8_t *seq = A.arena + A.seq_off[s];
        const int len = A.seq_len[s];
        poa_cell_t *mat = (poa_cell_t *)(slot + L.mat);
        const bool ran_dp = (flags & POA_SF_RAN_DP) != 0;
        if (ran_dp) {
            const bool piped = !LONG || !(flags & POA_SF_LONGSEQ);
            const int wp = piped ? POA_PIPE_STRIDE : poa_row_stride(len);
            const int64_t plane = (int64_t)(g.n_nodes + 1) * wp;
            PoaMatrices M = {mat, mat + plane, mat + 2 * plane, mat + 3 * plane, mat + 4 * plane, wp};
#ifdef GBX_POA_PHASE_STATS
            const unsigned long long t0_ = __builtin_readcyclecounter();
#endif
            if (piped) poa_traceback_wave(g, M, A.S, seq, len, mi, mj);
            else if (LONG) { poa_dp<8>(g, M, A, seq, len, mi, mj); poa_traceback(g, M, A.S, seq, mi, mj); }
#ifdef GBX_POA_PHASE_STATS
            if (lane == 0) { atomicAdd(A.cells + 2, __builtin_readcyclecounter() - t0_); atomicAdd(A.cells + 13, (unsigned long long)g.n_path); }
#endif
        }
#ifdef GBX_POA_PHASE_STATS
        const unsigned long long t1_ = __builtin_readcyclecounter();
#endif
        if (T.use && ran_dp) {
            // the DP's row ring has used the sort's LDS: state bytes back from the slot, previous ranks = n2r
            const uint8_t *save = (const uint8_t *)(slot + L.st8save);
            for (int i = lane; i < g.n_nodes; i += 64) { T.st8[i] = save[i]; T.old[i] = i < T.n_sorted ? (short)g.n2r[i] : (short)-1; }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        if (g.err == 0) poa_add_alignment_wave<INC>(g, seq, len, T);
        if (T.use && !(flags & POA_SF_LAST)) {
            uint8_t *save = (uint8_t *)(slot + L.st8save);
            for (int i = lane; i < g.n_nodes; i += 64) save[i] = T.st8[i];
        }
        if (DESC && !(flags & POA_SF_LAST) && g.err == 0) {
            // the row descriptors of the graph as it now is: what the next alignment's DP and traceback read (the path arrays and the
            // sort's order buffer, which three of them share, are free again)
            int32_t *d_pred3 = g.stack;
            for (int r = lane; r < g.n_nodes; r += 64) {
                poa_rowdesc_one(g, r);
                const int node = g.r2n[r];
                d_pred3[r] = g.in_cnt[node] > 3 ? g.n2r[PG_IN_SRC(g, node, 3)] + 1 : 0;
            }
        }
#ifdef GBX_POA_PHASE_STATS
        if (lane == 0) atomicAdd(A.cells + 3, __builtin_readcyclecounter() - t1_);
#endif
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (flags & POA_SF_LAST) {
#ifdef GBX_POA_PHASE_STATS
        const unsigned long long t2_ = __builtin_readcyclecounter();
#endif
        int clen = 0;
        if (g.err == 0) clen = poa_consensus(g, A.cons + w * A.cons_stride, (int)A.cons_stride);
        if (lane == 0) { A.cons_len[w] = clen; A.status[w] = g.err; }
#ifdef GBX_POA_PHASE_STATS
        if (lane == 0) atomicAdd(A.cells + 4, __builtin_readcyclecounter() - t2_);
#endif
    }
    PoaWinState out = {g.n_nodes, g.n_codes, g.err, A.lds_marks && !T.use ? -1 : T.n_sorted, T.flags_ok};
    return out;
}

// the few graph arrays the DP itself reads (row descriptors; the in-edge lists for a fifth predecessor), bound for the DP alone
__device__ __attribute__((always_inline)) inline void poa_bind_dp_graph(PoaGraph &g, char *slot, const SlotLayout &L, const PoaArgs &A, bool with_lists)
{
    g.ncap = A.ncap; g.deg = A.deg;
    g.r2n = (int32_t *)(slot + L.r2n); g.n2r = (int32_t *)(slot + L.n2r);
    g.in_src = (int32_t *)(slot + L.in_src); g.in_src_x = (int32_t *)(slot + L.in_src_x);
    g.stack = (int32_t *)(slot + L.stack); g.score = (int32_t *)(slot + L.score); g.pred = (int32_t *)(slot + L.pred);
    g.path_node = (int32_t *)(slot + L.path_node); g.path_pos = (int32_t *)(slot + L.path_pos);
    if (with_lists) {                                          // (the team builds the descriptors itself: it needs what poa_rowdesc_one reads)
        g.in_cnt = (uint8_t *)(slot + L.in_cnt); g.out_cnt = (uint8_t *)(slot + L.out_cnt);
        g.code = (uint8_t *)(slot + L.code); g.decoder = (uint8_t *)(slot + L.decoder);
    }
}

// the window kernel (one wavefront per window) with the serial phases out of line
template <bool LONG, int WAVES = 3, int RROWS = POA_RING_DEFAULT>
__global__ void __launch_bounds__(64, WAVES) poa_kernel2(PoaKernArgs K)
{
    const PoaArgs &A = K.A;
    char *slot = A.work + (int64_t)blockIdx.x * A.slot_bytes;
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    const int lane = threadIdx.x & 63;
    if (LONG && lane == 0) atomicAdd(A.cells + POA_LONG_STARTED, 1ull);
    unsigned long long cells = 0;
#ifdef GBX_POA_PHASE_STATS
    unsigned long long t_dp = 0, n_rows = 0;
#endif
    const unsigned nwork = (unsigned)A.cells[A.cnt_idx];
    for (;;) {
        unsigned long long wq = 0;
        if (lane == 0) wq = atomicAdd(A.cells + A.cur_idx, 1ull);
        const unsigned q32 = (unsigned)__builtin_amdgcn_readfirstlane((int)wq);
        if (q32 >= nwork) break;
        const int64_t w = (int64_t)A.wlist[q32];
        PoaWinState st = {0, 0, 0, 0, 0};
        const int64_t s0 = A.win_first_seq[w], s1 = A.win_first_seq[w + 1];
        if (s0 == s1) st = poa_serial_call<WAVES, LONG, true, true>(st, w, s0, -1, -1, POA_SF_FIRST | POA_SF_LAST | POA_SF_EMPTY, (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr());
        for (int64_t s = s0; s < s1; ++s) {
            const int len = A.seq_len[s];
            int mi = -1, mj = -1, flags = (s == s0 ? POA_SF_FIRST : 0) | (s + 1 == s1 ? POA_SF_LAST : 0);
            if (st.n_nodes != 0 && len != 0 && st.err == 0) {
                flags |= POA_SF_RAN_DP;
                cells += (unsigned long long)st.n_nodes * (unsigned long long)len;
                if (!LONG || len <= POA_PIPE_MAXLEN) {
                    const uint8_t *seq = A.arena + A.seq_off[s];
                    poa_cell_t *mat = (poa_cell_t *)(slot + K.L.mat);
                    const int64_t plane = (int64_t)(st.n_nodes + 1) * POA_PIPE_STRIDE;
                    PoaMatrices M = {mat, mat + plane, mat + 2 * plane, mat + 3 * plane, mat + 4 * plane, POA_PIPE_STRIDE};
                    PoaGraph gd;
                    poa_bind_dp_graph(gd, slot, K.L, A, false);
                    gd.n_nodes = st.n_nodes;
#ifdef GBX_POA_PHASE_STATS
                    const unsigned long long t0_ = __builtin_readcyclecounter();
#endif
                    __builtin_amdgcn_s_setprio(0);
                    poa_dp_pipelined<RROWS, true>(gd, M, A, seq, len, mi, mj, lds_raw);
                    __builtin_amdgcn_s_setprio(3);
#ifdef GBX_POA_PHASE_STATS
                    t_dp += __builtin_readcyclecounter() - t0_; n_rows += (unsigned long long)st.n_nodes;
#endif
                } else flags |= POA_SF_LONGSEQ;
            }
            st = poa_serial_call<WAVES, LONG, true, true>(st, w, s, mi, mj, flags, (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr());
            st.n_nodes = poa_uni(st.n_nodes); st.n_codes = poa_uni(st.n_codes); st.err = poa_uni(st.err);
            st.n_sorted = poa_uni(st.n_sorted); st.flags_ok = poa_uni(st.flags_ok);
        }
    }
    if (lane == 0) atomicAdd(A.cells, cells);
#ifdef GBX_POA_PHASE_STATS
    if (lane == 0) { atomicAdd(A.cells + 1, t_dp); atomicAdd(A.cells + 12, n_rows); }
#endif
}

// the team kernel with the serial phases out of line
template <bool LONG, int WAVES>
__global__ void __launch_bounds__(64 * POA_TEAM_NW, WAVES) poa_team2_kernel(PoaKernArgs K, int sync_off)
{
    constexpr int NW = POA_TEAM_NW;
    const PoaArgs &A = K.A;
    char *slot = A.work + (int64_t)blockIdx.x * A.slot_bytes;
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    lds_team *const sy = (lds_team *)((lds_u8 *)lds_raw + sync_off);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    if (LONG && threadIdx.x == 0) atomicAdd(A.cells + POA_LONG_STARTED, 1ull);

    unsigned long long cells = 0;
    const unsigned nwork = (unsigned)A.cells[A.cnt_idx];
    for (;;) {
        if (threadIdx.x == 0) {
            const unsigned long long wq = atomicAdd(A.cells + A.cur_idx, 1ull);
            *(volatile lds_i32 *)&sy->widx = (int)(wq < 0x7fffffffull ? wq : 0x7fffffffull);
        }
        __syncthreads();
        const unsigned q32 = (unsigned)__builtin_amdgcn_readfirstlane(*(volatile lds_i32 *)&sy->widx);
        if (q32 >= nwork) break;
        const int64_t w = (int64_t)A.wlist[q32];
        PoaWinState st = {0, 0, 0, 0, 0};
        int n_nodes = 0, err = 0;
        const int64_t s0 = A.win_first_seq[w], s1 = A.win_first_seq[w + 1];
        if (s0 == s1 && wave == 0) st = poa_serial_call<WAVES, LONG, true, false>(st, w, s0, -1, -1, POA_SF_FIRST | POA_SF_LAST | POA_SF_EMPTY, (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr());
        for (int64_t s = s0; s < s1; ++s) {
            const int len = A.seq_len[s];
            int mi = -1, mj = -1, flags = (s == s0 ? POA_SF_FIRST : 0) | (s + 1 == s1 ? POA_SF_LAST : 0);
            if (n_nodes != 0 && len != 0 && err == 0) {
                flags |= POA_SF_RAN_DP;
                const bool piped = !LONG || len <= POA_PIPE_MAXLEN;
                if (wave == 0) cells += (unsigned long long)n_nodes * (unsigned long long)len;
                if (piped) {
                    const uint8_t *seq = A.arena + A.seq_off[s];
                    poa_cell_t *mat = (poa_cell_t *)(slot + K.L.mat);
                    const int64_t plane = (int64_t)(n_nodes + 1) * POA_PIPE_STRIDE;
                    PoaMatrices M = {mat, mat + plane, mat + 2 * plane, mat + 3 * plane, mat + 4 * plane, POA_PIPE_STRIDE};
                    PoaGraph gd;
                    poa_bind_dp_graph(gd, slot, K.L, A, true);
#ifdef GBX_POA_PHASE_STATS
                    const unsigned long long t0_ = __builtin_readcyclecounter();
#endif
                    __builtin_amdgcn_s_setprio(0);
                    poa_dp_team<NW, POA_TEAM_RR, POA_TEAM_K>(gd, M, A, seq, len, n_nodes, lds_raw, sy, wave, mi, mj);
                    __builtin_amdgcn_s_setprio(3);
#ifdef GBX_POA_PHASE_STATS
                    if (threadIdx.x == 0) { atomicAdd(A.cells + 1, __builtin_readcyclecounter() - t0_); atomicAdd(A.cells + 12, (unsigned long long)n_nodes); }
#endif
                } else flags |= POA_SF_LONGSEQ;                  // column blocks: DP and traceback inside the call
            }
            if (wave == 0) {
                st = poa_serial_call<WAVES, LONG, true, false>(st, w, s, mi, mj, flags, (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr());
                if (lane == 0) { *(volatile lds_i32 *)&sy->n_nodes = st.n_nodes; *(volatile lds_i32 *)&sy->err = st.err; }
            }
            __syncthreads();
            n_nodes = __builtin_amdgcn_readfirstlane(*(volatile lds_i32 *)&sy->n_nodes);
            err = __builtin_amdgcn_readfirstlane(*(volatile lds_i32 *)&sy->err);
            if (wave == 0) { st.n_nodes = n_nodes; st.err = err; st.n_codes = poa_uni(st.n_codes); st.n_sorted = poa_uni(st.n_sorted); st.flags_ok = poa_uni(st.flags_ok); }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) atomicAdd(A.cells, cells);
}


// ---- the lock-step form (round 4): every window resident, one launch per phase and sequence index ----------------
// poa_kernel keeps a window in one wavefront from its first sequence to its consensus: the DP (throughput work: packed
// arithmetic and row traffic), the traceback (one dependent row fetch per step) and add_alignment / the sort (chains of
// dependent LDS reads) share one register budget (168 VGPRs, 234 spilled SGPRs) and one occupancy (12 windows per CU),
// and a CU's issue slots and memory queues are shared by whatever mix of phases its twelve windows happen to be in.
// With a slot per window (288 GB of HBM: 'large' takes 55 GB) the phases become launches of their own over ALL windows:
// for sequence index s = 1, 2, ...: poa_phase_kernel<DP> aligns sequence s of every window that has one against its graph,
// poa_phase_kernel<serial> adds the alignments and re-sorts.  Each kernel holds only what its phase needs (the DP no sort
// state and no LDS, the serial phases a third of the DP's registers), every wavefront of a launch is in the same phase,
// and what a window carries from launch to launch - the graph's counters, the alignment's end point and path, the state
// bytes of the incremental sort - sits in its slot (PoaSlotHdr, st8save; the previous ranks are n2r).  The windows with a
// sequence over 512 bases keep the window kernel (second launch, side stream), small jobs too.
// MEASURED (round 4, 6 000 windows, profiles/r04b_poa_variants.txt): 310 ms (traceback with the DP, 5 wavefronts per SIMD),
// 334 (6 per SIMD: all windows in flight), 327 / 334 (traceback with the serial phases) against 233 ms of the window kernel.
// The DP launches alone take 5.5 ms x 39: a chip doing nothing but DP rows moves 3.7 TB/s, which is what the window kernel
// averages WITH its serial phases hidden behind it - the DP is bound by its HBM traffic (8.8 B per cell), not by registers,
// occupancy or issue slots (300 VALU instructions per row: 45 % of the issue rate), and lock-step only takes away the
// overlap.  What the measurement pointed to instead is the traffic itself: poa_dp_pipelined's row ring.
struct PoaSlotHdr { int n_nodes, n_codes, n_path, err, path_lo, path_hi, mi, mj, n_sorted, flags_ok, dp_ran, pad_[5]; };
static_assert(sizeof(PoaSlotHdr) == 64, "one 64-byte line per window");

// WAVES = wavefronts per SIMD the instance is compiled for (DP alone: 87 VGPRs as the compiler likes it = 5, 80 VGPRs with two
// spilled = 6, i.e. 24 windows per CU: all 6 000 of 'large' in flight at once)
template <bool DO_DP, bool DO_TB, bool DO_ADD, int WAVES>
__global__ void __launch_bounds__(64, WAVES) poa_phase_kernel(PoaArgs A, SlotLayout L, int s_idx)
{
    const int q = blockIdx.x;                                  // position in the work list = the window's slot
    const int64_t w = (int64_t)A.wlist[q];
    const int64_t s0 = A.win_first_seq[w], s1 = A.win_first_seq[w + 1];
    const int nseq = (int)(s1 - s0);
    if (s_idx >= nseq && !(DO_ADD && s_idx == 0)) return;      // (a window without sequences still gets its empty consensus)
    char *slot = A.work + (int64_t)q * A.slot_bytes;
    PoaSlotHdr *hdr = (PoaSlotHdr *)(slot + L.hdr);
    PoaGraph g;
    poa_bind_graph(g, slot, L, A);
    const int lane = threadIdx.x & 63;
    if (DO_ADD && s_idx == 0) {
        poa_graph_reset(g);
        g.path_lo = g.path_hi = -1;
        if (nseq == 0) {
            const int clen = poa_consensus(g, A.cons + w * A.cons_stride, (int)A.cons_stride);
            if (lane == 0) { A.cons_len[w] = clen; A.status[w] = g.err; }
            return;
        }
    } else {
        g.n_nodes = hdr->n_nodes; g.n_codes = hdr->n_codes; g.n_path = hdr->n_path; g.err = hdr->err;
        g.path_lo = hdr->path_lo; g.path_hi = hdr->path_hi;
    }
    const uint8_t *seq = A.arena + A.seq_off[s0 + s_idx];
    const int len = A.seq_len[s0 + s_idx];
    poa_cell_t *mat = (poa_cell_t *)(slot + L.mat);
    const int wp = POA_PIPE_STRIDE;
    const int64_t plane = (int64_t)(g.n_nodes + 1) * wp;
    const PoaMatrices M = {mat, mat + plane, mat + 2 * plane, mat + 3 * plane, mat + 4 * plane, wp};
    int mi = -1, mj = -1, dp_ran = 0;
    if (DO_DP) {
        g.n_path = 0;
        if (g.n_nodes != 0 && len != 0 && g.err == 0) {
            extern __shared__ __attribute__((aligned(16))) char lds_dp[];
            poa_dp_pipelined<POA_RING_DEFAULT>(g, M, A, seq, len, mi, mj, lds_dp);     // the row ring: this kernel has no other use for LDS
            dp_ran = 1;
            if (lane == 0) atomicAdd(A.cells, (unsigned long long)g.n_nodes * (unsigned long long)len);
        }
        if (!DO_TB && lane == 0) { hdr->mi = mi; hdr->mj = mj; hdr->dp_ran = dp_ran; hdr->n_path = 0; }
    } else if (DO_TB) {
        mi = hdr->mi; mj = hdr->mj; dp_ran = hdr->dp_ran;
        if (s_idx == 0) dp_ran = 0;
    }
    if (DO_TB) {
        if (dp_ran) poa_traceback_wave(g, M, A.S, seq, len, mi, mj);
        else g.n_path = 0;
        if (!DO_ADD && lane == 0) { hdr->n_path = g.n_path; hdr->path_lo = g.path_lo; hdr->path_hi = g.path_hi; hdr->err = g.err; }
    }
    if (DO_ADD) {
        extern __shared__ __attribute__((aligned(16))) char lds_raw[];
        PoaTopoLds T;
        poa_bind_lds(T, lds_raw, A);
        uint8_t *const save = (uint8_t *)(slot + L.st8save);
        if (s_idx != 0) {
            T.n_sorted = hdr->n_sorted; T.flags_ok = hdr->flags_ok;
            if (T.n_sorted < 0) { T.use = 0; T.n_sorted = 0; }       // (the window has outgrown the sort's LDS arrays)
            if (T.use) {
                // the sort's state as the previous launch left it: state bytes from the slot, previous ranks = n2r
                for (int i = lane; i < g.n_nodes; i += 64) { T.st8[i] = save[i]; T.old[i] = i < T.n_sorted ? (short)g.n2r[i] : (short)-1; }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        }
        if (g.err == 0) poa_add_alignment_wave<true>(g, seq, len, T);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (s_idx == nseq - 1) {
            int clen = 0;
            if (g.err == 0) clen = poa_consensus(g, A.cons + w * A.cons_stride, (int)A.cons_stride);
            if (lane == 0) { A.cons_len[w] = clen; A.status[w] = g.err; }
        } else {
            if (T.use) for (int i = lane; i < g.n_nodes; i += 64) save[i] = T.st8[i];
            if (lane == 0) {
                hdr->n_nodes = g.n_nodes; hdr->n_codes = g.n_codes; hdr->n_path = 0; hdr->err = g.err; hdr->path_lo = -1; hdr->path_hi = -1;
                hdr->n_sorted = A.lds_marks && !T.use ? -1 : T.n_sorted; hdr->flags_ok = T.flags_ok; hdr->dp_ran = 0;
            }
        }
    }
}

}  // namespace

// wavefronts (= windows in flight) one CU keeps resident for this node capacity: registers and the LDS of
// the topological sort decide
// LDS of the topological sort per window: state byte + previous rank per node, the block cache, and a DFS stack
// sized so that as many windows as the registers allow (12 per CU at 168 VGPRs) fit the CU's 160 KB: the stack
// takes what is left of a window's share, between 128 and POA_LDS_STACK16 entries (a deeper walk falls back to the
// global-memory sort).  Returns the bytes, or 0 when the node capacity does not fit LDS at all.
static size_t poa_lds_plan(int ncap, int *stack_entries, int *lds_ncap)
{
    if (ncap >= 32768) return 0;
    int max_waves = 12;
    if (const char *e = getenv("GBX_POA_MAX_WAVES")) { const int v = atoi(e); if (v >= 8 && v <= 16) max_waves = v; }   // tuning aid
    // measured (node capacity 3364): twelve windows of 12304 B run together (336 ms); at 13024 B the twelfth is
    // resident only some of the time (362-386 ms), so the budget is 12 x 12544 B, not the nominal 160 KB.
    // Round 5: the per-node arrays hold what fits a window's share, not the graph's capacity: a window that outgrows them sorts in
    // global memory from then on (before, a job whose capacity did not fit lost windows per CU - or the LDS sort altogether - for all
    // its windows).
    const size_t share = ((size_t)12 * 12544 / (size_t)max_waves) & ~(size_t)31;
    int fit = (int)((share - POA_LDS_FIXED) / 3) & ~15;
    const int ncp = (ncap + 15) & ~15;
    if (fit > ncp) fit = ncp;
    if (const char *e = getenv("GBX_POA_LDS_NCAP")) { const int v = atoi(e) & ~15; if (v >= 64 && v < fit) fit = v; }   // test aid: windows outgrow the arrays early
    else if (fit < 512) return 0;
    size_t st = POA_LDS_STACK16;                          // the stack's region is fixed; fewer entries only as a tuning aid
    if (const char *e = getenv("GBX_POA_LDS_STACK")) { const size_t v = (size_t)atoi(e); if (v >= 32 && v <= st) st = v; }   // tuning aid
    *stack_entries = (int)st;
    *lds_ncap = fit;
    return (size_t)3 * (size_t)fit + POA_LDS_FIXED;
}

int poa_waves_per_cu(int ncap)
{
    int lds_stack = 0, lds_ncap = 0;
    const size_t lds_need = poa_lds_plan(ncap, &lds_stack, &lds_ncap);
    const bool lds_marks = lds_need != 0;
    int q = 0;
    const char *oe_ = getenv("GBX_POA_OCC");
    const bool occ4 = oe_ && atoi(oe_) == 4;                 // tuning aid: the 128-VGPR / four-ring-row instance
    const hipError_t qe = occ4 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&q, poa_kernel<false, 4, 4>, 64, std::max<size_t>(lds_marks ? lds_need : 0, (size_t)4 * POA_RING_SLOT))
                               : hipOccupancyMaxActiveBlocksPerMultiprocessor(&q, poa_kernel<false>, 64, std::max<size_t>(lds_marks ? lds_need : 0, (size_t)POA_RING_BYTES));
    if (qe != hipSuccess || q < 1) {
        (void)hipGetLastError();
        q = 8;
    }
    // measured on MI355X (6000 windows, cursor schedule): 8 per CU 410 ms, 9: 393, 10: 378, 11: 361, 12: 329 - twelve is what
    // 168 VGPRs admit; poa_lds_plan sizes the LDS stack so that twelve fit
    const int hw = q;
    int max_waves = 12;
    if (const char *e = getenv("GBX_POA_MAX_WAVES")) { const int v = atoi(e); if (v >= 8 && v <= 16) max_waves = v; }
    if (q > max_waves) q = max_waves;
    if (const char *e = getenv("GBX_POA_WAVES_PER_CU")) {      // tuning aid: another number of windows in flight (up to what the hardware admits)
        const int v = atoi(e);
        if (v >= 1 && v <= hw) q = v;
    }
    return q;
}

// The lock-step form needs a slot per window of the main list and pays for ~2 launches per sequence index.  It is built,
// tested and selectable (GBX_POA_LOCKSTEP=1 when the plan is made and at the launch) but NOT the default: see the measurement
// at poa_phase_kernel.
bool poa_lockstep_wanted(int64_t n_main, int64_t resident)
{
    if (const char *e = getenv("GBX_POA_LOCKSTEP")) return atoi(e) != 0 && n_main > 0;
    (void)resident;
    return false;          // measured (MI355X, 'large'): 310-334 ms against the window kernel's 233 - see the note at poa_phase_kernel
}
namespace {
bool poa_use_lockstep(const gbx_poa_plan *plan, int64_t n_main)
{
    if (n_main <= 0 || plan->n_slots < n_main) return false;
    int cus = 256, dev = 0;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    else (void)hipGetLastError();
    return poa_lockstep_wanted(n_main, (int64_t)cus * poa_waves_per_cu(plan->node_cap));
}
}  // namespace

// windows of a main list up to which the team kernel takes it: what the chip keeps resident as team workgroups (three per CU at
// 168 VGPRs; beyond that the workgroups queue and one wavefront per window, twelve per CU, is the better use of the SIMDs)
static int64_t poa_team_max_windows()
{
    if (const char *e = getenv("GBX_POA_TEAM_MAX")) return atoll(e);          // tuning aid
    int cus = 256, dev = 0;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    else (void)hipGetLastError();
    return (int64_t)cus * (GBX_POA_TEAM_WAVES + 1);          // (measured: 1 024 windows 85 against 91 ms, 1 500 windows 120 against 96)
}

// workspace = main slots | counter block | main work list | long-window list | long slots
size_t poa_slot_bytes(int ncap, int deg, int lmax, bool long_slot) { return (size_t)make_layout(ncap, deg, lmax, long_slot).total; }
namespace {
struct PoaWs { size_t counters, wlist, llist, lslots, total; };
PoaWs poa_ws(const gbx_poa_plan *pl)
{
    PoaWs w;
    const size_t ms = poa_slot_bytes(pl->node_cap, pl->max_seqs_per_window, pl->max_seq_len, false);
    const size_t ls = pl->long_slots > 0 ? poa_slot_bytes(pl->node_cap, pl->max_seqs_per_window, pl->max_seq_len, true) : 0;
    const size_t nw = (size_t)(pl->n_windows > 0 ? pl->n_windows : 0), nl = (size_t)(pl->n_long_windows > 0 ? pl->n_long_windows : 0);
    w.counters = ms * (size_t)(pl->n_slots > 0 ? pl->n_slots : 0);
    w.wlist = w.counters + POA_NCOUNTERS * 8;
    w.llist = w.wlist + ((nw * 4 + 255) & ~(size_t)255);
    w.lslots = w.llist + ((nl * 4 + 255) & ~(size_t)255);
    w.total = w.lslots + ls * (size_t)pl->long_slots;
    return w;
}
}  // namespace
size_t poa_workspace_bytes(const gbx_poa_plan *plan) { return poa_ws(plan).total; }

int poa_read_cells(const void *d_work, size_t slots_bytes, int64_t *cells, hipStream_t s)
{
    unsigned long long v = 0;
    GBX_HIP(hipMemcpyAsync(&v, (const char *)d_work + slots_bytes, sizeof(v), hipMemcpyDeviceToHost, s));
    GBX_HIP(hipStreamSynchronize(s));
    *cells = (int64_t)v;
    return GBX_OK;
}

bool poa_scores_fit_int16(const gbx_poa_params *p, int64_t ncap, int lmax)
{
    PoaScore S = {p->m, p->n, p->g, p->e, p->q, p->c, 0};
    if (S.g >= S.e) { S.e = S.q = S.c = S.g; S.linear = 1; }           // the linear subtype (poa_graph.h: PoaScore)
    else if (S.g <= S.q || S.e >= S.c) { S.q = S.g; S.c = S.e; }
    const int64_t worst = -(int64_t)(S.q < S.g ? -S.q : -S.g) * 2 - (int64_t)(S.c > S.e ? -S.c : -S.e) * (ncap + lmax);
    const int64_t worst_mis = (int64_t)S.n * lmax, hi = (int64_t)S.m * lmax;
    // the pipelined DP keeps H - F and H - O in one byte each (PoaPredIn): H - F <= smax - max(g,q) - g, H - O likewise with q
    const int smax = S.m > S.n ? S.m : S.n, delta = smax - (S.g > S.q ? S.g : S.q);
    if (delta - S.g > 255 || delta - S.q > 255) return false;
    return !(worst < -30000 || worst_mis < -30000 || hi > 30000);
}

int poa_launch(const gbx_poa_params *p, const gbx_poa_plan *plan, int64_t n_windows, const int64_t *d_win_first_seq, const int64_t *d_seq_off,
               const int32_t *d_seq_len, const uint8_t *d_arena,
               uint8_t *d_cons, int32_t *d_cons_len, int32_t *d_status, int64_t cons_stride,
               void *d_work, size_t work_bytes, hipStream_t s)
{
    if (n_windows == 0) return GBX_OK;
    if (n_windows != plan->n_windows) { set_error("poa: the plan was made for %lld windows, the call has %lld", (long long)plan->n_windows, (long long)n_windows); return GBX_ERR_ARG; }
    const int lmax = plan->max_seq_len, deg = plan->max_seqs_per_window, ncap = plan->node_cap;
    PoaScore S = {p->m, p->n, p->g, p->e, p->q, p->c, 0};
    if (S.g > 0 || S.q > 0 || S.e > 0 || S.c > 0) { set_error("poa: gap penalties must be non-positive"); return GBX_ERR_ARG; }
    // spoa's createAlignmentEngine: g >= e is the LINEAR subtype (one gap cost g: msa_spoa_omp.cpp:170-196 lets -o / -e produce it),
    // g <= q or e >= c the affine one (convex with both pieces equal), else convex
    if (S.g >= S.e) { S.e = S.q = S.c = S.g; S.linear = 1; }
    else if (S.g <= S.q || S.e >= S.c) { S.q = S.g; S.c = S.e; }
    const PoaWs ws = poa_ws(plan);
    if (work_bytes < ws.total) { set_error("poa: workspace too small"); return GBX_ERR_ARG; }
    if (n_windows >= ((int64_t)1 << 31)) { set_error("poa: more than 2^31 windows in one call"); return GBX_ERR_UNSUPPORTED; }
    char *wb = (char *)d_work;
    unsigned long long *d_cells = (unsigned long long *)(wb + ws.counters);
    GBX_HIP(hipMemsetAsync(d_cells, 0, POA_NCOUNTERS * 8, s));
    const Mat2 T = {S.e, S.g, S.q, S.c};
    PoaArgs A;
    A.n_windows = n_windows; A.win_first_seq = d_win_first_seq; A.seq_off = d_seq_off; A.seq_len = d_seq_len;
    A.arena = d_arena; A.cons = d_cons; A.cons_len = d_cons_len; A.status = d_status; A.cons_stride = cons_stride;
    A.cells = d_cells; A.ncap = ncap; A.deg = deg; A.lmax = lmax; A.S = S;
    for (int v = 0; v < 2; ++v) {
        A.Tc[v][0] = mp_pow(T, v == 0 ? 8 : 16);
        for (int k = 1; k < 4; ++k) A.Tc[v][k] = mp_mul(A.Tc[v][k - 1], A.Tc[v][k - 1]);
    }
    // int16 cells: every real score must stay above -30000 (poa_graph.h); worst case = one long gap
    if (!poa_scores_fit_int16(p, ncap, lmax)) {
        set_error("poa: scores may leave the int16 range for these capacities (nodes %d, length %d)", ncap, lmax);
        return GBX_ERR_UNSUPPORTED;
    }
    int lds_stack = 0, lds_ncap = 0;
    const size_t lds_need = poa_lds_plan(ncap, &lds_stack, &lds_ncap);
    A.lds_marks = lds_need != 0 ? 1 : 0;
    A.lds_stack = lds_stack; A.lds_ncap = lds_ncap;
    int32_t *d_wlist = (int32_t *)(wb + ws.wlist), *d_llist = (int32_t *)(wb + ws.llist);
    A.work = wb; A.slot_bytes = 0; A.wlist = d_wlist; A.cnt_idx = POA_CNT_MAIN; A.cur_idx = POA_CUR_MAIN;
    hipLaunchKernelGGL(poa_classify_kernel, dim3(1), dim3(1024), 0, s, A, d_wlist, d_llist, plan->n_long_windows);
    // the few windows with a long sequence: their own launch on a side stream, beside the main one (alone they would be a
    // serial tail: a window takes tens of milliseconds whatever else runs)
    const bool has_long = plan->long_slots > 0 && plan->n_long_windows > 0;
    const bool has_main = plan->n_slots > 0 && n_windows > plan->n_long_windows;
    SideStreams *ss = nullptr;
    std::unique_lock<std::mutex> side_lock;
    int rc;
    if (has_long && has_main) {
        if ((rc = side_streams(&ss))) return rc;
        side_lock = std::unique_lock<std::mutex>(ss->mu);
        if ((rc = ss->fork(s))) return rc;
    }
    // the team kernel (a window per workgroup of four wavefronts): for the windows of the long launch, and for a main list with
    // fewer windows than the chip keeps team workgroups resident (a job as long as its slowest window: see poa_team_kernel)
    const bool team_on = !(getenv("GBX_POA_TEAM") && atoi(getenv("GBX_POA_TEAM")) == 0);
    const int serial_form = getenv("GBX_POA_SERIAL_FORM") ? atoi(getenv("GBX_POA_SERIAL_FORM")) : 1;      // 2: poa_kernel2 (serial phases out of line)
    const int team_form = getenv("GBX_POA_TEAM_FORM") ? atoi(getenv("GBX_POA_TEAM_FORM")) : 1;      // 2: the serial phases out of line (poa_serial_call)
    const size_t team_ring = std::max<size_t>(A.lds_marks ? lds_need : 0, (size_t)POA_TEAM_RING_BYTES);
    const int team_sync_off = (int)((team_ring + 15) & ~(size_t)15);
    const size_t team_lds = (size_t)team_sync_off + sizeof(PoaTeamSync);
    // (the team form of the long launch takes a whole CU per window - 512 VGPRs per wavefront - which a job of many windows would
    // rather give to its main launch: there the long windows, a wavefront each, end well before it anyway: 165 against 215 ms on
    // 'large'; measured with the team form: 81 ms for them, 226-231 for the job)
    const bool team_long = team_on && n_windows - plan->n_long_windows <= 4 * poa_team_max_windows();
    if (has_long && team_long) {
        const SlotLayout LL = make_layout(ncap, deg, lmax, true);
        PoaArgs B = A;
        B.work = wb + ws.lslots; B.slot_bytes = LL.total; B.wlist = d_llist; B.cnt_idx = POA_CNT_LONG; B.cur_idx = POA_CUR_LONG;
        hipStream_t sl = ss ? ss->side[0] : s;
        Stage st("poa_window_long", sl);
        if (team_form == 2) {
            PoaKernArgs KB; KB.A = B; KB.L = LL;
            hipLaunchKernelGGL((poa_team2_kernel<true, GBX_POA_TEAM_LONG_WAVES>), dim3(plan->long_slots), dim3(64 * POA_TEAM_NW), team_lds, sl, KB, team_sync_off);
        } else
        hipLaunchKernelGGL((poa_team_kernel<true, GBX_POA_TEAM_LONG_WAVES>), dim3(plan->long_slots), dim3(64 * POA_TEAM_NW), team_lds, sl, B, LL, team_sync_off);
    } else if (has_long) {
        const SlotLayout LL = make_layout(ncap, deg, lmax, true);
        PoaArgs B = A;
        B.work = wb + ws.lslots; B.slot_bytes = LL.total; B.wlist = d_llist; B.cnt_idx = POA_CNT_LONG; B.cur_idx = POA_CUR_LONG;
        hipStream_t sl = ss ? ss->side[0] : s;
        Stage st("poa_window_long", sl);
        // (compiled for one wavefront per SIMD: the handful of long windows of a job run a wavefront per CU at most, and with
        // all 512 VGPRs the instance - column-block DP, ring, both sorts - has no spills; at 168 it spilled 222)
        if (serial_form == 2) {
            PoaKernArgs KB; KB.A = B; KB.L = LL;
            hipLaunchKernelGGL((poa_kernel2<true, GBX_POA_LONG_WAVES>), dim3(plan->long_slots), dim3(64), std::max<size_t>(A.lds_marks ? lds_need : 0, (size_t)POA_RING_BYTES), sl, KB);
        } else
        hipLaunchKernelGGL((poa_kernel<true, GBX_POA_LONG_WAVES>), dim3(plan->long_slots), dim3(64), std::max<size_t>(A.lds_marks ? lds_need : 0, (size_t)POA_RING_BYTES), sl, B, LL);
    }
    const int64_t n_main = n_windows - plan->n_long_windows;
    if (has_main && poa_use_lockstep(plan, n_main)) {
        // lock-step form: a slot per window, per sequence index one DP (+ traceback) launch and one launch of the serial phases
        // over all windows of the list (heaviest class first; a window without a sequence s leaves at once)
        const SlotLayout L = make_layout(ncap, deg, lmax, false);
        A.slot_bytes = L.total;
        const dim3 grid((unsigned)n_main), tb(64);
        const size_t lds = A.lds_marks ? lds_need : 0;
        // tuning aids, read per call: GBX_POA_TB_SERIAL=1 moves the traceback from the DP launch to the serial one,
        // GBX_POA_DP_OCC=5|6 picks the DP instance
        const bool tb_with_dp = !(getenv("GBX_POA_TB_SERIAL") && atoi(getenv("GBX_POA_TB_SERIAL")) != 0);
        const int occ = getenv("GBX_POA_DP_OCC") ? atoi(getenv("GBX_POA_DP_OCC")) : GBX_POA_DP_WAVES;
        for (int sidx = 0; sidx < plan->max_seqs_per_window; ++sidx) {
            if (sidx > 0) {
                Stage st("poa_dp", s);
                if (tb_with_dp && occ >= 6) hipLaunchKernelGGL((poa_phase_kernel<true, true, false, 6>), grid, tb, (size_t)POA_RING_BYTES, s, A, L, sidx);
                else if (tb_with_dp) hipLaunchKernelGGL((poa_phase_kernel<true, true, false, 5>), grid, tb, (size_t)POA_RING_BYTES, s, A, L, sidx);
                else if (occ >= 6) hipLaunchKernelGGL((poa_phase_kernel<true, false, false, 6>), grid, tb, (size_t)POA_RING_BYTES, s, A, L, sidx);
                else hipLaunchKernelGGL((poa_phase_kernel<true, false, false, 5>), grid, tb, (size_t)POA_RING_BYTES, s, A, L, sidx);
            }
            Stage st("poa_serial", s);
            if (tb_with_dp) hipLaunchKernelGGL((poa_phase_kernel<false, false, true, GBX_POA_SERIAL_WAVES>), grid, tb, lds, s, A, L, sidx);
            else hipLaunchKernelGGL((poa_phase_kernel<false, true, true, GBX_POA_SERIAL_WAVES>), grid, tb, lds, s, A, L, sidx);
        }
    } else if (has_main && team_on && n_main <= poa_team_max_windows()) {
        const SlotLayout L = make_layout(ncap, deg, lmax, false);
        A.slot_bytes = L.total;
        const int grid = (int)std::min<int64_t>(n_main, plan->n_slots);
        if (has_long && !(getenv("GBX_POA_GATE") && atoi(getenv("GBX_POA_GATE")) == 0)) {
            int cus = 256, dev = 0;
            if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
            hipLaunchKernelGGL(poa_gate_kernel, dim3(1), dim3(64), 0, s, (const unsigned long long *)(A.cells + POA_LONG_STARTED),
                               (unsigned)std::min<int64_t>(plan->long_slots, cus));
        }
        Stage st("poa_window_team", s);
        if (team_form == 2) {
            PoaKernArgs KA; KA.A = A; KA.L = L;
            hipLaunchKernelGGL((poa_team2_kernel<false, GBX_POA_TEAM_WAVES>), dim3(grid), dim3(64 * POA_TEAM_NW), team_lds, s, KA, team_sync_off);
        } else
        hipLaunchKernelGGL((poa_team_kernel<false, GBX_POA_TEAM_WAVES>), dim3(grid), dim3(64 * POA_TEAM_NW), team_lds, s, A, L, team_sync_off);
    } else if (has_main) {
        const SlotLayout L = make_layout(ncap, deg, lmax, false);
        A.slot_bytes = L.total;
        // (the long launch was queued first and its wavefronts need their place on the chip: a full main grid would keep them
        // out until its first wavefronts retire, i.e. turn the long windows into a tail)
        int64_t resident = plan->n_slots;
        if (has_long && resident > 2 * (int64_t)plan->long_slots) resident -= plan->long_slots;
        const int grid = (int)std::min<int64_t>(n_windows - plan->n_long_windows, resident);
        if (has_long && !(getenv("GBX_POA_GATE") && atoi(getenv("GBX_POA_GATE")) == 0)) {
            int cus = 256, dev = 0;
            if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
            hipLaunchKernelGGL(poa_gate_kernel, dim3(1), dim3(64), 0, s, (const unsigned long long *)(A.cells + POA_LONG_STARTED),
                               (unsigned)std::min<int64_t>(plan->long_slots, cus));
        }
        Stage st("poa_window", s);
        const char *oe = getenv("GBX_POA_OCC");             // tuning aid: 2 = the instance compiled for two wavefronts per SIMD (no spills, nine ring rows)
        if (oe && atoi(oe) == 4 && serial_form == 2) {      // sixteen windows per CU with the serial phases out of line (128 VGPRs without the spills)
            PoaKernArgs KA; KA.A = A; KA.L = L;
            hipLaunchKernelGGL((poa_kernel2<false, 4, 4>), dim3(grid), dim3(64), std::max<size_t>(A.lds_marks ? lds_need : 0, (size_t)4 * POA_RING_SLOT), s, KA);
        } else if (oe && atoi(oe) == 4)   // with GBX_POA_MAX_WAVES=16: sixteen windows per CU (128 VGPRs, four ring rows; the sort's LDS arrays hold 2 352 nodes)
            hipLaunchKernelGGL((poa_kernel<false, 4, 4>), dim3(grid), dim3(64), std::max<size_t>(A.lds_marks ? lds_need : 0, (size_t)4 * POA_RING_SLOT), s, A, L);
        else if (oe && atoi(oe) == 2)
            hipLaunchKernelGGL((poa_kernel<false, 2, 9>), dim3(grid), dim3(64), std::max<size_t>(A.lds_marks ? lds_need : 0, (size_t)9 * POA_RING_SLOT), s, A, L);
        else if (serial_form == 2) {
            PoaKernArgs KA; KA.A = A; KA.L = L;
            hipLaunchKernelGGL((poa_kernel2<false>), dim3(grid), dim3(64), std::max<size_t>(A.lds_marks ? lds_need : 0, (size_t)POA_RING_BYTES), s, KA);
        } else
            hipLaunchKernelGGL((poa_kernel<false>), dim3(grid), dim3(64), std::max<size_t>(A.lds_marks ? lds_need : 0, (size_t)POA_RING_BYTES), s, A, L);
    }
    if (ss && (rc = ss->join(s))) return rc;
    GBX_HIP(hipGetLastError());
    side_lock = std::unique_lock<std::mutex>();
    GBX_GUARD_CHECK("poa");
    return GBX_OK;
}

// The wide launch (poa_wide_kernel): every window of the call on int32 cells.  Workspace = counter block | n_slots slots of
// poa_wide_slot_bytes(ncap, deg, lmax).
size_t poa_wide_slot_bytes(int ncap, int deg, int lmax) { return (size_t)make_layout(ncap, deg, lmax, true, 4).total; }
size_t poa_wide_workspace_bytes(int ncap, int deg, int lmax, int n_slots)
{
    return (size_t)POA_NCOUNTERS * 8 + 256 + poa_wide_slot_bytes(ncap, deg, lmax) * (size_t)(n_slots > 0 ? n_slots : 0);
}
int poa_launch_wide(const gbx_poa_params *p, int64_t n_windows, const int64_t *d_win_first_seq, const int64_t *d_seq_off, const int32_t *d_seq_len,
                    const uint8_t *d_arena, uint8_t *d_cons, int32_t *d_cons_len, int32_t *d_status, int64_t cons_stride,
                    int ncap, int deg, int lmax, int n_slots, void *d_work, size_t work_bytes, hipStream_t s)
{
    if (n_windows == 0) return GBX_OK;
    if (n_slots < 1 || work_bytes < poa_wide_workspace_bytes(ncap, deg, lmax, n_slots)) { set_error("poa: wide workspace too small"); return GBX_ERR_ARG; }
    if (n_windows >= ((int64_t)1 << 31)) { set_error("poa: more than 2^31 windows in one call"); return GBX_ERR_UNSUPPORTED; }
    PoaScore S = {p->m, p->n, p->g, p->e, p->q, p->c, 0};
    if (S.g > 0 || S.q > 0 || S.e > 0 || S.c > 0) { set_error("poa: gap penalties must be non-positive"); return GBX_ERR_ARG; }
    if (S.g >= S.e) { S.e = S.q = S.c = S.g; S.linear = 1; }
    else if (S.g <= S.q || S.e >= S.c) { S.q = S.g; S.c = S.e; }
    // int32 cells with -2^29 as -infinity: real scores stay far above it (|score| <= 128 x (nodes + length))
    if (((int64_t)ncap + lmax) * 130 > ((int64_t)1 << 28)) { set_error("poa: window too large even for 32-bit cells (%d nodes, length %d)", ncap, lmax); return GBX_ERR_UNSUPPORTED; }
    char *wb = (char *)d_work;
    unsigned long long *d_cells = (unsigned long long *)wb;
    GBX_HIP(hipMemsetAsync(d_cells, 0, POA_NCOUNTERS * 8, s));
    const Mat2 T = {S.e, S.g, S.q, S.c};
    PoaArgs A;
    A.n_windows = n_windows; A.win_first_seq = d_win_first_seq; A.seq_off = d_seq_off; A.seq_len = d_seq_len;
    A.arena = d_arena; A.cons = d_cons; A.cons_len = d_cons_len; A.status = d_status; A.cons_stride = cons_stride;
    A.cells = d_cells; A.ncap = ncap; A.deg = deg; A.lmax = lmax; A.S = S;
    for (int v = 0; v < 2; ++v) {
        A.Tc[v][0] = mp_pow(T, v == 0 ? 8 : 16);
        for (int k = 1; k < 4; ++k) A.Tc[v][k] = mp_mul(A.Tc[v][k - 1], A.Tc[v][k - 1]);
    }
    A.lds_marks = 0; A.lds_stack = 0; A.lds_ncap = 0;
    const SlotLayout L = make_layout(ncap, deg, lmax, true, 4);
    A.work = wb + (size_t)POA_NCOUNTERS * 8 + 256; A.slot_bytes = L.total; A.wlist = nullptr; A.cnt_idx = POA_CNT_MAIN; A.cur_idx = POA_CUR_MAIN;
    const int grid = (int)std::min<int64_t>(n_windows, n_slots);
    Stage st("poa_window_wide", s);
    hipLaunchKernelGGL(poa_wide_kernel, dim3(grid), dim3(64), (size_t)POA_LDS_FIXED + 64, s, A, L);
    GBX_HIP(hipGetLastError());
    GBX_GUARD_CHECK("poa");
    return GBX_OK;
}

}  // namespace gbx
