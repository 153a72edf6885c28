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
    unsigned long long *cells;        // DP cells (graph nodes x sequence length, summed over alignments)
    int ncap, deg, lmax;
    PoaScore S;
    Mat2 Tc1, Tc2, Tc4, Tc8;          // (T^CPL)^(1,2,4,8): uniform factors of the row_shr scan steps
};

// byte offsets of the arrays inside one workspace slot
struct SlotLayout {
    int64_t code, in_cnt, out_cnt, aln_cnt, out_slot, mark, check, decoder, coder;
    int64_t in_src, in_wt, out_dst, aln, r2n, n2r, stack, score, pred, path_node, path_pos, mat, total;
    int stk_cap, path_cap;
};

__host__ __device__ inline int64_t align_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }

__host__ __device__ inline SlotLayout make_layout(int ncap, int deg, int lmax)
{
    SlotLayout L;
    int64_t o = 0;
    auto take = [&](int64_t bytes) { int64_t r = o; o = align_up(o + bytes, 16); return r; };
    L.stk_cap = ncap * 4 + 64;
    L.path_cap = ncap + lmax + 8;
    L.code = take(ncap); L.in_cnt = take(ncap); L.out_cnt = take(ncap); L.aln_cnt = take(ncap);
    L.out_slot = take((int64_t)ncap * deg); L.mark = take(ncap); L.check = take(ncap);
    L.decoder = take(256); L.coder = take(512);
    L.in_src = take((int64_t)ncap * deg * 4); L.in_wt = take((int64_t)ncap * deg * 4);
    L.out_dst = take((int64_t)ncap * deg * 4); L.aln = take((int64_t)ncap * POA_ALN_CAP * 4);
    L.r2n = take((int64_t)ncap * 4); L.n2r = take((int64_t)ncap * 4);
    L.stack = take((int64_t)L.stk_cap * 4); L.score = take((int64_t)ncap * 4); L.pred = take((int64_t)ncap * 4);
    L.path_node = take((int64_t)L.path_cap * 4); L.path_pos = take((int64_t)L.path_cap * 4);
    L.mat = take((int64_t)(ncap + 1) * (lmax + 1) * 5 * 4);
    L.total = align_up(o, 256);
    return L;
}

// ---- DP of one sequence against the graph: fills M, returns the best sink cell -------------
template <int CPL>
__device__ void poa_dp(const PoaGraph &g, const PoaMatrices &M, const PoaArgs &A, const uint8_t *seq, int len,
                       int &max_i, int &max_j)
{
    const int lane = threadIdx.x & 63;
    const PoaScore S = A.S;
    const int W = M.W;
    const int n = g.n_nodes;
    constexpr int BLK = 64 * CPL;
    // lane-dependent max-plus matrices: Tc^(lane&15 + 1), Tc^(lane&31 + 1), Tc^lane
    const Mat2 P16 = mp_pow(A.Tc1, (lane & 15) + 1);
    const Mat2 P32 = mp_pow(A.Tc1, (lane & 31) + 1);
    const Mat2 PC = mp_pow(A.Tc1, lane);

    // row 0 (sisd_alignment_engine `initialize`)
    for (int j = lane; j < W; j += 64) {
        const int e0 = j == 0 ? 0 : S.g + (j - 1) * S.e, q0 = j == 0 ? 0 : S.q + (j - 1) * S.c;
        M.E[j] = e0; M.Q[j] = q0;
        M.F[j] = j == 0 ? 0 : POA_NEG_INF; M.O[j] = j == 0 ? 0 : POA_NEG_INF;
        M.H[j] = j == 0 ? 0 : max(q0, e0);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");

    const bool single = len <= BLK;
    int sq[CPL];
    auto load_seq = [&](int base) {
#pragma unroll
        for (int c = 0; c < CPL; ++c) { const int j = base + lane * CPL + 1 + c; sq[c] = j <= len ? seq[j - 1] : -1; }
    };
    if (single) load_seq(0);

    int best = POA_NEG_INF;
    max_i = -1; max_j = -1;
    for (int r = 0; r < n; ++r) {
        const int node = g.r2n[r], i = r + 1;
        const int ic = g.in_cnt[node];
        const int64_t ro = (int64_t)i * W;
        // column 0
        int po = ic == 0 ? S.q - S.c : POA_NEG_INF, pf = ic == 0 ? S.g - S.e : POA_NEG_INF;
        for (int k = 0; k < ic; ++k) {
            const int64_t pi = (int64_t)(g.n2r[g.in_src[node * g.deg + k]] + 1) * W;
            po = max(po, M.O[pi]); pf = max(pf, M.F[pi]);
        }
        const int O0 = po + S.c, F0 = pf + S.e, H0 = max(O0, F0);
        if (lane == 0) { M.O[ro] = O0; M.F[ro] = F0; M.H[ro] = H0; M.E[ro] = POA_NEG_INF; M.Q[ro] = POA_NEG_INF; }
        const int letter = g.decoder[g.code[node]];
        // (E,Q) at the first column of the block; column 1: E = H0+g, Q = H0+q (E[0] = Q[0] = -inf)
        int cE = H0 + S.g, cQ = H0 + S.q;
        for (int base = 0; base < len; base += BLK) {
            const int j0 = base + lane * CPL + 1;                  // lane's first column
            if (!single) load_seq(base);
            int Fa[CPL], Oa[CPL], Ha[CPL];
            for (int p = 0; p < (ic ? ic : 1); ++p) {
                const int64_t po_ = ic ? (int64_t)(g.n2r[g.in_src[node * g.deg + p]] + 1) * W : 0;
                const int32_t *Hp = M.H + po_, *Fp = M.F + po_, *Op = M.O + po_;
                int hl = j0 - 1 <= len ? Hp[j0 - 1] : 0;           // H(pred, j-1) for the lane's first column
#pragma unroll
                for (int c = 0; c < CPL; ++c) {
                    const int j = j0 + c;
                    const bool in = j <= len;
                    const int hp = in ? Hp[j] : 0, fp = in ? Fp[j] : 0, op = in ? Op[j] : 0;
                    const int sc = sq[c] == letter ? S.m : S.n;
                    const int f = max(hp + S.g, fp + S.e);
                    const int o = max(hp + S.q, op + S.c);
                    const int h = hl + sc;
                    hl = hp;
                    if (p == 0) { Fa[c] = f; Oa[c] = o; Ha[c] = h; }
                    else { Fa[c] = max(Fa[c], f); Oa[c] = max(Oa[c], o); Ha[c] = max(Ha[c], h); }
                }
            }
            int Aa[CPL];
#pragma unroll
            for (int c = 0; c < CPL; ++c) Aa[c] = max(Ha[c], max(Fa[c], Oa[c]));
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
            mp_apply(A.Tc1, dpp_i<0x111>(SNEG, xE), dpp_i<0x111>(SNEG, xQ), tE, tQ); xE = max(xE, tE); xQ = max(xQ, tQ);
            mp_apply(A.Tc2, dpp_i<0x112>(SNEG, xE), dpp_i<0x112>(SNEG, xQ), tE, tQ); xE = max(xE, tE); xQ = max(xQ, tQ);
            mp_apply(A.Tc4, dpp_i<0x114>(SNEG, xE), dpp_i<0x114>(SNEG, xQ), tE, tQ); xE = max(xE, tE); xQ = max(xQ, tQ);
            mp_apply(A.Tc8, dpp_i<0x118>(SNEG, xE), dpp_i<0x118>(SNEG, xQ), tE, tQ); xE = max(xE, tE); xQ = max(xQ, tQ);
            mp_apply(P16, dpp_i<0x142, 0xa>(SNEG, xE), dpp_i<0x142, 0xa>(SNEG, xQ), tE, tQ); xE = max(xE, tE); xQ = max(xQ, tQ);
            mp_apply(P32, dpp_i<0x143, 0xc>(SNEG, xE), dpp_i<0x143, 0xc>(SNEG, xQ), tE, tQ); xE = max(xE, tE); xQ = max(xQ, tQ);
            // (E,Q) entering this lane = previous lane's inclusive value (+) Tc^lane (x) block carry
            int vE = dpp_i<0x138>(SNEG, xE), vQ = dpp_i<0x138>(SNEG, xQ);
            if (lane == 0) { vE = SNEG; vQ = SNEG; }
            mp_apply(PC, cE, cQ, tE, tQ);
            vE = max(vE, tE); vQ = max(vQ, tQ);
            // carry for the next block: (E,Q) leaving lane 63
            {
                int oE, oQ;
                mp_apply(A.Tc1, tE, tQ, oE, oQ);               // Tc^(lane+1) (x) carry
                const int lE = max(xE, oE), lQ = max(xQ, oQ);
                cE = __builtin_amdgcn_readlane(lE, 63); cQ = __builtin_amdgcn_readlane(lQ, 63);
            }
            // ---- pass 2: exact E, Q, H of the lane's columns; store the row
#pragma unroll
            for (int c = 0; c < CPL; ++c) {
                const int j = j0 + c;
                const int h = max(Aa[c], max(vE, vQ));
                if (j <= len) {
                    M.H[ro + j] = h; M.F[ro + j] = Fa[c]; M.O[ro + j] = Oa[c]; M.E[ro + j] = vE; M.Q[ro + j] = vQ;
                }
                const int ne = max(h + S.g, vE + S.e), nq = max(h + S.q, vQ + S.c);
                vE = ne; vQ = nq;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        if (g.out_cnt[node] == 0) {                                // NW: best sink at the last column
            const int v = M.H[ro + len];
            if (best < v) { best = v; max_i = i; max_j = len; }
        }
    }
}

template <int CPL>
__global__ void __launch_bounds__(64) poa_kernel(PoaArgs A, SlotLayout L)
{
    char *slot = A.work + (int64_t)blockIdx.x * A.slot_bytes;
    PoaGraph g;
    g.ncap = A.ncap; g.deg = A.deg; g.stk_cap = L.stk_cap; g.aln_path_cap = L.path_cap;
    g.code = (uint8_t *)(slot + L.code); g.in_cnt = (uint8_t *)(slot + L.in_cnt);
    g.out_cnt = (uint8_t *)(slot + L.out_cnt); g.aln_cnt = (uint8_t *)(slot + L.aln_cnt);
    g.out_slot = (uint8_t *)(slot + L.out_slot); g.mark = (uint8_t *)(slot + L.mark); g.check = (uint8_t *)(slot + L.check);
    g.decoder = (uint8_t *)(slot + L.decoder); g.coder = (int16_t *)(slot + L.coder);
    g.in_src = (int32_t *)(slot + L.in_src); g.in_wt = (int32_t *)(slot + L.in_wt);
    g.out_dst = (int32_t *)(slot + L.out_dst); g.aln = (int32_t *)(slot + L.aln);
    g.r2n = (int32_t *)(slot + L.r2n); g.n2r = (int32_t *)(slot + L.n2r);
    g.stack = (int32_t *)(slot + L.stack); g.score = (int32_t *)(slot + L.score); g.pred = (int32_t *)(slot + L.pred);
    g.path_node = (int32_t *)(slot + L.path_node); g.path_pos = (int32_t *)(slot + L.path_pos);
    int32_t *mat = (int32_t *)(slot + L.mat);

    unsigned long long cells = 0;
    for (int64_t w = blockIdx.x; w < A.n_windows; w += gridDim.x) {
        poa_graph_reset(g);
        const int64_t s0 = A.win_first_seq[w], s1 = A.win_first_seq[w + 1];
        for (int64_t s = s0; s < s1; ++s) {
            const uint8_t *seq = A.arena + A.seq_off[s];
            const int len = A.seq_len[s];
            g.n_path = 0;
            if (g.n_nodes != 0 && len != 0 && g.err == 0) {
                const int64_t plane = (int64_t)(g.n_nodes + 1) * (len + 1);
                PoaMatrices M = {mat, mat + plane, mat + 2 * plane, mat + 3 * plane, mat + 4 * plane, len + 1};
                int mi, mj;
                cells += (unsigned long long)g.n_nodes * (unsigned long long)len;
                poa_dp<CPL>(g, M, A, seq, len, mi, mj);
                poa_traceback(g, M, A.S, seq, mi, mj);
            }
            if (g.err == 0) poa_add_alignment(g, seq, len);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
        int clen = 0;
        if (g.err == 0) clen = poa_consensus(g, A.cons + w * A.cons_stride, (int)A.cons_stride);
        if ((threadIdx.x & 63) == 0) { A.cons_len[w] = clen; A.status[w] = g.err; }
    }
    if ((threadIdx.x & 63) == 0) atomicAdd(A.cells, cells);
}

}  // namespace

// workspace = slots * slot_bytes
size_t poa_slot_bytes(int ncap, int deg, int lmax) { return (size_t)make_layout(ncap, deg, lmax).total; }

int poa_read_cells(const void *d_work, size_t slots_bytes, int64_t *cells, hipStream_t s)
{
    unsigned long long v = 0;
    GBX_HIP(hipMemcpyAsync(&v, (const char *)d_work + slots_bytes, sizeof(v), hipMemcpyDeviceToHost, s));
    GBX_HIP(hipStreamSynchronize(s));
    *cells = (int64_t)v;
    return GBX_OK;
}

int poa_launch(const gbx_poa_params *p, int64_t n_windows, const int64_t *d_win_first_seq, const int64_t *d_seq_off,
               const int32_t *d_seq_len, const uint8_t *d_arena, int lmax, int deg, int ncap, int n_slots,
               uint8_t *d_cons, int32_t *d_cons_len, int32_t *d_status, int64_t cons_stride,
               void *d_work, size_t work_bytes, hipStream_t s)
{
    if (n_windows == 0) return GBX_OK;
    PoaScore S = {p->m, p->n, p->g, p->e, p->q, p->c};
    if (S.g > 0 || S.q > 0 || S.e > 0 || S.c > 0) { set_error("poa: gap penalties must be non-positive"); return GBX_ERR_ARG; }
    if (S.g >= S.e) { set_error("poa: linear gap mode (g >= e) is not supported by the device path"); return GBX_ERR_UNSUPPORTED; }
    if (S.g <= S.q || S.e >= S.c) { S.q = S.g; S.c = S.e; }          // affine == convex with both pieces equal
    const SlotLayout L = make_layout(ncap, deg, lmax);
    if (work_bytes < (size_t)L.total * (size_t)n_slots + 64) { set_error("poa: workspace too small"); return GBX_ERR_ARG; }
    unsigned long long *d_cells = (unsigned long long *)((char *)d_work + (size_t)L.total * (size_t)n_slots);
    GBX_HIP(hipMemsetAsync(d_cells, 0, 8, s));
    int cpl = lmax <= 256 ? 4 : lmax <= 512 ? 8 : lmax <= 768 ? 12 : 16;
    const Mat2 T = {S.e, S.g, S.q, S.c};
    const Mat2 Tc = mp_pow(T, cpl);
    PoaArgs A;
    A.n_windows = n_windows; A.win_first_seq = d_win_first_seq; A.seq_off = d_seq_off; A.seq_len = d_seq_len;
    A.arena = d_arena; A.cons = d_cons; A.cons_len = d_cons_len; A.status = d_status; A.cons_stride = cons_stride;
    A.work = (char *)d_work; A.slot_bytes = L.total; A.cells = d_cells; A.ncap = ncap; A.deg = deg; A.lmax = lmax; A.S = S;
    A.Tc1 = Tc; A.Tc2 = mp_mul(Tc, Tc); A.Tc4 = mp_mul(A.Tc2, A.Tc2); A.Tc8 = mp_mul(A.Tc4, A.Tc4);
    const int grid = (int)std::min<int64_t>(n_windows, n_slots);
    Stage st("poa_window", s);
    switch (cpl) {
    case 4: hipLaunchKernelGGL(poa_kernel<4>, dim3(grid), dim3(64), 0, s, A, L); break;
    case 8: hipLaunchKernelGGL(poa_kernel<8>, dim3(grid), dim3(64), 0, s, A, L); break;
    case 12: hipLaunchKernelGGL(poa_kernel<12>, dim3(grid), dim3(64), 0, s, A, L); break;
    default: hipLaunchKernelGGL(poa_kernel<16>, dim3(grid), dim3(64), 0, s, A, L); break;
    }
    GBX_HIP(hipGetLastError());
    return GBX_OK;
}

}  // namespace gbx
