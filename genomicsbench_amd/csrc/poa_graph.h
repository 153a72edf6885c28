// poa_graph.h — partial-order graph state of one window and the serial graph
// algorithms of spoa (add_alignment, topological_sort, heaviest-bundle
// consensus, NW traceback), written once for device and host.
//
// On the GPU every lane of the window's wavefront runs this code with the same
// (wave-uniform) control flow; loads are broadcasts and stores are idempotent,
// so no lane election is needed.  The same header is compiled for the host by
// tests/poa_hostcheck.cpp — a TEST-ONLY build that lets the CPU suite compare
// these routines with the oracle; libgbx.so never runs them on the host.
//
// What spoa call each routine restates (spoa v3, un-vendored; call sites in
// R/benchmarks/poa/msa_spoa_omp.cpp:242,247,252; algorithm notes in
// SURVEY.md Appendix D):
//   poa_add_alignment   Graph::add_alignment(alignment, sequence, weight = 1)
//   poa_topo_sort       Graph::topological_sort
//   poa_consensus       Graph::generate_consensus -> traverse_heaviest_bundle (+ branch_completion)
//   poa_traceback       backtrack part of SisdAlignmentEngine::align (kNW, affine/convex)
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define PG_HD __host__ __device__ inline
#else
#define PG_HD inline
#endif

namespace gbx {

// DP cells are stored as int16.  spoa's kNegativeInfinity (INT32_MIN + 1024) only ever appears as an exact
// sentinel (row 0 of F/O, column 0 of E/Q) that loses every max() and fails every equality test against a
// real score; -32000 plays the same role as long as real scores stay above -30000, which the host plan
// guarantees (it rejects windows whose worst-case score could go lower).
typedef int16_t poa_cell_t;
constexpr int POA_NEG_INF = -32000;
constexpr int POA_COL0 = 7;                        // column j lives at index j + 7: column 1 is 16-byte aligned
PG_HD int poa_row_stride(int len) { return 8 + ((len + 15) / 16) * 16; }
constexpr int POA_ALN_CAP = 7;                     // aligned nodes per node (8 distinct letters per column)
constexpr int POA_ALN_STRIDE = 8;                  // row stride of the aligned-node lists (16-byte aligned rows)
struct alignas(16) PoaInt4 { int32_t v[4]; };      // one 16-byte load of four list entries

// error bits (per window)
constexpr int POA_ERR_NODES = 1;                   // node capacity exceeded
constexpr int POA_ERR_DEGREE = 2;                  // edge fan-in/out capacity exceeded
constexpr int POA_ERR_LETTERS = 4;                 // more than POA_ALN_CAP+1 letters aligned in one column
constexpr int POA_ERR_STACK = 8;                   // DFS stack capacity exceeded
constexpr int POA_ERR_CONS = 16;                   // consensus longer than the output row

struct PoaGraph {
    // capacities
    int ncap, deg, stk_cap, aln_path_cap;
    // graph (per node)
    uint8_t *code;          // [ncap] letter code
    uint8_t *in_cnt;        // [ncap]
    uint8_t *out_cnt;       // [ncap]
    uint8_t *aln_cnt;       // [ncap]
    // Edge lists, insertion order.  The first four entries of a node are "hot" ([ncap*4], 16 bytes per node,
    // neighbouring nodes share cache lines); entries 4..deg-1 are "cold" ([ncap*(deg-4)], rarely touched).
    int32_t *in_src, *in_src_x;     // source node of in-edge k
    int32_t *in_wt, *in_wt_x;       // total weight of in-edge k
    int32_t *out_dst, *out_dst_x;   // destination of out-edge k
    uint8_t *out_slot, *out_slot_x; // position of that out-edge in the destination's in-list
    int32_t *aln;           // [ncap*POA_ALN_STRIDE] aligned node ids (insertion order), 16-byte aligned rows
    int32_t *r2n, *n2r;     // [ncap] topological order
    // scratch
    uint8_t *mark, *check;  // [ncap]
    int32_t *stack;         // [stk_cap] DFS stack of the topological sort (LDS on the GPU when it fits)
    int32_t *cons_path;     // [ncap] heaviest-bundle path (consensus backtrack)
    int32_t *score, *pred;  // [ncap]
    int32_t *path_node, *path_pos;   // [aln_path_cap] alignment, stored in traceback (reverse) order
    // letters
    int16_t *coder;         // [256] letter -> code or -1
    uint8_t *decoder;       // [256]
    // counters (kept in registers by the caller between calls)
    int n_nodes, n_codes, n_path, err;
    int path_lo, path_hi;   // smallest / largest sequence position on the alignment path (set by poa_traceback)
};

#define PG_EDGE(hot, cold, n, k) ((k) < 4 ? (hot)[(int64_t)(n) * 4 + (k)] : (cold)[(int64_t)(n) * (g.deg - 4) + ((k) - 4)])
#define PG_IN_SRC(g, n, k) PG_EDGE((g).in_src, (g).in_src_x, n, k)
#define PG_IN_WT(g, n, k) PG_EDGE((g).in_wt, (g).in_wt_x, n, k)
#define PG_OUT_DST(g, n, k) PG_EDGE((g).out_dst, (g).out_dst_x, n, k)
#define PG_OUT_SLOT(g, n, k) PG_EDGE((g).out_slot, (g).out_slot_x, n, k)

PG_HD void poa_graph_reset(PoaGraph &g)
{
    g.n_nodes = 0; g.n_codes = 0; g.n_path = 0; g.err = 0;
    for (int i = 0; i < 256; ++i) g.coder[i] = -1;
}

PG_HD int poa_add_node(PoaGraph &g, int code)
{
    if (g.n_nodes >= g.ncap) { g.err |= POA_ERR_NODES; return g.ncap - 1; }
    const int id = g.n_nodes++;
    g.code[id] = (uint8_t)code;
    g.in_cnt[id] = 0; g.out_cnt[id] = 0; g.aln_cnt[id] = 0;
    return id;
}

// Graph::add_edge: bump an existing b->e edge, else append it to both adjacency lists
PG_HD void poa_add_edge(PoaGraph &g, int b, int e, int w)
{
    const int oc = g.out_cnt[b];
    for (int k = 0; k < oc; ++k)
        if (PG_OUT_DST(g, b, k) == e) { PG_IN_WT(g, e, PG_OUT_SLOT(g, b, k)) += w; return; }
    const int ic = g.in_cnt[e];
    if (oc >= g.deg || ic >= g.deg) { g.err |= POA_ERR_DEGREE; return; }
    PG_OUT_DST(g, b, oc) = e; PG_OUT_SLOT(g, b, oc) = (uint8_t)ic; g.out_cnt[b] = (uint8_t)(oc + 1);
    PG_IN_SRC(g, e, ic) = b; PG_IN_WT(g, e, ic) = w; g.in_cnt[e] = (uint8_t)(ic + 1);
}

// Graph::add_sequence: a fresh chain for seq[begin,end); returns its first node or -1
PG_HD int poa_add_chain(PoaGraph &g, const uint8_t *seq, int begin, int end)
{
    if (begin == end) return -1;
    const int first = poa_add_node(g, g.coder[seq[begin]]);
    int prev = first;
    for (int i = begin + 1; i < end; ++i) {
        const int id = poa_add_node(g, g.coder[seq[i]]);
        poa_add_edge(g, prev, id, 2);
        prev = id;
    }
    return first;
}

// Graph::topological_sort (iterative DFS; aligned nodes are emitted side by side).
// Everything a visit needs (marks, degrees, the first four in-edges and aligned nodes) is loaded up
// front, independent of each other, so a visit costs about two memory round trips on the GPU.
// Requires g.deg % 4 == 0 (16-byte aligned in-edge rows).
PG_HD void poa_topo_sort(PoaGraph &g)
{
    const int n = g.n_nodes;
    for (int i = 0; i < n; ++i) { g.mark[i] = 0; g.check[i] = 1; }
    int sp = 0, nr = 0;
    long long visits = 0;
    const long long visit_cap = (long long)n * (2 * (g.deg + POA_ALN_STRIDE) + 4) + 64;
    for (int i = 0; i < n; ++i) {
        if (g.mark[i] != 0) continue;
        g.stack[sp++] = i;
        while (sp) {
            if (++visits > visit_cap) { g.err |= POA_ERR_STACK; return; }      // (a node is pushed once per list that names it)
            const int id = g.stack[sp - 1];
            const int mk = g.mark[id], ic = g.in_cnt[id], ac = g.aln_cnt[id];
            const bool chk = g.check[id] != 0;
            const PoaInt4 e4 = *(const PoaInt4 *)(g.in_src + (int64_t)id * 4);
            const PoaInt4 a4 = *(const PoaInt4 *)(g.aln + (int64_t)id * POA_ALN_STRIDE);
            const PoaInt4 b4 = *(const PoaInt4 *)(g.aln + (int64_t)id * POA_ALN_STRIDE + 4);
            bool valid = true;
            if (mk != 2) {
                for (int k = 0; k < ic; ++k) {
                    const int b = k < 4 ? e4.v[k] : PG_IN_SRC(g, id, k);
                    if (g.mark[b] != 2) {
                        if (sp >= g.stk_cap) { g.err |= POA_ERR_STACK; return; }
                        g.stack[sp++] = b; valid = false;
                    }
                }
                if (chk) {
                    for (int k = 0; k < ac; ++k) {
                        const int a = k < 4 ? a4.v[k] : b4.v[k - 4];
                        if (g.mark[a] != 2) {
                            if (sp >= g.stk_cap) { g.err |= POA_ERR_STACK; return; }
                            g.stack[sp++] = a; g.check[a] = 0; valid = false;
                        }
                    }
                }
                if (valid) {
                    g.mark[id] = 2;
                    if (chk) {
                        g.n2r[id] = nr; g.r2n[nr++] = id;
                        for (int k = 0; k < ac; ++k) { const int a = k < 4 ? a4.v[k] : b4.v[k - 4]; g.n2r[a] = nr; g.r2n[nr++] = a; }
                    }
                } else g.mark[id] = 1;
            }
            if (valid) --sp;
        }
    }
}

// Graph::add_alignment(alignment, sequence, weight = 1).  The alignment is read from
// g.path_* in REVERSE (it was stored in traceback order); n_path == 0 means "no alignment".
PG_HD void poa_add_alignment(PoaGraph &g, const uint8_t *seq, int len)
{
    if (len == 0) return;
    for (int i = 0; i < len; ++i) {
        const int c = seq[i];
        if (g.coder[c] < 0) { g.coder[c] = (int16_t)g.n_codes; g.decoder[g.n_codes] = (uint8_t)c; ++g.n_codes; }
    }
    const int np = g.n_path;
    if (np == 0) {
        poa_add_chain(g, seq, 0, len);
        if (g.err == 0) poa_topo_sort(g);
        return;
    }
    int first_pos = -1, last_pos = -1;
    for (int t = np - 1; t >= 0; --t) {
        const int pos = g.path_pos[t];
        if (pos != -1) { if (first_pos < 0) first_pos = pos; last_pos = pos; }
    }
    const int before = g.n_nodes;
    poa_add_chain(g, seq, 0, first_pos);
    int head = before == g.n_nodes ? -1 : g.n_nodes - 1;
    const int tail = poa_add_chain(g, seq, last_pos + 1, len);
    int prev_w = head == -1 ? 0 : 1;
    for (int t = np - 1; t >= 0; --t) {
        const int pos = g.path_pos[t];
        if (pos == -1) continue;
        if (g.err) return;                                   // a capacity was exceeded: the window is abandoned
        const int node = g.path_node[t];
        const int code = g.coder[seq[pos]];
        int id;
        if (node == -1) {
            id = poa_add_node(g, code);
        } else if (g.code[node] == code) {
            id = node;
        } else {
            int found = -1;
            const int ac = g.aln_cnt[node];
            for (int k = 0; k < ac; ++k) {
                const int a = g.aln[node * POA_ALN_STRIDE + k];
                if (g.code[a] == code) { found = a; break; }
            }
            if (found == -1) {
                id = poa_add_node(g, code);
                if (ac + 1 > POA_ALN_CAP) { g.err |= POA_ERR_LETTERS; }
                else {
                    for (int k = 0; k < ac; ++k) {
                        const int a = g.aln[node * POA_ALN_STRIDE + k];
                        g.aln[id * POA_ALN_STRIDE + g.aln_cnt[id]] = a; g.aln_cnt[id] = (uint8_t)(g.aln_cnt[id] + 1);
                        g.aln[a * POA_ALN_STRIDE + g.aln_cnt[a]] = id; g.aln_cnt[a] = (uint8_t)(g.aln_cnt[a] + 1);
                    }
                    g.aln[id * POA_ALN_STRIDE + g.aln_cnt[id]] = node; g.aln_cnt[id] = (uint8_t)(g.aln_cnt[id] + 1);
                    g.aln[node * POA_ALN_STRIDE + ac] = id; g.aln_cnt[node] = (uint8_t)(ac + 1);
                }
            } else id = found;
        }
        if (head != -1) poa_add_edge(g, head, id, prev_w + 1);
        head = id;
        prev_w = 1;
    }
    if (tail != -1) poa_add_edge(g, head, tail, prev_w + 1);
    if (g.err) return;
    poa_topo_sort(g);
}

// Graph::branch_completion
PG_HD int poa_branch_completion(PoaGraph &g, int rank)
{
    const int node_id = g.r2n[rank];
    const int oc = g.out_cnt[node_id];
    for (int k = 0; k < oc; ++k) {
        const int t = PG_OUT_DST(g, node_id, k);
        const int ic = g.in_cnt[t];
        for (int z = 0; z < ic; ++z) {
            const int b = PG_IN_SRC(g, t, z);
            if (b != node_id) g.score[b] = -1;
        }
    }
    int max_score = 0, max_id = 0;
    for (int i = rank + 1; i < g.n_nodes; ++i) {
        const int id = g.r2n[i];
        int sc = -1, pr = -1;
        const int ic = g.in_cnt[id];
        for (int k = 0; k < ic; ++k) {
            const int b = PG_IN_SRC(g, id, k), w = PG_IN_WT(g, id, k);
            if (g.score[b] == -1) continue;
            if (sc < w || (sc == w && g.score[pr] <= g.score[b])) { sc = w; pr = b; }
        }
        if (pr != -1) sc += g.score[pr];
        g.score[id] = sc; g.pred[id] = pr;
        if (max_score < sc) { max_score = sc; max_id = id; }
    }
    return max_id;
}

// Graph::traverse_heaviest_bundle + generate_consensus; returns the consensus length (<= cap written)
PG_HD int poa_consensus(PoaGraph &g, uint8_t *out, int cap)
{
    const int n = g.n_nodes;
    if (n == 0) return 0;
    int max_id = 0;
    for (int i = 0; i < n; ++i) { g.score[i] = -1; g.pred[i] = -1; }
    for (int r = 0; r < n; ++r) {
        const int id = g.r2n[r];
        int sc = -1, pr = -1;
        const int ic = g.in_cnt[id];
        for (int k = 0; k < ic; ++k) {
            const int b = PG_IN_SRC(g, id, k), w = PG_IN_WT(g, id, k);
            if (sc < w || (sc == w && g.score[pr] <= g.score[b])) { sc = w; pr = b; }
        }
        if (pr != -1) sc += g.score[pr];
        g.score[id] = sc; g.pred[id] = pr;
        if (g.score[max_id] < sc) max_id = id;
    }
    // (every completion moves to a node of higher rank: at most n_nodes of them; the bound only matters on a corrupted graph)
    for (int guard = 0; g.out_cnt[max_id] != 0; ++guard) {
        if (guard > g.n_nodes) { g.err |= POA_ERR_STACK; break; }
        max_id = poa_branch_completion(g, g.n2r[max_id]);
    }
    // backtrack, then emit reversed
    int len = 0;
    while (g.pred[max_id] != -1) { g.cons_path[len++] = max_id; max_id = g.pred[max_id]; if (len >= g.ncap) { g.err |= POA_ERR_STACK; break; } }
    g.cons_path[len++] = max_id;
    if (len > cap) g.err |= POA_ERR_CONS;
    for (int k = 0; k < len && k < cap; ++k) out[k] = g.decoder[g.code[g.cons_path[len - 1 - k]]];
    return len;
}

// DP matrices of one alignment: (n_nodes+1) rows x W = len+1 columns, row-major
// Cell = int16 (the device's fast paths; real scores above -30000, checked by the host plan) or int32 (the wide path: windows
// whose scores may leave that range - long reads - as spoa falls back to 32-bit lanes); `neg` = the -infinity sentinel of each.
template <class Cell> struct PoaMatricesT {
    typedef Cell cell;
    static constexpr int neg = sizeof(Cell) == 2 ? POA_NEG_INF : -(1 << 29);
    Cell *H, *F, *E, *O, *Q;
    int Wp;                                       // row stride in cells (poa_row_stride)
};
typedef PoaMatricesT<poa_cell_t> PoaMatrices;
typedef PoaMatricesT<int32_t> PoaMatricesW;

// linear: spoa's LINEAR subtype (g >= e at createAlignmentEngine: one gap cost g, no gap states).  Its score matrix is the affine
// one with e = q = c = g - the launch sets them so and the DP runs unchanged - but its backtrack takes one cell per step and
// never walks an extension (a vertical or horizontal candidate is H + g only), which the tracebacks honour.
struct PoaScore { int m, n, g, e, q, c, linear; };

// Row descriptors of the current topological order, indexed by rank r (DP row r+1); stored in
// g.score / g.pred, which are free until the consensus:
//   rd_pred[r] = DP row of the node's first in-edge source (0 = the virtual start row)
//   rd_info[r] = letter | in-degree << 8 | sink << 16
// and, only for the DP (the traceback overwrites them): DP rows of the 2nd and 3rd in-edge source in
// g.path_node[r] / g.path_pos[r].
PG_HD void poa_rowdesc_one(PoaGraph &g, int r)
{
    const int node = g.r2n[r];
    const int ic = g.in_cnt[node];
    g.score[r] = ic ? g.n2r[PG_IN_SRC(g, node, 0)] + 1 : 0;
    g.pred[r] = (int)g.decoder[g.code[node]] | (ic << 8) | ((g.out_cnt[node] == 0) << 16);
    g.path_node[r] = ic > 1 ? g.n2r[PG_IN_SRC(g, node, 1)] + 1 : 0;
    g.path_pos[r] = ic > 2 ? g.n2r[PG_IN_SRC(g, node, 2)] + 1 : 0;
}

// Backtrack of SisdAlignmentEngine::align for kNW with affine/convex gaps; fills g.path_* in
// traceback order (the reference reverses it afterwards; poa_add_alignment reads it backwards).
// Needs the row descriptors.  All candidate cells of a step are read up front (independent loads),
// then the reference's priority order (diagonal, vertical F/H/O/H, horizontal E/H/Q/H) is applied.
template <class MT> PG_HD void poa_traceback(PoaGraph &g, const MT &M, const PoaScore &S, const uint8_t *seq, int max_i, int max_j)
{
    g.n_path = 0;
    if (max_i == -1 && max_j == -1) return;
    const int Wp = M.Wp;
    const int32_t *rd_pred = g.score, *rd_info = g.pred;
    int i = max_i, j = max_j, prev_i = 0, prev_j = 0, np = 0;
#define PG_AT(A, a, b) ((int)(A)[(int64_t)(a) * Wp + (b) + POA_COL0])
    int plo = -1, phi = -1;                      // positions come out in descending order
#define PG_PUSH(nd, ps) do { const int ps_ = (ps); if (np < g.aln_path_cap) { g.path_node[np] = (nd); g.path_pos[np] = ps_; } \
                             if (ps_ != -1) { if (phi < 0) phi = ps_; plo = ps_; } ++np; } while (0)
    while (!(i == 0 && j == 0)) {
        const int Hij = PG_AT(M.H, i, j);
        bool found = false, ext_left = false, ext_up = false;
        int node = -1, ic = 0, p0 = 0;
        if (i != 0) { p0 = rd_pred[i - 1]; const int info = rd_info[i - 1]; ic = (info >> 8) & 0xff; node = g.r2n[i - 1];
            if (j != 0) {
                const int mc = (info & 0xff) == seq[j - 1] ? S.m : S.n;
                for (int p = 0; p < (ic ? ic : 1) && !found; ++p) {
                    const int pi = p ? g.n2r[PG_IN_SRC(g, node, p)] + 1 : p0;
                    if (Hij == PG_AT(M.H, pi, j - 1) + mc) { prev_i = pi; prev_j = j - 1; found = true; }
                }
            }
            if (!found) {
                for (int p = 0; p < (ic ? ic : 1) && !found; ++p) {
                    const int pi = p ? g.n2r[PG_IN_SRC(g, node, p)] + 1 : p0;
                    const int fv = PG_AT(M.F, pi, j), hv = PG_AT(M.H, pi, j), ov = PG_AT(M.O, pi, j);
                    const bool c1 = !S.linear && Hij == fv + S.e;
                    const bool c2 = !c1 && Hij == hv + S.g;
                    const bool c3 = !S.linear && !c1 && !c2 && Hij == ov + S.c;
                    const bool c4 = !c1 && !c2 && !c3 && Hij == hv + S.q;
                    ext_up = ext_up || c1 || c3;
                    if (c1 || c2 || c3 || c4) { prev_i = pi; prev_j = j; found = true; }
                }
            }
        }
        if (!found && j != 0) {
            const int ev = PG_AT(M.E, i, j - 1), hv = PG_AT(M.H, i, j - 1), qv = PG_AT(M.Q, i, j - 1);
            const bool c1 = !S.linear && Hij == ev + S.e;
            const bool c2 = !c1 && Hij == hv + S.g;
            const bool c3 = !S.linear && !c1 && !c2 && Hij == qv + S.c;
            const bool c4 = !c1 && !c2 && !c3 && Hij == hv + S.q;
            ext_left = c1 || c3;
            if (c1 || c2 || c3 || c4) { prev_i = i; prev_j = j - 1; found = true; }
        }
        PG_PUSH(i == prev_i ? -1 : node, j == prev_j ? -1 : j - 1);
        i = prev_i; j = prev_j;
        if (ext_left) {
            for (;;) {
                PG_PUSH(-1, j - 1);
                --j;
                if (j <= 0) break;                           // column 0 ends every horizontal run (E = Q = -infinity there); never below it
                if (PG_AT(M.E, i, j) + S.e != PG_AT(M.E, i, j + 1) && PG_AT(M.Q, i, j) + S.c != PG_AT(M.Q, i, j + 1)) break;
            }
        } else if (ext_up) {
            for (int guard = 0;; ++guard) {
                if (guard > g.n_nodes) { g.err |= POA_ERR_STACK; break; }     // (rows strictly decrease on a sound matrix)
                bool stop = false;
                prev_i = 0;
                const int nd = g.r2n[i - 1];
                const int icu = (rd_info[i - 1] >> 8) & 0xff;
                const int fij = PG_AT(M.F, i, j), oij = PG_AT(M.O, i, j);
                for (int p = 0; p < icu; ++p) {
                    const int pi = p ? g.n2r[PG_IN_SRC(g, nd, p)] + 1 : rd_pred[i - 1];
                    const int hv = PG_AT(M.H, pi, j);
                    const bool s1 = fij == hv + S.g;
                    const bool s2 = !s1 && fij == PG_AT(M.F, pi, j) + S.e;
                    const bool s3 = !s1 && !s2 && oij == hv + S.q;
                    const bool s4 = !s1 && !s2 && !s3 && oij == PG_AT(M.O, pi, j) + S.c;
                    // `stop = c1 || .. || (stop = c3) || ..`: the last assignment evaluated wins
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
    g.n_path = np <= g.aln_path_cap ? np : 0;
    g.path_lo = plo; g.path_hi = phi;
}

}  // namespace gbx
