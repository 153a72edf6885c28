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

constexpr int POA_NEG_INF = INT32_MIN + 1024;      // spoa kNegativeInfinity
constexpr int POA_ALN_CAP = 7;                     // aligned nodes per node (8 distinct letters per column)

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
    int32_t *in_src;        // [ncap*deg] source node of in-edge k (insertion order)
    int32_t *in_wt;         // [ncap*deg] total weight of in-edge k
    int32_t *out_dst;       // [ncap*deg] destination of out-edge k (insertion order)
    uint8_t *out_slot;      // [ncap*deg] position of that edge in the destination's in-list
    int32_t *aln;           // [ncap*POA_ALN_CAP] aligned node ids (insertion order)
    int32_t *r2n, *n2r;     // [ncap] topological order
    // scratch
    uint8_t *mark, *check;  // [ncap]
    int32_t *stack;         // [stk_cap]
    int32_t *score, *pred;  // [ncap]
    int32_t *path_node, *path_pos;   // [aln_path_cap] alignment, stored in traceback (reverse) order
    // letters
    int16_t *coder;         // [256] letter -> code or -1
    uint8_t *decoder;       // [256]
    // counters (kept in registers by the caller between calls)
    int n_nodes, n_codes, n_path, err;
};

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
        if (g.out_dst[b * g.deg + k] == e) { g.in_wt[e * g.deg + g.out_slot[b * g.deg + k]] += w; return; }
    const int ic = g.in_cnt[e];
    if (oc >= g.deg || ic >= g.deg) { g.err |= POA_ERR_DEGREE; return; }
    g.out_dst[b * g.deg + oc] = e; g.out_slot[b * g.deg + oc] = (uint8_t)ic; g.out_cnt[b] = (uint8_t)(oc + 1);
    g.in_src[e * g.deg + ic] = b; g.in_wt[e * g.deg + ic] = w; g.in_cnt[e] = (uint8_t)(ic + 1);
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

// Graph::topological_sort (iterative DFS; aligned nodes are emitted side by side)
PG_HD void poa_topo_sort(PoaGraph &g)
{
    const int n = g.n_nodes;
    for (int i = 0; i < n; ++i) { g.mark[i] = 0; g.check[i] = 1; }
    int sp = 0, nr = 0;
    for (int i = 0; i < n; ++i) {
        if (g.mark[i] != 0) continue;
        g.stack[sp++] = i;
        while (sp) {
            const int id = g.stack[sp - 1];
            bool valid = true;
            if (g.mark[id] != 2) {
                const int ic = g.in_cnt[id];
                for (int k = 0; k < ic; ++k) {
                    const int b = g.in_src[id * g.deg + k];
                    if (g.mark[b] != 2) {
                        if (sp >= g.stk_cap) { g.err |= POA_ERR_STACK; return; }
                        g.stack[sp++] = b; valid = false;
                    }
                }
                const bool chk = g.check[id] != 0;
                const int ac = g.aln_cnt[id];
                if (chk) {
                    for (int k = 0; k < ac; ++k) {
                        const int a = g.aln[id * POA_ALN_CAP + k];
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
                        for (int k = 0; k < ac; ++k) { const int a = g.aln[id * POA_ALN_CAP + k]; g.n2r[a] = nr; g.r2n[nr++] = a; }
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
                const int a = g.aln[node * POA_ALN_CAP + k];
                if (g.code[a] == code) { found = a; break; }
            }
            if (found == -1) {
                id = poa_add_node(g, code);
                if (ac + 1 > POA_ALN_CAP) { g.err |= POA_ERR_LETTERS; }
                else {
                    for (int k = 0; k < ac; ++k) {
                        const int a = g.aln[node * POA_ALN_CAP + k];
                        g.aln[id * POA_ALN_CAP + g.aln_cnt[id]] = a; g.aln_cnt[id] = (uint8_t)(g.aln_cnt[id] + 1);
                        g.aln[a * POA_ALN_CAP + g.aln_cnt[a]] = id; g.aln_cnt[a] = (uint8_t)(g.aln_cnt[a] + 1);
                    }
                    g.aln[id * POA_ALN_CAP + g.aln_cnt[id]] = node; g.aln_cnt[id] = (uint8_t)(g.aln_cnt[id] + 1);
                    g.aln[node * POA_ALN_CAP + ac] = id; g.aln_cnt[node] = (uint8_t)(ac + 1);
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
        const int t = g.out_dst[node_id * g.deg + k];
        const int ic = g.in_cnt[t];
        for (int z = 0; z < ic; ++z) {
            const int b = g.in_src[t * g.deg + z];
            if (b != node_id) g.score[b] = -1;
        }
    }
    int max_score = 0, max_id = 0;
    for (int i = rank + 1; i < g.n_nodes; ++i) {
        const int id = g.r2n[i];
        int sc = -1, pr = -1;
        const int ic = g.in_cnt[id];
        for (int k = 0; k < ic; ++k) {
            const int b = g.in_src[id * g.deg + k], w = g.in_wt[id * g.deg + k];
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
            const int b = g.in_src[id * g.deg + k], w = g.in_wt[id * g.deg + k];
            if (sc < w || (sc == w && g.score[pr] <= g.score[b])) { sc = w; pr = b; }
        }
        if (pr != -1) sc += g.score[pr];
        g.score[id] = sc; g.pred[id] = pr;
        if (g.score[max_id] < sc) max_id = id;
    }
    while (g.out_cnt[max_id] != 0) max_id = poa_branch_completion(g, g.n2r[max_id]);
    // backtrack into the stack array, then emit reversed
    int len = 0;
    while (g.pred[max_id] != -1) { g.stack[len++] = max_id; max_id = g.pred[max_id]; if (len >= g.stk_cap) { g.err |= POA_ERR_STACK; break; } }
    g.stack[len++] = max_id;
    if (len > cap) g.err |= POA_ERR_CONS;
    for (int k = 0; k < len && k < cap; ++k) out[k] = g.decoder[g.code[g.stack[len - 1 - k]]];
    return len;
}

// DP matrices of one alignment: (n_nodes+1) rows x W = len+1 columns, row-major
struct PoaMatrices {
    int32_t *H, *F, *E, *O, *Q;
    int W;
};

struct PoaScore { int m, n, g, e, q, c; };

// Backtrack of SisdAlignmentEngine::align for kNW with affine/convex gaps; fills g.path_* in
// traceback order (the reference reverses it afterwards; poa_add_alignment reads it backwards).
PG_HD void poa_traceback(PoaGraph &g, const PoaMatrices &M, const PoaScore &S, const uint8_t *seq, int max_i, int max_j)
{
    g.n_path = 0;
    if (max_i == -1 && max_j == -1) return;
    const int W = M.W;
    int i = max_i, j = max_j, prev_i = 0, prev_j = 0, np = 0;
#define PG_AT(A, a, b) (A)[(int64_t)(a) * W + (b)]
#define PG_PUSH(nd, ps) do { if (np < g.aln_path_cap) { g.path_node[np] = (nd); g.path_pos[np] = (ps); } ++np; } while (0)
    while (!(i == 0 && j == 0)) {
        const int Hij = PG_AT(M.H, i, j);
        bool found = false, ext_left = false, ext_up = false;
        if (i != 0 && j != 0) {
            const int node = g.r2n[i - 1];
            const int mc = g.decoder[g.code[node]] == seq[j - 1] ? S.m : S.n;
            const int ic = g.in_cnt[node];
            for (int p = 0; p < (ic ? ic : 1) && !found; ++p) {
                const int pi = ic ? g.n2r[g.in_src[node * g.deg + p]] + 1 : 0;
                if (Hij == PG_AT(M.H, pi, j - 1) + mc) { prev_i = pi; prev_j = j - 1; found = true; }
            }
        }
        if (!found && i != 0) {
            const int node = g.r2n[i - 1];
            const int ic = g.in_cnt[node];
            for (int p = 0; p < (ic ? ic : 1) && !found; ++p) {
                const int pi = ic ? g.n2r[g.in_src[node * g.deg + p]] + 1 : 0;
                const bool c1 = Hij == PG_AT(M.F, pi, j) + S.e;
                const bool c2 = !c1 && Hij == PG_AT(M.H, pi, j) + S.g;
                const bool c3 = !c1 && !c2 && Hij == PG_AT(M.O, pi, j) + S.c;
                const bool c4 = !c1 && !c2 && !c3 && Hij == PG_AT(M.H, pi, j) + S.q;
                ext_up = ext_up || c1 || c3;
                if (c1 || c2 || c3 || c4) { prev_i = pi; prev_j = j; found = true; }
            }
        }
        if (!found && j != 0) {
            const bool c1 = Hij == PG_AT(M.E, i, j - 1) + S.e;
            const bool c2 = !c1 && Hij == PG_AT(M.H, i, j - 1) + S.g;
            const bool c3 = !c1 && !c2 && Hij == PG_AT(M.Q, i, j - 1) + S.c;
            const bool c4 = !c1 && !c2 && !c3 && Hij == PG_AT(M.H, i, j - 1) + S.q;
            ext_left = c1 || c3;
            if (c1 || c2 || c3 || c4) { prev_i = i; prev_j = j - 1; found = true; }
        }
        PG_PUSH(i == prev_i ? -1 : g.r2n[i - 1], j == prev_j ? -1 : j - 1);
        i = prev_i; j = prev_j;
        if (ext_left) {
            for (;;) {
                PG_PUSH(-1, j - 1);
                --j;
                if (PG_AT(M.E, i, j) + S.e != PG_AT(M.E, i, j + 1) && PG_AT(M.Q, i, j) + S.c != PG_AT(M.Q, i, j + 1)) break;
            }
        } else if (ext_up) {
            for (;;) {
                bool stop = false;
                prev_i = 0;
                const int node = g.r2n[i - 1];
                const int ic = g.in_cnt[node];
                for (int p = 0; p < ic; ++p) {
                    const int pi = g.n2r[g.in_src[node * g.deg + p]] + 1;
                    const bool s1 = PG_AT(M.F, i, j) == PG_AT(M.H, pi, j) + S.g;
                    const bool s2 = !s1 && PG_AT(M.F, i, j) == PG_AT(M.F, pi, j) + S.e;
                    const bool s3 = !s1 && !s2 && PG_AT(M.O, i, j) == PG_AT(M.H, pi, j) + S.q;
                    const bool s4 = !s1 && !s2 && !s3 && PG_AT(M.O, i, j) == PG_AT(M.O, pi, j) + S.c;
                    // `stop = c1 || .. || (stop = c3) || ..`: the last assignment evaluated wins
                    if (s1) stop = true; else if (s2) stop = false; else stop = s3;
                    if (s1 || s2 || s3 || s4) { prev_i = pi; break; }
                }
                PG_PUSH(node, -1);
                i = prev_i;
                if (stop || i == 0) break;
            }
        }
        if (np > g.aln_path_cap) { g.err |= POA_ERR_NODES; break; }
    }
#undef PG_AT
#undef PG_PUSH
    g.n_path = np <= g.aln_path_cap ? np : 0;
}

}  // namespace gbx
