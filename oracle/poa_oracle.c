/* poa_oracle.c — CPU restatement of spoa's partial-order alignment consensus as
 * the GenomicsBench `poa` driver uses it (R/benchmarks/poa/msa_spoa_omp.cpp:184-252:
 * createAlignmentEngine(kNW, m, n, g, e, q, c), createGraph, align,
 * add_alignment, generate_consensus).  TEST INFRASTRUCTURE ONLY.
 *
 * Parity: UNPINNED.  The arithmetic lives in libspoa, built from the un-vendored
 * submodule tools/spoa (arun-sub/spoa, default branch, commit unknown:
 * R/.gitmodules:18-20; the directory is empty in this checkout) and the
 * reference tree holds no expected consensus.  This file restates the published
 * spoa v3 algorithm (sisd_alignment_engine linear / affine / convex NW, Graph::add_alignment,
 * Graph::topological_sort, Graph::traverse_heaviest_bundle + branch_completion;
 * SURVEY.md Appendix D) and is anchored by the known-answer tests in
 * tests/test_poa_cpu.py (identical reads, majority votes, order invariants).
 */
#include <limits.h>
#include <stdlib.h>
#include <string.h>
#include "gbx_oracle.h"

#define NEG_INF (INT_MIN + 1024)          /* spoa kNegativeInfinity */

typedef struct { int *v; int n, cap; } ivec;
static void iv_push(ivec *a, int x)
{
    if (a->n == a->cap) { a->cap = a->cap ? a->cap * 2 : 4; a->v = (int *)realloc(a->v, sizeof(int) * (size_t)a->cap); }
    a->v[a->n++] = x;
}

typedef struct { int begin, end; int64_t weight; } edge_t;
typedef struct { int code; ivec in, out, aligned; } node_t;

typedef struct {
    node_t *nodes; int n_nodes, cap_nodes;
    edge_t *edges; int n_edges, cap_edges;
    int num_seqs, num_codes;
    int coder[256]; char decoder[256];
    ivec rank_to_node;
    ivec consensus;
} graph_t;

static void graph_init(graph_t *g)
{
    memset(g, 0, sizeof(*g));
    for (int i = 0; i < 256; ++i) g->coder[i] = -1;
}
static void graph_free(graph_t *g)
{
    for (int i = 0; i < g->n_nodes; ++i) { free(g->nodes[i].in.v); free(g->nodes[i].out.v); free(g->nodes[i].aligned.v); }
    free(g->nodes); free(g->edges); free(g->rank_to_node.v); free(g->consensus.v);
}
static int add_node(graph_t *g, int code)
{
    if (g->n_nodes == g->cap_nodes) {
        g->cap_nodes = g->cap_nodes ? g->cap_nodes * 2 : 256;
        g->nodes = (node_t *)realloc(g->nodes, sizeof(node_t) * (size_t)g->cap_nodes);
    }
    node_t *nd = &g->nodes[g->n_nodes];
    memset(nd, 0, sizeof(*nd));
    nd->code = code;
    return g->n_nodes++;
}
/* Graph::add_edge: bump an existing begin->end edge or append a new one to both adjacency lists */
static void add_edge(graph_t *g, int b, int e, int64_t w)
{
    node_t *nb = &g->nodes[b];
    for (int k = 0; k < nb->out.n; ++k)
        if (g->edges[nb->out.v[k]].end == e) { g->edges[nb->out.v[k]].weight += w; return; }
    if (g->n_edges == g->cap_edges) {
        g->cap_edges = g->cap_edges ? g->cap_edges * 2 : 512;
        g->edges = (edge_t *)realloc(g->edges, sizeof(edge_t) * (size_t)g->cap_edges);
    }
    g->edges[g->n_edges].begin = b; g->edges[g->n_edges].end = e; g->edges[g->n_edges].weight = w;
    iv_push(&g->nodes[b].out, g->n_edges);
    iv_push(&g->nodes[e].in, g->n_edges);
    ++g->n_edges;
}
/* Graph::add_sequence: a fresh chain for seq[begin,end); returns its first node or -1 */
static int add_chain(graph_t *g, const char *seq, int begin, int end)
{
    if (begin == end) return -1;
    int first = add_node(g, g->coder[(unsigned char)seq[begin]]);
    for (int i = begin + 1; i < end; ++i) {
        int id = add_node(g, g->coder[(unsigned char)seq[i]]);
        add_edge(g, id - 1, id, 2);               /* weights[i-1] + weights[i], every base weight 1 */
    }
    return first;
}

/* Graph::topological_sort: iterative DFS, aligned nodes emitted next to each other */
static void topo_sort(graph_t *g)
{
    const int n = g->n_nodes;
    unsigned char *mark = (unsigned char *)calloc((size_t)n + 1, 1);
    unsigned char *check = (unsigned char *)malloc((size_t)n + 1);
    memset(check, 1, (size_t)n + 1);
    ivec st = {0, 0, 0};
    g->rank_to_node.n = 0;
    for (int i = 0; i < n; ++i) {
        if (mark[i]) continue;
        iv_push(&st, i);
        while (st.n) {
            const int id = st.v[st.n - 1];
            int valid = 1;
            if (mark[id] != 2) {
                const node_t *nd = &g->nodes[id];
                for (int k = 0; k < nd->in.n; ++k) {
                    const int b = g->edges[nd->in.v[k]].begin;
                    if (mark[b] != 2) { iv_push(&st, b); valid = 0; }
                }
                if (check[id]) {
                    for (int k = 0; k < nd->aligned.n; ++k) {
                        const int a = nd->aligned.v[k];
                        if (mark[a] != 2) { iv_push(&st, a); check[a] = 0; valid = 0; }
                    }
                }
                if (valid) {
                    mark[id] = 2;
                    if (check[id]) {
                        iv_push(&g->rank_to_node, id);
                        for (int k = 0; k < nd->aligned.n; ++k) iv_push(&g->rank_to_node, nd->aligned.v[k]);
                    }
                } else mark[id] = 1;
            }
            if (valid) --st.n;
        }
    }
    free(mark); free(check); free(st.v);
}

typedef struct { int node, pos; } apair;
typedef struct { apair *v; int n, cap; } avec;
static void av_push(avec *a, int node, int pos)
{
    if (a->n == a->cap) { a->cap = a->cap ? a->cap * 2 : 256; a->v = (apair *)realloc(a->v, sizeof(apair) * (size_t)a->cap); }
    a->v[a->n].node = node; a->v[a->n].pos = pos; ++a->n;
}

/* Graph::add_alignment(alignment, sequence, weight = 1) */
static void add_alignment(graph_t *g, const avec *aln, const char *seq, int len)
{
    if (len == 0) return;
    for (int i = 0; i < len; ++i) {
        const unsigned char c = (unsigned char)seq[i];
        if (g->coder[c] == -1) { g->coder[c] = g->num_codes; g->decoder[g->num_codes] = (char)c; ++g->num_codes; }
    }
    if (aln->n == 0) {
        add_chain(g, seq, 0, len);
        ++g->num_seqs;
        topo_sort(g);
        return;
    }
    int first_pos = -1, last_pos = -1;
    for (int i = 0; i < aln->n; ++i)
        if (aln->v[i].pos != -1) { if (first_pos < 0) first_pos = aln->v[i].pos; last_pos = aln->v[i].pos; }
    const int before = g->n_nodes;
    add_chain(g, seq, 0, first_pos);
    int head = before == g->n_nodes ? -1 : g->n_nodes - 1;
    const int tail = add_chain(g, seq, last_pos + 1, len);
    int64_t prev_w = head == -1 ? 0 : 1;
    for (int i = 0; i < aln->n; ++i) {
        if (aln->v[i].pos == -1) continue;
        const int code = g->coder[(unsigned char)seq[aln->v[i].pos]];
        int id;
        if (aln->v[i].node == -1) {
            id = add_node(g, code);
        } else if (g->nodes[aln->v[i].node].code == code) {
            id = aln->v[i].node;
        } else {
            const int base = aln->v[i].node;
            int found = -1;
            for (int k = 0; k < g->nodes[base].aligned.n; ++k)
                if (g->nodes[g->nodes[base].aligned.v[k]].code == code) { found = g->nodes[base].aligned.v[k]; break; }
            if (found == -1) {
                id = add_node(g, code);
                const int na = g->nodes[base].aligned.n;
                for (int k = 0; k < na; ++k) {
                    const int a = g->nodes[base].aligned.v[k];
                    iv_push(&g->nodes[id].aligned, a);
                    iv_push(&g->nodes[a].aligned, id);
                }
                iv_push(&g->nodes[id].aligned, base);
                iv_push(&g->nodes[base].aligned, id);
            } else id = found;
        }
        if (head != -1) add_edge(g, head, id, prev_w + 1);
        head = id;
        prev_w = 1;
    }
    if (tail != -1) add_edge(g, head, tail, prev_w + 1);
    ++g->num_seqs;
    topo_sort(g);
}

/* SisdAlignmentEngine::align, kNW, affine/convex gaps (affine == convex with q=g, c=e) */
static int64_t align_nw(const graph_t *gr, const char *seq, int len, const gbx_poa_params *P, avec *out)
{
    out->n = 0;
    if (gr->n_nodes == 0 || len == 0) return 0;
    const int m_ = P->m, n_ = P->n, g_ = P->g, e_ = P->e;
    int q_ = P->q, c_ = P->c;
    if (g_ <= q_ || e_ >= c_) { q_ = g_; c_ = e_; }            /* affine subtype */
    const int W = len + 1, Hh = gr->n_nodes + 1;
    const size_t cells = (size_t)W * (size_t)Hh;
    int *H = (int *)malloc(sizeof(int) * cells * 5);
    int *F = H + cells, *E = F + cells, *O = E + cells, *Q = O + cells;
    int *rank = (int *)malloc(sizeof(int) * (size_t)gr->n_nodes);
    const int *r2n = gr->rank_to_node.v;
    for (int i = 0; i < gr->n_nodes; ++i) rank[r2n[i]] = i;

    /* initialise, sisd_alignment_engine.cpp `initialize` */
    O[0] = Q[0] = 0; F[0] = E[0] = 0;
    for (int j = 1; j < W; ++j) {
        O[j] = NEG_INF; Q[j] = q_ + (j - 1) * c_;
        F[j] = NEG_INF; E[j] = g_ + (j - 1) * e_;
    }
    for (int i = 1; i < Hh; ++i) {
        const node_t *nd = &gr->nodes[r2n[i - 1]];
        int po = nd->in.n == 0 ? q_ - c_ : NEG_INF, pf = nd->in.n == 0 ? g_ - e_ : NEG_INF;
        for (int k = 0; k < nd->in.n; ++k) {
            const size_t pi = (size_t)(rank[gr->edges[nd->in.v[k]].begin] + 1) * W;
            if (O[pi] > po) po = O[pi];
            if (F[pi] > pf) pf = F[pi];
        }
        O[(size_t)i * W] = po + c_; Q[(size_t)i * W] = NEG_INF;
        F[(size_t)i * W] = pf + e_; E[(size_t)i * W] = NEG_INF;
    }
    H[0] = 0;
    for (int j = 1; j < W; ++j) H[j] = Q[j] > E[j] ? Q[j] : E[j];
    for (int i = 1; i < Hh; ++i) { const size_t o = (size_t)i * W; H[o] = O[o] > F[o] ? O[o] : F[o]; }

    int max_score = NEG_INF, max_i = -1, max_j = -1;
    for (int r = 0; r < gr->n_nodes; ++r) {
        const node_t *nd = &gr->nodes[r2n[r]];
        const char letter = gr->decoder[nd->code];
        const size_t ro = (size_t)(r + 1) * W;
        int *Hr = H + ro, *Fr = F + ro, *Er = E + ro, *Or = O + ro, *Qr = Q + ro;
        for (int p = 0; p < (nd->in.n ? nd->in.n : 1); ++p) {
            const size_t po = nd->in.n ? (size_t)(rank[gr->edges[nd->in.v[p]].begin] + 1) * W : 0;
            const int *Hp = H + po, *Fp = F + po, *Op = O + po;
            for (int j = 1; j < W; ++j) {
                const int f = Hp[j] + g_ > Fp[j] + e_ ? Hp[j] + g_ : Fp[j] + e_;
                const int o = Hp[j] + q_ > Op[j] + c_ ? Hp[j] + q_ : Op[j] + c_;
                const int h = Hp[j - 1] + (letter == seq[j - 1] ? m_ : n_);
                if (p == 0) { Fr[j] = f; Or[j] = o; Hr[j] = h; }
                else { if (f > Fr[j]) Fr[j] = f; if (o > Or[j]) Or[j] = o; if (h > Hr[j]) Hr[j] = h; }
            }
        }
        for (int j = 1; j < W; ++j) {
            Er[j] = Hr[j - 1] + g_ > Er[j - 1] + e_ ? Hr[j - 1] + g_ : Er[j - 1] + e_;
            Qr[j] = Hr[j - 1] + q_ > Qr[j - 1] + c_ ? Hr[j - 1] + q_ : Qr[j - 1] + c_;
            int h = Hr[j];
            if (Fr[j] > h) h = Fr[j];
            if (Er[j] > h) h = Er[j];
            if (Or[j] > h) h = Or[j];
            if (Qr[j] > h) h = Qr[j];
            Hr[j] = h;
        }
        if (nd->out.n == 0 && max_score < Hr[W - 1]) { max_score = Hr[W - 1]; max_i = r + 1; max_j = W - 1; }
    }

    if (!(max_i == -1 && max_j == -1)) {
        int i = max_i, j = max_j, prev_i = 0, prev_j = 0;
#define AT(M, a, b) (M)[(size_t)(a) * W + (b)]
        while (!(i == 0 && j == 0)) {
            const int Hij = AT(H, i, j);
            int found = 0, ext_left = 0, ext_up = 0;
            if (i != 0 && j != 0) {
                const node_t *nd = &gr->nodes[r2n[i - 1]];
                const int mc = gr->decoder[nd->code] == seq[j - 1] ? m_ : n_;
                for (int p = 0; p < (nd->in.n ? nd->in.n : 1) && !found; ++p) {
                    const int pi = nd->in.n ? rank[gr->edges[nd->in.v[p]].begin] + 1 : 0;
                    if (Hij == AT(H, pi, j - 1) + mc) { prev_i = pi; prev_j = j - 1; found = 1; }
                }
            }
            if (!found && i != 0) {
                const node_t *nd = &gr->nodes[r2n[i - 1]];
                for (int p = 0; p < (nd->in.n ? nd->in.n : 1) && !found; ++p) {
                    const int pi = nd->in.n ? rank[gr->edges[nd->in.v[p]].begin] + 1 : 0;
                    if ((ext_up |= (Hij == AT(F, pi, j) + e_)) || Hij == AT(H, pi, j) + g_ ||
                        (ext_up |= (Hij == AT(O, pi, j) + c_)) || Hij == AT(H, pi, j) + q_) {
                        prev_i = pi; prev_j = j; found = 1;
                    }
                }
            }
            if (!found && j != 0) {
                if ((ext_left |= (Hij == AT(E, i, j - 1) + e_)) || Hij == AT(H, i, j - 1) + g_ ||
                    (ext_left |= (Hij == AT(Q, i, j - 1) + c_)) || Hij == AT(H, i, j - 1) + q_) {
                    prev_i = i; prev_j = j - 1; found = 1;
                }
            }
            av_push(out, i == prev_i ? -1 : r2n[i - 1], j == prev_j ? -1 : j - 1);
            i = prev_i; j = prev_j;
            if (ext_left) {
                for (;;) {
                    av_push(out, -1, j - 1);
                    --j;
                    if (AT(E, i, j) + e_ != AT(E, i, j + 1) && AT(Q, i, j) + c_ != AT(Q, i, j + 1)) break;
                }
            } else if (ext_up) {
                for (;;) {
                    int stop = 0;
                    prev_i = 0;
                    const node_t *nd = &gr->nodes[r2n[i - 1]];
                    for (int p = 0; p < nd->in.n; ++p) {
                        const int pi = rank[gr->edges[nd->in.v[p]].begin] + 1;
                        if ((stop = (AT(F, i, j) == AT(H, pi, j) + g_)) || AT(F, i, j) == AT(F, pi, j) + e_ ||
                            (stop = (AT(O, i, j) == AT(H, pi, j) + q_)) || AT(O, i, j) == AT(O, pi, j) + c_) {
                            prev_i = pi;
                            break;
                        }
                    }
                    av_push(out, r2n[i - 1], -1);
                    i = prev_i;
                    if (stop || i == 0) break;
                }
            }
        }
#undef AT
        for (int a = 0, b = out->n - 1; a < b; ++a, --b) { apair t = out->v[a]; out->v[a] = out->v[b]; out->v[b] = t; }
    }
    free(H); free(rank);
    return (int64_t)gr->n_nodes * (int64_t)len;
}

/* SisdAlignmentEngine::align for the LINEAR subtype (spoa v3 createAlignmentEngine: `g >= e` selects it and sets e = g; the
 * driver's CLI reaches it with -o 0,... : msa_spoa_omp.cpp:170-196).  One matrix: H(i,j) = max over the in-edge sources p of
 * H(p,j-1) + score and H(p,j) + g, then H(i,j-1) + g along the row; first row j*g, first column max over sources of H(p,0) + g
 * (g for a node without in-edges).  Backtrack: diagonal candidates in in-edge order, then vertical ones in in-edge order, then
 * the horizontal one - one cell per step, no extension walks. */
static int64_t align_nw_linear(const graph_t *gr, const char *seq, int len, const gbx_poa_params *P, avec *out)
{
    out->n = 0;
    if (gr->n_nodes == 0 || len == 0) return 0;
    const int m_ = P->m, n_ = P->n, g_ = P->g;
    const int W = len + 1, Hh = gr->n_nodes + 1;
    int *H = (int *)malloc(sizeof(int) * (size_t)W * (size_t)Hh);
    int *rank = (int *)malloc(sizeof(int) * (size_t)gr->n_nodes);
    const int *r2n = gr->rank_to_node.v;
    for (int i = 0; i < gr->n_nodes; ++i) rank[r2n[i]] = i;
#define AT(a, b) H[(size_t)(a) * W + (b)]
    AT(0, 0) = 0;
    for (int j = 1; j < W; ++j) AT(0, j) = j * g_;
    for (int i = 1; i < Hh; ++i) {
        const node_t *nd = &gr->nodes[r2n[i - 1]];
        int pen = nd->in.n == 0 ? 0 : NEG_INF;
        for (int k = 0; k < nd->in.n; ++k) {
            const int v = AT(rank[gr->edges[nd->in.v[k]].begin] + 1, 0);
            if (v > pen) pen = v;
        }
        AT(i, 0) = pen + g_;
    }
    int max_score = NEG_INF, max_i = -1, max_j = -1;
    for (int r = 0; r < gr->n_nodes; ++r) {
        const node_t *nd = &gr->nodes[r2n[r]];
        const char letter = gr->decoder[nd->code];
        const int i = r + 1;
        for (int p = 0; p < (nd->in.n ? nd->in.n : 1); ++p) {
            const int pi = nd->in.n ? rank[gr->edges[nd->in.v[p]].begin] + 1 : 0;
            for (int j = 1; j < W; ++j) {
                const int d = AT(pi, j - 1) + (letter == seq[j - 1] ? m_ : n_), v = AT(pi, j) + g_;
                const int h = d > v ? d : v;
                if (p == 0 || h > AT(i, j)) AT(i, j) = h;
            }
        }
        for (int j = 1; j < W; ++j) if (AT(i, j - 1) + g_ > AT(i, j)) AT(i, j) = AT(i, j - 1) + g_;
        if (nd->out.n == 0 && max_score < AT(i, W - 1)) { max_score = AT(i, W - 1); max_i = i; max_j = W - 1; }
    }
    if (!(max_i == -1 && max_j == -1)) {
        int i = max_i, j = max_j, prev_i = 0, prev_j = 0;
        while (!(i == 0 && j == 0)) {
            const int Hij = AT(i, j);
            int found = 0;
            const node_t *nd = i ? &gr->nodes[r2n[i - 1]] : NULL;
            if (i != 0 && j != 0) {
                const int mc = gr->decoder[nd->code] == seq[j - 1] ? m_ : n_;
                for (int p = 0; p < (nd->in.n ? nd->in.n : 1) && !found; ++p) {
                    const int pi = nd->in.n ? rank[gr->edges[nd->in.v[p]].begin] + 1 : 0;
                    if (Hij == AT(pi, j - 1) + mc) { prev_i = pi; prev_j = j - 1; found = 1; }
                }
            }
            if (!found && i != 0) {
                for (int p = 0; p < (nd->in.n ? nd->in.n : 1) && !found; ++p) {
                    const int pi = nd->in.n ? rank[gr->edges[nd->in.v[p]].begin] + 1 : 0;
                    if (Hij == AT(pi, j) + g_) { prev_i = pi; prev_j = j; found = 1; }
                }
            }
            if (!found && j != 0 && Hij == AT(i, j - 1) + g_) { prev_i = i; prev_j = j - 1; found = 1; }
            av_push(out, i == prev_i ? -1 : r2n[i - 1], j == prev_j ? -1 : j - 1);
            i = prev_i; j = prev_j;
        }
        for (int a = 0, b = out->n - 1; a < b; ++a, --b) { apair t = out->v[a]; out->v[a] = out->v[b]; out->v[b] = t; }
    }
#undef AT
    free(H); free(rank);
    return (int64_t)gr->n_nodes * (int64_t)len;
}

/* Graph::branch_completion */
static int branch_completion(const graph_t *g, int64_t *scores, int *pred, int rank)
{
    const int *r2n = g->rank_to_node.v;
    const int node_id = r2n[rank];
    const node_t *nd = &g->nodes[node_id];
    for (int k = 0; k < nd->out.n; ++k) {
        const node_t *tn = &g->nodes[g->edges[nd->out.v[k]].end];
        for (int z = 0; z < tn->in.n; ++z) {
            const int b = g->edges[tn->in.v[z]].begin;
            if (b != node_id) scores[b] = -1;
        }
    }
    int64_t max_score = 0;
    int max_id = 0;
    for (int i = rank + 1; i < g->n_nodes; ++i) {
        const int id = r2n[i];
        const node_t *x = &g->nodes[id];
        scores[id] = -1; pred[id] = -1;
        for (int k = 0; k < x->in.n; ++k) {
            const edge_t *ed = &g->edges[x->in.v[k]];
            if (scores[ed->begin] == -1) continue;
            if (scores[id] < ed->weight || (scores[id] == ed->weight && scores[pred[id]] <= scores[ed->begin])) {
                scores[id] = ed->weight; pred[id] = ed->begin;
            }
        }
        if (pred[id] != -1) scores[id] += scores[pred[id]];
        if (max_score < scores[id]) { max_score = scores[id]; max_id = id; }
    }
    return max_id;
}

/* Graph::traverse_heaviest_bundle + generate_consensus */
static int consensus(graph_t *g, char *out, int64_t cap)
{
    const int n = g->n_nodes;
    if (n == 0) return 0;
    int *pred = (int *)malloc(sizeof(int) * (size_t)n);
    int64_t *scores = (int64_t *)malloc(sizeof(int64_t) * (size_t)n);
    for (int i = 0; i < n; ++i) { pred[i] = -1; scores[i] = -1; }
    const int *r2n = g->rank_to_node.v;
    int max_id = 0;
    for (int r = 0; r < n; ++r) {
        const int id = r2n[r];
        const node_t *x = &g->nodes[id];
        for (int k = 0; k < x->in.n; ++k) {
            const edge_t *ed = &g->edges[x->in.v[k]];
            if (scores[id] < ed->weight || (scores[id] == ed->weight && scores[pred[id]] <= scores[ed->begin])) {
                scores[id] = ed->weight; pred[id] = ed->begin;
            }
        }
        if (pred[id] != -1) scores[id] += scores[pred[id]];
        if (scores[max_id] < scores[id]) max_id = id;
    }
    if (g->nodes[max_id].out.n != 0) {
        int *n2r = (int *)malloc(sizeof(int) * (size_t)n);
        for (int i = 0; i < n; ++i) n2r[r2n[i]] = i;
        while (g->nodes[max_id].out.n != 0) max_id = branch_completion(g, scores, pred, n2r[max_id]);
        free(n2r);
    }
    g->consensus.n = 0;
    while (pred[max_id] != -1) { iv_push(&g->consensus, max_id); max_id = pred[max_id]; }
    iv_push(&g->consensus, max_id);
    const int len = g->consensus.n;
    for (int k = 0; k < len && k < cap; ++k) out[k] = g->decoder[g->nodes[g->consensus.v[len - 1 - k]].code];
    free(pred); free(scores);
    return len;
}

/* one window (the driver's per-batch loop, msa_spoa_omp.cpp:237-252); stats: [0]=nodes [1]=edges [2]=DP cells */
int oracle_poa_window(const gbx_poa_params *P, int n_seqs, const char *const *seqs, const int32_t *lens,
                      char *cons, int64_t cons_cap, int64_t *stats)
{
    graph_t g;
    graph_init(&g);
    avec aln = {0, 0, 0};
    int64_t cells = 0;
    for (int s = 0; s < n_seqs; ++s) {
        /* createAlignmentEngine: g >= e is the linear subtype, else affine / convex (align_nw decides between those two) */
        cells += P->g >= P->e ? align_nw_linear(&g, seqs[s], lens[s], P, &aln) : align_nw(&g, seqs[s], lens[s], P, &aln);
        add_alignment(&g, &aln, seqs[s], lens[s]);
    }
    const int len = consensus(&g, cons, cons_cap);
    if (stats) { stats[0] = g.n_nodes; stats[1] = g.n_edges; stats[2] = cells; }
    free(aln.v);
    graph_free(&g);
    return len;
}

void oracle_poa_consensus(const gbx_poa_params *P, int64_t n_windows, const int64_t *win_first_seq,
                          const int64_t *seq_off, const int32_t *seq_len, const char *arena,
                          char *cons, int32_t *cons_len, int64_t cons_stride, int nthreads, int64_t *cells)
{
    int64_t total = 0;
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads) reduction(+ : total)
    for (int64_t w = 0; w < n_windows; ++w) {
        const int64_t s0 = win_first_seq[w], ns = win_first_seq[w + 1] - s0;
        const char **ptrs = (const char **)malloc(sizeof(char *) * (size_t)(ns > 0 ? ns : 1));
        for (int64_t k = 0; k < ns; ++k) ptrs[k] = arena + seq_off[s0 + k];
        int64_t st[3];
        cons_len[w] = oracle_poa_window(P, (int)ns, ptrs, seq_len + s0, cons + w * cons_stride, cons_stride, st);
        total += st[2];
        free(ptrs);
    }
    if (cells) *cells = total;
}
