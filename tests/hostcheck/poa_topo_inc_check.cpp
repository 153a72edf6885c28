// poa_topo_inc_check.cpp — TEST-ONLY host model of the incremental topological sort of csrc/poa_kernels.hip
// (poa_topo_sort_lds): the previous order block by block, untouched blocks copied, the walk repeated from the roots of
// the blocks that hold a node with a new in-edge or aligned node.  Run on real windows beside the full sort
// (poa_topo_sort, poa_graph.h) after every add_alignment; returns the number of sorts whose orders differ.
#include "poa_hostcheck.cpp"

namespace {
struct Inc {
    std::vector<uint8_t> mark, check, root, changed;
    std::vector<int32_t> stack, ord;
    int nr = 0;
    // poa_topo_sort's walk from one root, marks / check shared with the rest of this sort
    void walk(PoaGraph &g, int i)
    {
        int sp = 0;
        stack[sp++] = i;
        while (sp) {
            const int id = stack[sp - 1];
            const int mk = mark[id], ic = g.in_cnt[id], ac = g.aln_cnt[id];
            const bool chk = check[id] != 0;
            bool valid = true;
            if (mk != 2) {
                for (int k = 0; k < ic; ++k) { const int b = PG_IN_SRC(g, id, k); if (mark[b] != 2) { stack[sp++] = b; valid = false; } }
                if (chk) for (int k = 0; k < ac; ++k) { const int a = g.aln[id * POA_ALN_STRIDE + k]; if (mark[a] != 2) { stack[sp++] = a; check[a] = 0; valid = false; } }
                if (valid) {
                    mark[id] = 2;
                    if (chk) { ord[nr++] = id; for (int k = 0; k < ac; ++k) ord[nr++] = g.aln[id * POA_ALN_STRIDE + k]; }
                } else mark[id] = 1;
            }
            if (valid) --sp;
        }
    }
};
}

// returns sorts whose incremental order differs from the full one; stats[0] = sorts, [1] = blocks walked, [2] = blocks in all
extern "C" int hostcheck_poa_topo_inc(const gbx_poa_params *P, int n_seqs, const char *const *seqs, const int32_t *lens, int ncap, int deg, int64_t *stats)
{
    int lmax = 0;
    for (int s = 0; s < n_seqs; ++s) lmax = std::max<int>(lmax, lens[s]);
    PoaGraph g;
    deg = (deg + 3) & ~3;
    g.ncap = ncap; g.deg = deg; g.stk_cap = ncap * 4 + 64; g.aln_path_cap = ncap + lmax + 8;
    std::vector<uint8_t> code(ncap), icnt(ncap), ocnt(ncap), acnt(ncap), oslot((size_t)ncap * deg), mark(ncap), check(ncap), dec(256);
    std::vector<int32_t> isrc((size_t)ncap * deg), iwt((size_t)ncap * deg), odst((size_t)ncap * deg), aln((size_t)ncap * POA_ALN_STRIDE + 8),
        r2n(ncap), n2r(ncap), stack(g.stk_cap), score(ncap), pred(ncap), pn(g.aln_path_cap), pp(g.aln_path_cap);
    std::vector<int16_t> coder(256);
    g.code = code.data(); g.in_cnt = icnt.data(); g.out_cnt = ocnt.data(); g.aln_cnt = acnt.data();
    g.in_src = isrc.data(); g.in_wt = iwt.data(); g.out_dst = odst.data(); g.out_slot = oslot.data(); g.aln = aln.data();
    std::vector<int32_t> isx((size_t)ncap * deg), iwx((size_t)ncap * deg), odx((size_t)ncap * deg); std::vector<uint8_t> osx((size_t)ncap * deg);
    g.in_src_x = isx.data(); g.in_wt_x = iwx.data(); g.out_dst_x = odx.data(); g.out_slot_x = osx.data();
    g.r2n = r2n.data(); g.n2r = n2r.data(); g.mark = mark.data(); g.check = check.data(); g.stack = stack.data();
    g.score = score.data(); g.pred = pred.data(); std::vector<int32_t> cpath(ncap + 1); g.cons_path = cpath.data(); g.path_node = pn.data(); g.path_pos = pp.data();
    g.coder = coder.data(); g.decoder = dec.data();
    poa_graph_reset(g);
    PoaScore S = {P->m, P->n, P->g, P->e, P->q, P->c, 0};
    if (S.g >= S.e) { S.e = S.q = S.c = S.g; S.linear = 1; }      // as poa_launch does (poa_kernels.hip)
    else if (S.g <= S.q || S.e >= S.c) { S.q = S.g; S.c = S.e; }
    const size_t plane = (size_t)(ncap + 1) * poa_row_stride(lmax);
    std::vector<poa_cell_t> mat(plane * 5);
    PoaMatrices M = {mat.data(), mat.data() + plane, mat.data() + 2 * plane, mat.data() + 3 * plane, mat.data() + 4 * plane, 0};
    Inc I;
    I.mark.assign(ncap, 0); I.check.assign(ncap, 1); I.root.assign(ncap, 0); I.changed.assign(ncap, 0); I.stack.assign(g.stk_cap, 0); I.ord.assign(ncap, 0);
    std::vector<int32_t> old_order;                 // the incremental sort's own previous order
    std::vector<uint8_t> ic0(ncap), ac0(ncap);
    int bad = 0;
    int64_t sorts = 0, walked = 0, blocks = 0;
    for (int s = 0; s < n_seqs; ++s) {
        const uint8_t *seq = (const uint8_t *)seqs[s];
        g.n_path = 0;
        if (g.n_nodes != 0 && lens[s] != 0 && g.err == 0) {
            int mi, mj;
            scalar_dp(g, M, S, seq, lens[s], &mi, &mj);
            for (int r = 0; r < g.n_nodes; ++r) poa_rowdesc_one(g, r);
            poa_traceback(g, M, S, seq, mi, mj);
        }
        const int n_old = g.n_nodes;
        for (int i = 0; i < n_old; ++i) { ic0[i] = g.in_cnt[i]; ac0[i] = g.aln_cnt[i]; }
        if (g.err == 0) poa_add_alignment(g, seq, lens[s]);      // full sort inside: g.r2n is the reference order
        if (g.err) break;
        const int n = g.n_nodes;
        // ---- the incremental sort
        for (int i = 0; i < n; ++i) { I.mark[i] = 0; I.check[i] = 1; I.changed[i] = i < n_old && (g.in_cnt[i] != ic0[i] || g.aln_cnt[i] != ac0[i]); if (i >= n_old) I.root[i] = 0; }
        I.nr = 0;
        if ((int)old_order.size() == n_old && n_old > 0) {
            // a block = what one root's walk emitted: the nodes it pulled in, then the root, then the root's aligned nodes
            // (a root is always the first of its group to be examined, so it emits the group)
            int q = 0;
            while (q < n_old) {
                // blocks from q on: copy the untouched ones, walk the first touched one
                int t = q, blk_start = q, c = -1, root_rank = -1, blk_end = -1;
                while (t < n_old) {
                    int e = t;
                    while (e < n_old && !I.root[old_order[e]]) ++e;
                    if (e >= n_old) { blk_start = n_old; break; }      // (cannot happen: the last node emitted belongs to a root's group)
                    const int r = old_order[e];
                    int mates = 0;
                    for (int k = 0; k < g.aln_cnt[r]; ++k) mates += g.aln[r * POA_ALN_STRIDE + k] < n_old;
                    const int end = e + mates;                          // last rank of the block
                    bool any = false;
                    for (int u = blk_start; u <= end; ++u) any = any || I.changed[old_order[u]];
                    if (any) { c = blk_start; root_rank = e; blk_end = end; break; }
                    t = end + 1; blk_start = t;
                }
                const int copy_end = c >= 0 ? c : n_old;
                for (int u = q; u < copy_end; ++u) { I.ord[I.nr++] = old_order[u]; I.mark[old_order[u]] = 2; }
                if (c < 0) break;
                I.walk(g, old_order[root_rank]);
                ++walked;
                q = blk_end + 1;
            }
            for (int q2 = 0; q2 < n_old; ++q2) blocks += I.root[old_order[q2]];
            for (int i = n_old; i < n; ++i) if (I.mark[i] == 0) { I.root[i] = 1; I.walk(g, i); }
            ++sorts;
        } else {
            for (int i = 0; i < n; ++i) if (I.mark[i] == 0) { I.root[i] = 1; I.walk(g, i); }
        }
        bool same = I.nr == n;
        for (int r = 0; r < n && same; ++r) same = I.ord[r] == g.r2n[r];
        if (!same) ++bad;
        old_order.assign(g.r2n, g.r2n + n);        // continue from the reference order ...
        if (!same) {                                // ... with root flags recomputed from it when the incremental sort went wrong
            std::vector<uint8_t> mk(n, 0);
            Inc J; J.mark.assign(n, 0); J.check.assign(n, 1); J.stack.assign(g.stk_cap, 0); J.ord.assign(n, 0);
            for (int i = 0; i < n; ++i) { I.root[i] = 0; if (J.mark[i] == 0) { I.root[i] = 1; J.walk(g, i); } }
        }
    }
    if (stats) { stats[0] = sorts; stats[1] = walked; stats[2] = blocks; }
    return bad;
}
