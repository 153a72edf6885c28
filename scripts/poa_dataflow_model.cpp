// poa_dataflow_model.cpp - development aid (host only): how much row-level parallelism the DP of a window has.
// Runs windows of the 'large' generator through the host build of the product's graph code (poa_graph.h, the scalar DP of
// tests/hostcheck) and, at every alignment, plays the rows of the topological order through three schedules with a row = 1 time
// unit: wavefronts taking the rows in turn (the team kernel: static owner, run-ahead bound), rows taken in order by whichever
// wavefront is free first, and no limit on wavefronts (the critical path).
//   g++ -O2 -std=c++17 -fopenmp scripts/poa_dataflow_model.cpp genomicsbench_amd/datagen/datagen.c -o build_tmp/poa_dataflow_model
//   build_tmp/poa_dataflow_model [windows]
#include <cstdio>
#include <string>
#include "../tests/hostcheck/poa_hostcheck.cpp"
extern "C" void gbx_gen_poa_window(uint64_t seed, int64_t window, int mode, int32_t *n_reads, int32_t *read_len, char *out);

struct Acc { double serial = 0, st4 = 0, st4b16 = 0, st2 = 0, dyn4 = 0, dyn2 = 0, dyn8 = 0, ideal = 0, near1 = 0, near3 = 0, rows = 0; };

static void simulate(const PoaGraph &g, Acc &a)
{
    const int n = g.n_nodes;
    std::vector<std::vector<int>> preds(n + 1);
    for (int r = 0; r < n; ++r) {
        const int node = g.r2n[r];
        for (int k = 0; k < g.in_cnt[node]; ++k) preds[r + 1].push_back(g.n2r[PG_IN_SRC(g, node, k)] + 1);
    }
    a.serial += n; a.rows += n;
    for (int i = 1; i <= n; ++i) {
        int nearest = 1 << 30;
        for (int p : preds[i]) nearest = std::min(nearest, i - p);
        a.near1 += nearest == 1; a.near3 += nearest <= 3;
    }
    auto stat = [&](int NW, int BOUND, double ovh) {
        std::vector<double> fin(n + 1, 0.0), pre(n + 1, 0.0);      // pre[i] = max finish of rows <= i
        std::vector<double> wfree(NW, 0.0);
        for (int i = 1; i <= n; ++i) {
            double st = wfree[(i - 1) % NW];
            for (int p : preds[i]) st = std::max(st, fin[p]);
            if (i - BOUND >= 1) st = std::max(st, pre[i - BOUND]);
            fin[i] = st + 1.0 + ovh;
            wfree[(i - 1) % NW] = fin[i];
            pre[i] = std::max(pre[i - 1], fin[i]);
        }
        return pre[n];
    };
    auto dyn = [&](int NW, double ovh) {
        std::vector<double> fin(n + 1, 0.0), wfree(NW, 0.0);
        double grabbed = 0;                                        // rows are handed out in order: row i not before row i - 1 was taken
        for (int i = 1; i <= n; ++i) {
            int w = 0;
            for (int k = 1; k < NW; ++k) if (wfree[k] < wfree[w]) w = k;
            double st = std::max(wfree[w], grabbed);
            grabbed = st;
            for (int p : preds[i]) st = std::max(st, fin[p]);
            fin[i] = st + 1.0 + ovh;
            wfree[w] = fin[i];
        }
        double m = 0; for (double f : fin) m = std::max(m, f);
        return m;
    };
    a.st4 += stat(4, 6, 0.1); a.st4b16 += stat(4, 16, 0.1); a.st2 += stat(2, 6, 0.1);
    a.dyn4 += dyn(4, 0.1); a.dyn2 += dyn(2, 0.1); a.dyn8 += dyn(8, 0.1);
    {
        std::vector<double> fin(n + 1, 0.0); double m = 0;
        for (int i = 1; i <= n; ++i) { double st = 0; for (int p : preds[i]) st = std::max(st, fin[p]); fin[i] = st + 1.0; m = std::max(m, fin[i]); }
        a.ideal += m;
    }
}

int main(int argc, char **argv)
{
    const int nwin = argc > 1 ? atoi(argv[1]) : 8;
    gbx_poa_params P; P.m = 2; P.n = -4; P.g = -6; P.e = -2; P.q = -25; P.c = -1;
    Acc a;
    for (int w = 0; w < nwin; ++w) {
        int32_t nr, lens[64]; static char buf[64 * 1024];
        gbx_gen_poa_window(4001, w, 2, &nr, lens, buf);
        std::vector<const char *> seqs; const char *p = buf;
        for (int k = 0; k < nr; ++k) { seqs.push_back(p); p += lens[k]; }
        // the harness of hostcheck_poa_window with the schedule model at every alignment
        int lmax = 0; for (int s = 0; s < nr; ++s) lmax = std::max<int>(lmax, lens[s]);
        const int ncap = 6 * lmax + 256, deg = 64;
        PoaGraph g;
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
        PoaScore S = {P.m, P.n, P.g, P.e, P.q, P.c};
        const size_t plane = (size_t)(ncap + 1) * poa_row_stride(lmax);
        std::vector<poa_cell_t> mat(plane * 5);
        PoaMatrices M = {mat.data(), mat.data() + plane, mat.data() + 2 * plane, mat.data() + 3 * plane, mat.data() + 4 * plane, 0};
        for (int s = 0; s < nr; ++s) {
            const uint8_t *seq = (const uint8_t *)seqs[s];
            g.n_path = 0;
            if (g.n_nodes != 0 && lens[s] != 0 && g.err == 0) {
                int mi, mj;
                simulate(g, a);
                scalar_dp(g, M, S, seq, lens[s], &mi, &mj);
                for (int r = 0; r < g.n_nodes; ++r) poa_rowdesc_one(g, r);
                poa_traceback(g, M, S, seq, mi, mj);
            }
            if (g.err == 0) poa_add_alignment(g, seq, lens[s]);
        }
        fprintf(stderr, "window %d: %d reads, %d nodes, err %d\n", w, nr, g.n_nodes, g.err);
    }
    printf("%d windows, %.0f DP rows; nearest predecessor is the row before in %.1f %%, within three rows in %.1f %%\n", nwin, a.rows, 100 * a.near1 / a.rows, 100 * a.near3 / a.rows);
    printf("speed-up of the DP over one wavefront (row = 1, hand-over overhead 0.1):\n");
    printf("  4 wavefronts, rows in turn, run-ahead 6 (the team kernel)   %.2f\n", a.serial / a.st4);
    printf("  4 wavefronts, rows in turn, run-ahead 16                    %.2f\n", a.serial / a.st4b16);
    printf("  2 wavefronts, rows in turn, run-ahead 6                     %.2f\n", a.serial / a.st2);
    printf("  2 / 4 / 8 wavefronts, next row to the first free one        %.2f / %.2f / %.2f\n", a.serial / a.dyn2, a.serial / a.dyn4, a.serial / a.dyn8);
    printf("  no limit (critical path)                                    %.2f\n", a.serial / a.ideal);
    return 0;
}
