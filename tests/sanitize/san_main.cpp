// san_main.cpp — CPU-only sanitizer run of the host-side code that has no GPU dependency:
//   * oracle/*.c               (the five restatements, single- and multi-threaded entry points),
//   * tests/hostcheck/poa_hostcheck.cpp (the product's serial graph code, csrc/poa_graph.h, host build),
//   * genomicsbench_amd/datagen/datagen.c (the generators that feed them).
// Built by tests/sanitize/Makefile with -fsanitize=address,undefined and run by tests/test_sanitize_cpu.py;
// exits 0 when every run completes and the oracle and the product's graph code agree.  TEST INFRASTRUCTURE.
// GPU code is never sanitized this way (no GPU ASAN on this pool).
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../../oracle/gbx_oracle.h"

extern "C" {
void gbx_gen_bsw_lengths(uint64_t seed, int64_t first, int64_t n, int32_t *len1, int32_t *len2, int32_t *h0);
void gbx_gen_bsw_fill(uint64_t seed, int64_t first, int64_t n, const int32_t *len1, const int32_t *len2,
                      const int64_t *idr, const int64_t *idq, uint8_t *ref, uint8_t *qer);
void gbx_gen_chain_counts_many(uint64_t seed, int64_t first, int64_t n_calls, int64_t *counts);
void gbx_gen_chain_fill_many(uint64_t seed, int64_t first, int64_t n_calls, const int64_t *off, uint64_t *ax, uint64_t *ay);
void gbx_gen_phmm_counts_many(uint64_t seed, int64_t first, int64_t n_batches, int32_t *n_reads, int32_t *n_haps);
void gbx_gen_phmm_lengths_many(uint64_t seed, int64_t first, int64_t n_batches, const int64_t *roff, const int64_t *hoff,
                               int32_t *read_len, int32_t *hap_len);
void gbx_gen_phmm_fill_many(uint64_t seed, int64_t first, int64_t n_batches, const int64_t *roff, const int64_t *hoff,
                            const int64_t *read_off, const int64_t *hap_off, char *rs, char *q, char *qi, char *qd, char *qc, char *hap);
void gbx_gen_poa_counts_many(uint64_t seed, int64_t first, int64_t n_windows, int32_t *n_reads);
void gbx_gen_poa_many(uint64_t seed, int64_t first, int64_t n_windows, int mode, const int64_t *wf, int32_t *seq_len,
                      const int64_t *seq_off, char *arena);
void gbx_gen_abea_model(uint64_t seed, float *level_mean, float *level_stdv);
void gbx_gen_abea_counts_many(uint64_t seed, int64_t first, int64_t n_reads, int32_t *seq_len, int64_t *n_events);
void gbx_gen_abea_fill_many(uint64_t seed, int64_t first, int64_t n_reads, const float *level_mean, const float *level_stdv,
                            const int64_t *seq_off, const int64_t *event_off, char *seq, float *ev, float *scale, float *shift);
int hostcheck_poa_window(const gbx_poa_params *P, int n_seqs, const char *const *seqs, const int32_t *lens,
                         char *cons, int cons_cap, int ncap, int deg, int64_t *stats);
}

static int fails = 0;
#define CHECK(c, ...) do { if (!(c)) { fprintf(stderr, "san_main: " __VA_ARGS__); fputc('\n', stderr); ++fails; } } while (0)

static void run_bsw()
{
    const int64_t n = 1500;
    std::vector<int32_t> l1(n), l2(n), h0(n);
    gbx_gen_bsw_lengths(1001, 0, n, l1.data(), l2.data(), h0.data());
    std::vector<int64_t> idr(n), idq(n);
    int64_t rb = 0, qb = 0;
    for (int64_t k = 0; k < n; ++k) { idr[k] = rb; idq[k] = qb; rb += l1[k]; qb += l2[k]; }   // tight packing: no slack after a sequence
    std::vector<uint8_t> ref((size_t)rb), qer((size_t)qb);
    gbx_gen_bsw_fill(1001, 0, n, l1.data(), l2.data(), idr.data(), idq.data(), ref.data(), qer.data());
    gbx_bsw_params p;
    memset(&p, 0, sizeof(p));
    p.o_del = p.o_ins = 6; p.e_del = p.e_ins = 1; p.zdrop = 100; p.end_bonus = 5; p.w = 100;
    for (int i = 0; i < 5; ++i) for (int j = 0; j < 5; ++j) p.mat[i * 5 + j] = (i == 4 || j == 4) ? -1 : (i == j ? 1 : -4);
    std::vector<gbx_bsw_result> a(n), b(n);
    int64_t c1 = 0, c2 = 0;
    oracle_bsw_extend(&p, n, ref.data(), qer.data(), idr.data(), idq.data(), l1.data(), l2.data(), h0.data(), a.data(), 1, &c1);
    oracle_bsw_extend(&p, n, ref.data(), qer.data(), idr.data(), idq.data(), l1.data(), l2.data(), h0.data(), b.data(), 4, &c2);
    CHECK(c1 == c2 && !memcmp(a.data(), b.data(), sizeof(gbx_bsw_result) * (size_t)n), "bsw: 1 thread != 4 threads");
    // edge: 1x1, h0 = 0
    uint8_t one = 2;
    gbx_bsw_result r;
    oracle_bsw_extend_one(&p, 1, &one, 1, &one, 0, &r, nullptr);
    printf("bsw: %lld pairs, %lld in-band cells\n", (long long)n, (long long)c1);
}

static void run_chain()
{
    const int64_t nc = 12;
    std::vector<int64_t> cnt(nc), off(nc + 1, 0);
    gbx_gen_chain_counts_many(2001, 0, nc, cnt.data());
    for (int64_t c = 0; c < nc; ++c) off[c + 1] = off[c] + cnt[c];
    std::vector<uint64_t> ax((size_t)off[nc]), ay((size_t)off[nc]);
    gbx_gen_chain_fill_many(2001, 0, nc, off.data(), ax.data(), ay.data());
    std::vector<gbx_chain_call> hdr(nc);
    for (auto &h : hdr) { h.avg_qspan = 15.f; h.max_dist_x = h.max_dist_y = 5000; h.bw = 500; h.n_segs = 1; }
    const size_t na = (size_t)off[nc];
    std::vector<int32_t> s1(na), p1(na), t1(na), k1(na), s2(na), p2(na), t2(na), k2(na);
    int64_t e1 = 0, e2 = 0;
    oracle_chain(nc, off.data(), ax.data(), ay.data(), hdr.data(), s1.data(), p1.data(), t1.data(), k1.data(), 1, &e1);
    oracle_chain(nc, off.data(), ax.data(), ay.data(), hdr.data(), s2.data(), p2.data(), t2.data(), k2.data(), 3, &e2);
    CHECK(e1 == e2 && s1 == s2 && p1 == p2 && t1 == t2 && k1 == k2, "chain: 1 thread != 3 threads");
    printf("chain: %lld calls, %zu anchors, %lld evaluated pairs\n", (long long)nc, na, (long long)e1);
}

static void run_phmm()
{
    const int64_t nb = 6;
    std::vector<int32_t> nr(nb), nh(nb);
    gbx_gen_phmm_counts_many(3001, 0, nb, nr.data(), nh.data());
    std::vector<int64_t> roff(nb + 1, 0), hoff(nb + 1, 0);
    for (int64_t b = 0; b < nb; ++b) { roff[b + 1] = roff[b] + nr[b]; hoff[b + 1] = hoff[b] + nh[b]; }
    std::vector<int32_t> rl((size_t)roff[nb]), hl((size_t)hoff[nb]);
    gbx_gen_phmm_lengths_many(3001, 0, nb, roff.data(), hoff.data(), rl.data(), hl.data());
    std::vector<int64_t> ro(rl.size() + 1, 0), ho(hl.size() + 1, 0);
    for (size_t k = 0; k < rl.size(); ++k) ro[k + 1] = ro[k] + rl[k];
    for (size_t k = 0; k < hl.size(); ++k) ho[k + 1] = ho[k] + hl[k];
    std::vector<char> rs((size_t)ro.back()), q(rs.size()), qi(rs.size()), qd(rs.size()), qc(rs.size()), hap((size_t)ho.back());
    gbx_gen_phmm_fill_many(3001, 0, nb, roff.data(), hoff.data(), ro.data(), ho.data(), rs.data(), q.data(), qi.data(), qd.data(), qc.data(), hap.data());
    std::vector<int32_t> pr, ph;
    for (int64_t b = 0; b < nb; ++b)
        for (int r = 0; r < nr[b]; ++r)
            for (int h = 0; h < nh[b]; ++h) { pr.push_back((int32_t)(roff[b] + r)); ph.push_back((int32_t)(hoff[b] + h)); }
    std::vector<double> o1(pr.size()), o2(pr.size());
    int64_t nd1 = 0, nd2 = 0;
    oracle_phmm_init();
    oracle_phmm_forward((int64_t)pr.size(), pr.data(), ph.data(), ro.data(), rl.data(), rs.data(), q.data(), qi.data(), qd.data(), qc.data(),
                        ho.data(), hl.data(), hap.data(), o1.data(), 1, &nd1);
    oracle_phmm_forward((int64_t)pr.size(), pr.data(), ph.data(), ro.data(), rl.data(), rs.data(), q.data(), qi.data(), qd.data(), qc.data(),
                        ho.data(), hl.data(), hap.data(), o2.data(), 4, &nd2);
    CHECK(o1 == o2 && nd1 == nd2, "phmm: 1 thread != 4 threads");
    printf("phmm: %zu pairs, %lld fp64 redos\n", pr.size(), (long long)nd1);
}

static void run_poa()
{
    const int64_t nw = 5;
    std::vector<int32_t> nr(nw);
    gbx_gen_poa_counts_many(4001, 0, nw, nr.data());
    std::vector<int64_t> wf(nw + 1, 0);
    for (int64_t w = 0; w < nw; ++w) wf[w + 1] = wf[w] + nr[w];
    std::vector<int32_t> len((size_t)wf[nw]);
    std::vector<int64_t> off(len.size() + 1, 0);
    gbx_gen_poa_many(4001, 0, nw, 1, wf.data(), len.data(), off.data(), nullptr);
    for (size_t k = 0; k < len.size(); ++k) off[k + 1] = off[k] + len[k];
    std::vector<char> arena((size_t)off.back());
    gbx_gen_poa_many(4001, 0, nw, 2, wf.data(), len.data(), off.data(), arena.data());
    gbx_poa_params P;
    memset(&P, 0, sizeof(P));
    P.m = 2; P.n = -4; P.g = -6; P.e = -2; P.q = -25; P.c = -1;
    const int64_t stride = 1400;
    std::vector<char> cons((size_t)(nw * stride));
    std::vector<int32_t> clen((size_t)nw);
    int64_t cells = 0;
    oracle_poa_consensus(&P, nw, wf.data(), off.data(), len.data(), arena.data(), cons.data(), clen.data(), stride, 2, &cells);
    // the product's serial graph code (host build) on the same windows
    for (int64_t w = 0; w < nw; ++w) {
        std::vector<std::string> seqs;
        std::vector<const char *> ptr;
        std::vector<int32_t> ls;
        int64_t bases = 0;
        for (int64_t s = wf[w]; s < wf[w + 1]; ++s) { seqs.emplace_back(arena.data() + off[s], (size_t)len[s]); bases += len[s]; }
        for (auto &s : seqs) { ptr.push_back(s.data()); ls.push_back((int32_t)s.size()); }
        std::vector<char> c2((size_t)stride);
        int64_t st[2] = {0, 0};
        const int ncap = (int)(bases < 4000 ? bases + 8 : 4000);
        const int n = hostcheck_poa_window(&P, (int)seqs.size(), ptr.data(), ls.data(), c2.data(), (int)stride, ncap, (int)seqs.size(), st);
        CHECK(st[1] == 0, "poa hostcheck: window %lld error bits %lld", (long long)w, (long long)st[1]);
        CHECK(n == clen[(size_t)w] && !memcmp(c2.data(), cons.data() + w * stride, (size_t)n), "poa: product graph code != oracle in window %lld", (long long)w);
    }
    printf("poa: %lld windows, %lld cells\n", (long long)nw, (long long)cells);
}

static void run_abea()
{
    const int64_t n = 6;
    std::vector<float> lm(4096), ls(4096);
    gbx_gen_abea_model(5001, lm.data(), ls.data());
    std::vector<gbx_abea_model> model(4096);
    for (int k = 0; k < 4096; ++k) { model[k].level_mean = lm[k]; model[k].level_stdv = ls[k]; model[k].level_log_stdv = logf(ls[k]); }
    std::vector<int32_t> sl(n);
    std::vector<int64_t> ne(n), so(n + 1, 0), eo(n + 1, 0);
    gbx_gen_abea_counts_many(5001, 0, n, sl.data(), ne.data());
    for (int64_t r = 0; r < n; ++r) { so[r + 1] = so[r] + sl[r]; eo[r + 1] = eo[r] + ne[r]; }
    std::vector<char> seq((size_t)so[n] + 8);
    std::vector<float> ev((size_t)eo[n] + 4), sc(n), sh(n);
    gbx_gen_abea_fill_many(5001, 0, n, lm.data(), ls.data(), so.data(), eo.data(), seq.data(), ev.data(), sc.data(), sh.data());
    std::vector<gbx_abea_pair> o1((size_t)(2 * eo[n])), o2(o1.size());
    std::vector<int32_t> n1(n), n2(n);
    int64_t c1 = 0, c2 = 0;
    oracle_abea_align(n, so.data(), sl.data(), seq.data(), eo.data(), ev.data(), model.data(), sc.data(), sh.data(), o1.data(), n1.data(), 1, &c1);
    oracle_abea_align(n, so.data(), sl.data(), seq.data(), eo.data(), ev.data(), model.data(), sc.data(), sh.data(), o2.data(), n2.data(), 4, &c2);
    bool same = n1 == n2 && c1 == c2;
    int64_t aligned = 0;
    for (int64_t r = 0; r < n && same; ++r) {
        same = !memcmp(&o1[(size_t)(2 * eo[r])], &o2[(size_t)(2 * eo[r])], (size_t)n1[r] * sizeof(gbx_abea_pair));
        aligned += n1[r];
    }
    CHECK(same, "abea: 1 thread != 4 threads");
    CHECK(aligned > 0, "abea: no read aligned");
    printf("abea: %lld reads, %lld pairs, %lld cells\n", (long long)n, (long long)aligned, (long long)c1);
}

int main()
{
    run_bsw();
    run_chain();
    run_phmm();
    run_poa();
    run_abea();
    if (fails) { fprintf(stderr, "san_main: %d check(s) failed\n", fails); return 1; }
    printf("san_main: ok\n");
    return 0;
}
