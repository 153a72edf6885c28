/* datagen.c — seeded synthetic datasets in the reference benchmarks' input
 * shapes (the real input-datasets tarball, R/README.md:18, is not available
 * offline).  Tooling, not product and not oracle: plain C, deterministic per
 * (seed, item index) so any sub-range can be regenerated on any rank.
 * Distributions: SURVEY.md §8(d).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

typedef struct { uint64_t s; } rng_t;
static inline uint64_t mix64(uint64_t z)
{
    z += 0x9e3779b97f4a7c15ULL;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}
static inline void rng_seed(rng_t *r, uint64_t seed, uint64_t stream) { r->s = mix64(seed ^ mix64(stream * 0x632be59bd9b4e019ULL + 1)); }
static inline uint64_t rng_u64(rng_t *r) { r->s += 0x9e3779b97f4a7c15ULL; uint64_t z = r->s; z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL; z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL; return z ^ (z >> 31); }
static inline double rng_unif(rng_t *r) { return (double)(rng_u64(r) >> 11) * (1.0 / 9007199254740992.0); }
static inline uint32_t rng_below(rng_t *r, uint32_t n) { return (uint32_t)(((rng_u64(r) >> 32) * (uint64_t)n) >> 32); }
static inline double rng_norm(rng_t *r)
{
    double u1 = rng_unif(r), u2 = rng_unif(r);
    if (u1 < 1e-300) u1 = 1e-300;
    return sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
}

/* ------------------------------------------------------------------- bsw
 * Per simulated 151-bp read: seed length L = min(151, 19 + Exp(mean 40)),
 * seed start p ~ U[0, 151-L]; pair k is the left (k even) or right (k odd)
 * extension: qlen = p or 151-p-L (redrawn until > 0); h0 = L;
 * tlen = qlen + min(max(qlen-5,1), 200)  (bwa cal_max_gap with a=1,o=6,e=1,w=100).
 */
#define READ_LEN 151
static void bsw_lengths(uint64_t seed, int64_t k, int *qlen, int *tlen, int *h0)
{
    rng_t r;
    rng_seed(&r, seed, (uint64_t)k * 2 + 0);
    for (;;) {
        double u = rng_unif(&r);
        if (u < 1e-300) u = 1e-300;
        int L = 19 + (int)floor(-40.0 * log(u));
        if (L > READ_LEN) L = READ_LEN;
        int p = (int)rng_below(&r, (uint32_t)(READ_LEN - L + 1));
        int q = (k & 1) ? READ_LEN - p - L : p;
        if (q <= 0) continue;
        int gap = q - 5; if (gap < 1) gap = 1; if (gap > 200) gap = 200;
        *qlen = q; *tlen = q + gap; *h0 = L;
        return;
    }
}

void gbx_gen_bsw_lengths(uint64_t seed, int64_t first, int64_t n, int32_t *len1, int32_t *len2, int32_t *h0)
{
#pragma omp parallel for schedule(static)
    for (int64_t k = 0; k < n; ++k) {
        int q, t, h;
        bsw_lengths(seed, first + k, &q, &t, &h);
        len1[k] = t; len2[k] = q; h0[k] = h;
    }
}

/* query ~ U{0..3} (1 % of pairs get one ambiguous base 4); target = query with
 * 1 % substitutions, 0.05 % deletions, 0.15 % insertions, then a random tail. */
void gbx_gen_bsw_fill(uint64_t seed, int64_t first, int64_t n,
                      const int32_t *len1, const int32_t *len2,
                      const int64_t *idr, const int64_t *idq, uint8_t *ref, uint8_t *qer)
{
#pragma omp parallel for schedule(static)
    for (int64_t k = 0; k < n; ++k) {
        rng_t r;
        rng_seed(&r, seed, (uint64_t)(first + k) * 2 + 1);
        const int ql = len2[k], tl = len1[k];
        uint8_t *q = qer + idq[k], *t = ref + idr[k];
        for (int j = 0; j < ql; ++j) q[j] = (uint8_t)rng_below(&r, 4);
        if (rng_below(&r, 100) == 0) q[rng_below(&r, (uint32_t)ql)] = 4;
        int o = 0;
        for (int j = 0; j < ql && o < tl; ++j) {
            double u = rng_unif(&r);
            if (u < 0.0005) continue;                                   /* deletion from the target */
            uint8_t b = q[j] > 3 ? (uint8_t)rng_below(&r, 4) : q[j];
            if (u < 0.0105) b = (uint8_t)((b + 1 + rng_below(&r, 3)) & 3);   /* substitution */
            t[o++] = b;
            if (o < tl && rng_unif(&r) < 0.0015) t[o++] = (uint8_t)rng_below(&r, 4);   /* insertion */
        }
        while (o < tl) t[o++] = (uint8_t)rng_below(&r, 4);
    }
}

/* ----------------------------------------------------------------- chain
 * call c: n ~ LogNormal(median 3000, sigma 0.8) clipped to [50, 60000];
 * 1-4 colinear diagonals with jitter (dx ~ U[0,60], dy-dx ~ N(0,8)) + 20 % noise;
 * y = (15<<32) | qpos; anchors sorted by x (insertion into place by construction
 * of monotone x per diagonal then a merge by x).  avg_qspan 15, max_dist 5000, bw 500.
 */
int64_t gbx_gen_chain_count(uint64_t seed, int64_t call)
{
    rng_t r;
    rng_seed(&r, seed, (uint64_t)call * 2 + 0);
    double v = 3000.0 * exp(0.8 * rng_norm(&r));
    if (v < 50) v = 50;
    if (v > 60000) v = 60000;
    return (int64_t)v;
}

static int cmp_anchor(const void *a, const void *b)
{
    const uint64_t *x = (const uint64_t *)a, *y = (const uint64_t *)b;
    if (x[0] != y[0]) return x[0] < y[0] ? -1 : 1;
    if (x[1] != y[1]) return x[1] < y[1] ? -1 : 1;
    return 0;
}

#include <stdlib.h>
#include <stdio.h>
void gbx_gen_chain_fill(uint64_t seed, int64_t call, int64_t n, uint64_t *ax, uint64_t *ay)
{
    rng_t r;
    rng_seed(&r, seed, (uint64_t)call * 2 + 1);
    uint64_t *tmp = (uint64_t *)malloc(sizeof(uint64_t) * 2 * (size_t)n);
    const int ndiag = 1 + (int)rng_below(&r, 4);
    const int64_t n_noise = n / 5, n_sig = n - n_noise;
    const int64_t span = n * 30 + 1000;                 /* read/ref extent */
    int64_t o = 0;
    for (int d = 0; d < ndiag; ++d) {
        int64_t cnt = n_sig / ndiag + (d < n_sig % ndiag ? 1 : 0);
        int64_t x = 1000 + (int64_t)rng_below(&r, 2000) + (int64_t)d * 7919;
        int64_t y = 100 + (int64_t)rng_below(&r, 500);
        for (int64_t k = 0; k < cnt; ++k) {
            int64_t dx = (int64_t)rng_below(&r, 61);
            int64_t dy = dx + (int64_t)llround(8.0 * rng_norm(&r));
            if (dy < 0) dy = 0;
            x += dx; y += dy;
            tmp[2 * o] = (uint64_t)x;
            tmp[2 * o + 1] = ((uint64_t)15 << 32) | (uint64_t)(uint32_t)(y & 0x7fffffff);
            ++o;
        }
    }
    for (int64_t k = 0; k < n_noise; ++k) {
        tmp[2 * o] = (uint64_t)(1000 + (int64_t)(rng_unif(&r) * (double)span));
        tmp[2 * o + 1] = ((uint64_t)15 << 32) | (uint64_t)(uint32_t)(100 + (int64_t)(rng_unif(&r) * (double)span));
        ++o;
    }
    qsort(tmp, (size_t)n, 2 * sizeof(uint64_t), cmp_anchor);
    for (int64_t k = 0; k < n; ++k) { ax[k] = tmp[2 * k]; ay[k] = tmp[2 * k + 1]; }
    free(tmp);
}

/* chain, "realistic" structure (beside the SURVEY 8d workload above, which keeps one strand and one reference id
 * and therefore has one dense run of anchors per call): what minimap2 hands mm_chain_dp for a long read against a
 * genome - x = strand << 63 | rid << 32 | rpos (host_data.h; compared as one 64-bit word, host_kernel.cpp:56,59) with
 * six reference sequences of 14-21 Mbp on both strands; 55 % of a call's anchors lie on the read's true locus (1-2
 * colinear diagonals with the jitter of the plain generator), 25 % on 2-6 partial repeat copies elsewhere (shorter
 * colinear runs on random strands / references), 20 % are isolated minimizer hits spread over the whole genome.
 * Sorted by x the call falls apart into pieces further than max_dist_x from each other. */
void gbx_gen_chain_fill_real(uint64_t seed, int64_t call, int64_t n, uint64_t *ax, uint64_t *ay)
{
    static const int64_t REF_LEN[6] = {15072434, 15279421, 13783801, 17493829, 20924180, 17718942};   /* C. elegans I-V, X */
    rng_t r;
    rng_seed(&r, seed, (uint64_t)call * 2 + 1);
    uint64_t *tmp = (uint64_t *)malloc(sizeof(uint64_t) * 2 * (size_t)n);
    const int64_t n_noise = n / 5, n_rep = n / 4, n_true = n - n_noise - n_rep;
    const int64_t qspan_total = n_true * 30 + 1000;      /* the read's extent in query coordinates */
    int64_t o = 0;
    /* a colinear run of cnt anchors starting at (strand, rid, x0, y0) */
#define RUN(cnt_, key_, x0_, y0_) do { int64_t x_ = (x0_), y_ = (y0_); \
        for (int64_t k_ = 0; k_ < (cnt_); ++k_) { \
            int64_t dx_ = (int64_t)rng_below(&r, 61), dy_ = dx_ + (int64_t)llround(8.0 * rng_norm(&r)); \
            if (dy_ < 0) dy_ = 0; \
            x_ += dx_; y_ += dy_; \
            tmp[2 * o] = (key_) | (uint64_t)(x_ & 0x7fffffff); \
            tmp[2 * o + 1] = ((uint64_t)15 << 32) | (uint64_t)(uint32_t)(y_ & 0x7fffffff); \
            ++o; } } while (0)
    {
        const int rid = (int)rng_below(&r, 6), strand = (int)rng_below(&r, 2);
        const uint64_t key = ((uint64_t)strand << 63) | ((uint64_t)rid << 32);
        const int64_t room = REF_LEN[rid] - qspan_total - 20000;
        const int64_t pos = 1000 + (int64_t)(rng_unif(&r) * (double)(room > 1 ? room : 1));
        const int ndiag = 1 + (int)rng_below(&r, 2);
        for (int d = 0; d < ndiag; ++d) {
            const int64_t cnt = n_true / ndiag + (d < n_true % ndiag ? 1 : 0);
            RUN(cnt, key, pos + (int64_t)rng_below(&r, 2000) + (int64_t)d * 7919, 100 + (int64_t)rng_below(&r, 500));
        }
    }
    {
        const int ncopy = 2 + (int)rng_below(&r, 5);
        for (int c = 0; c < ncopy; ++c) {
            const int64_t cnt = n_rep / ncopy + (c < n_rep % ncopy ? 1 : 0);
            const int rid = (int)rng_below(&r, 6), strand = (int)rng_below(&r, 2);
            const uint64_t key = ((uint64_t)strand << 63) | ((uint64_t)rid << 32);
            const int64_t room = REF_LEN[rid] - cnt * 30 - 20000;
            const int64_t pos = 1000 + (int64_t)(rng_unif(&r) * (double)(room > 1 ? room : 1));
            RUN(cnt, key, pos, 100 + (int64_t)(rng_unif(&r) * (double)qspan_total));
        }
    }
    for (int64_t k = 0; k < n_noise; ++k) {
        const int rid = (int)rng_below(&r, 6), strand = (int)rng_below(&r, 2);
        tmp[2 * o] = ((uint64_t)strand << 63) | ((uint64_t)rid << 32) | (uint64_t)(1000 + (int64_t)(rng_unif(&r) * (double)(REF_LEN[rid] - 2000)));
        tmp[2 * o + 1] = ((uint64_t)15 << 32) | (uint64_t)(uint32_t)(100 + (int64_t)(rng_unif(&r) * (double)qspan_total));
        ++o;
    }
#undef RUN
    qsort(tmp, (size_t)n, 2 * sizeof(uint64_t), cmp_anchor);
    for (int64_t k = 0; k < n; ++k) { ax[k] = tmp[2 * k]; ay[k] = tmp[2 * k + 1]; }
    free(tmp);
}

/* ------------------------------------------------------------------ phmm
 * batch b: num_reads ~ U[1,120], num_haps ~ U[2,16]; a backbone of U[150,450]
 * bases; every haplotype = backbone with 1 % SNPs and 0.5 % single-base indels;
 * reads are 151 bp (10 % shorter, U[30,151]) sampled from a random haplotype
 * with 1 % substitution errors; 0.1 % 'N' everywhere; q ~ U[2,41] (stored
 * already normalised as the driver does: minus 33, q clamped >= 6,
 * PairHMMUnitTest.cpp:89-93,110-113), i,d ~ U[40,50], c = 10.
 * mode 0: counts only; mode 1: + lengths; mode 2: + bytes.
 */
static const char BASES[4] = {'A', 'C', 'G', 'T'};
void gbx_gen_phmm_batch(uint64_t seed, int64_t batch, int mode, int32_t *n_reads, int32_t *n_haps,
                        int32_t *read_len, int32_t *hap_len,
                        char *rs, char *q, char *qi, char *qd, char *qc, char *hap)
{
    rng_t r;
    rng_seed(&r, seed, (uint64_t)batch);
    const int nr = 1 + (int)rng_below(&r, 120), nh = 2 + (int)rng_below(&r, 15);
    *n_reads = nr; *n_haps = nh;
    if (mode == 0) return;
    char backbone[512];
    const int lb = 150 + (int)rng_below(&r, 301);
    for (int k = 0; k < lb; ++k) backbone[k] = BASES[rng_below(&r, 4)];
    char haps[16][480];
    int hl[16];
    for (int h = 0; h < nh; ++h) {
        int o = 0;
        for (int k = 0; k < lb && o < 470; ++k) {
            double u = rng_unif(&r);
            if (u < 0.0025) continue;                                  /* deletion */
            char b = backbone[k];
            if (u < 0.0125) b = BASES[rng_below(&r, 4)];               /* SNP (may be silent) */
            if (rng_below(&r, 1000) == 0) b = 'N';
            haps[h][o++] = b;
            if (rng_unif(&r) < 0.0025) haps[h][o++] = BASES[rng_below(&r, 4)];   /* insertion */
        }
        if (o < 8) { while (o < 8) haps[h][o++] = BASES[rng_below(&r, 4)]; }
        hl[h] = o; hap_len[h] = o;
        if (mode == 2) { memcpy(hap, haps[h], (size_t)o); hap += o; }
    }
    for (int k = 0; k < nr; ++k) {
        const int h = (int)rng_below(&r, (uint32_t)nh);
        int len = rng_below(&r, 10) == 0 ? 30 + (int)rng_below(&r, 122) : 151;
        if (len > hl[h]) len = hl[h];
        const int start = (int)rng_below(&r, (uint32_t)(hl[h] - len + 1));
        read_len[k] = len;
        for (int j = 0; j < len; ++j) {
            char b = haps[h][start + j];
            const uint32_t e = rng_below(&r, 1000);
            if (e < 10) b = BASES[rng_below(&r, 4)];
            else if (e == 10) b = 'N';
            int qq = 2 + (int)rng_below(&r, 40); if (qq < 6) qq = 6;
            const int ii = 40 + (int)rng_below(&r, 11), dd = 40 + (int)rng_below(&r, 11);
            if (mode == 2) { rs[j] = b; q[j] = (char)qq; qi[j] = (char)ii; qd[j] = (char)dd; qc[j] = 10; }
        }
        if (mode == 2) { rs += len; q += len; qi += len; qd += len; qc += len; }
    }
}

/* ------------------------------------------------------------------- poa
 * window w: a 500-bp backbone ~U{ACGT}; 20-40 reads = backbone with 6 %
 * substitutions, 4 % insertions, 4 % deletions, each read trimmed by up to 30
 * bases at either end (lengths ~ 450-560).  mode 0: count; 1: lengths; 2: bytes.
 */
void gbx_gen_poa_window(uint64_t seed, int64_t window, int mode, int32_t *n_reads, int32_t *read_len, char *out)
{
    rng_t r;
    rng_seed(&r, seed, (uint64_t)window);
    const int nr = 20 + (int)rng_below(&r, 21);
    *n_reads = nr;
    if (mode == 0) return;
    char backbone[500];
    for (int k = 0; k < 500; ++k) backbone[k] = BASES[rng_below(&r, 4)];
    char buf[1024];
    for (int k = 0; k < nr; ++k) {
        const int a = (int)rng_below(&r, 31), b = 500 - (int)rng_below(&r, 31);
        int o = 0;
        for (int j = a; j < b && o < 1000; ++j) {
            const uint32_t u = rng_below(&r, 100);
            if (u < 4) continue;                                             /* deletion */
            char ch = backbone[j];
            if (u < 10) ch = BASES[(rng_below(&r, 3) + 1 + (uint32_t)(strchr("ACGT", ch) - "ACGT")) & 3];   /* substitution */
            buf[o++] = ch;
            if (rng_below(&r, 100) < 4) buf[o++] = BASES[rng_below(&r, 4)];  /* insertion */
        }
        read_len[k] = o;
        if (mode == 2) { memcpy(out, buf, (size_t)o); out += o; }
    }
}

/* ---------------------------------------------------------- batched fills
 * The per-item generators above are deterministic per (seed, item), so whole
 * ranges are filled with one OpenMP loop (a multi-GPU bench generates every
 * rank's shard on rank 0 before scattering it).
 */
void gbx_gen_chain_fill_many(uint64_t seed, int64_t first, int64_t n_calls, const int64_t *off, uint64_t *ax, uint64_t *ay)
{
#pragma omp parallel for schedule(dynamic, 8)
    for (int64_t c = 0; c < n_calls; ++c)
        gbx_gen_chain_fill(seed, first + c, off[c + 1] - off[c], ax + off[c], ay + off[c]);
}

void gbx_gen_chain_fill_real_many(uint64_t seed, int64_t first, int64_t n_calls, const int64_t *off, uint64_t *ax, uint64_t *ay)
{
#pragma omp parallel for schedule(dynamic, 8)
    for (int64_t c = 0; c < n_calls; ++c)
        gbx_gen_chain_fill_real(seed, first + c, off[c + 1] - off[c], ax + off[c], ay + off[c]);
}

void gbx_gen_phmm_counts_many(uint64_t seed, int64_t first, int64_t n_batches, int32_t *n_reads, int32_t *n_haps)
{
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < n_batches; ++b)
        gbx_gen_phmm_batch(seed, first + b, 0, n_reads + b, n_haps + b, 0, 0, 0, 0, 0, 0, 0, 0);
}

/* roff/hoff: first read / haplotype index of each batch (n_batches+1 entries). */
void gbx_gen_phmm_lengths_many(uint64_t seed, int64_t first, int64_t n_batches, const int64_t *roff, const int64_t *hoff,
                               int32_t *read_len, int32_t *hap_len)
{
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t b = 0; b < n_batches; ++b) {
        int32_t nr, nh;
        gbx_gen_phmm_batch(seed, first + b, 1, &nr, &nh, read_len + roff[b], hap_len + hoff[b], 0, 0, 0, 0, 0, 0);
    }
}

/* read_off/hap_off: byte offset of each read / haplotype in its arena. */
void gbx_gen_phmm_fill_many(uint64_t seed, int64_t first, int64_t n_batches, const int64_t *roff, const int64_t *hoff,
                            const int64_t *read_off, const int64_t *hap_off,
                            char *rs, char *q, char *qi, char *qd, char *qc, char *hap)
{
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t b = 0; b < n_batches; ++b) {
        int32_t nr, nh, tl[128], th[16];
        const int64_t ro = read_off[roff[b]], ho = hap_off[hoff[b]];
        gbx_gen_phmm_batch(seed, first + b, 2, &nr, &nh, tl, th, rs + ro, q + ro, qi + ro, qd + ro, qc + ro, hap + ho);
    }
}

void gbx_gen_poa_counts_many(uint64_t seed, int64_t first, int64_t n_windows, int32_t *n_reads)
{
#pragma omp parallel for schedule(static)
    for (int64_t w = 0; w < n_windows; ++w) gbx_gen_poa_window(seed, first + w, 0, n_reads + w, 0, 0);
}

/* wf: first sequence of each window (n_windows+1); mode 1 fills seq_len, mode 2 the bytes at seq_off[wf[w]]. */
void gbx_gen_poa_many(uint64_t seed, int64_t first, int64_t n_windows, int mode, const int64_t *wf,
                      int32_t *seq_len, const int64_t *seq_off, char *arena)
{
#pragma omp parallel for schedule(dynamic, 16)
    for (int64_t w = 0; w < n_windows; ++w) {
        int32_t nr, tl[64];
        if (mode == 1) gbx_gen_poa_window(seed, first + w, 1, &nr, seq_len + wf[w], 0);
        else gbx_gen_poa_window(seed, first + w, 2, &nr, tl, arena + seq_off[wf[w]]);
    }
}

void gbx_gen_chain_counts_many(uint64_t seed, int64_t first, int64_t n_calls, int64_t *counts)
{
#pragma omp parallel for schedule(static)
    for (int64_t c = 0; c < n_calls; ++c) counts[c] = gbx_gen_chain_count(seed, first + c);
}

/* ------------------------------------------------------------------ abea
 * Synthetic pore model and reads with events, in the shapes f5c's align() consumes (R/benchmarks/abea/src/f5c.h:
 * model_t :122, event_t :104, scalings_t :139).  The real r9.4 6-mer model is a table in the reference's source tree
 * and real signal needs fast5 files; neither is used: the algorithm only asks for 4096 (mean, stdv) levels.
 * model state s: level_mean ~ U[65,125] pA, level_stdv ~ U[1.2,3.2].
 * read r: length ~ LogNormal(median 6000, sigma 0.6) clipped to [400, 40000] bases ~U{ACGT}; per k-mer the number of
 * events is 0 (skip, 3 %), else 1 + Geometric(0.45) (stays); event mean = scale*level_mean + shift + N(0, level_stdv);
 * scalings: shift ~ N(0, 4), scale ~ U[0.94, 1.06] (what estimate_scalings would have found: the generator hands
 * them over exact).  mode 0: (seq_len, n_events); mode 1: fill.
 */
void gbx_gen_abea_model(uint64_t seed, float *level_mean, float *level_stdv)
{
    rng_t r;
    rng_seed(&r, seed, 0xabeaULL);
    for (int s = 0; s < 4096; ++s) {
        level_mean[s] = (float)(65.0 + 60.0 * rng_unif(&r));
        level_stdv[s] = (float)(1.2 + 2.0 * rng_unif(&r));
    }
}

static void abea_read(uint64_t seed, int64_t read, int mode, const float *level_mean, const float *level_stdv,
                      int32_t *seq_len, int64_t *n_events, char *seq, float *ev, float *scale, float *shift)
{
    rng_t r;
    rng_seed(&r, seed, (uint64_t)read * 2 + 1);
    double v = 6000.0 * exp(0.6 * rng_norm(&r));
    if (v < 400) v = 400;
    if (v > 40000) v = 40000;
    const int len = (int)v;
    const float sc = (float)(0.94 + 0.12 * rng_unif(&r)), sh = (float)(4.0 * rng_norm(&r));
    *seq_len = len;
    if (mode == 1) { *scale = sc; *shift = sh; }
    /* the sequence and the event counts come from one stream so that both modes agree */
    uint32_t rank = 0;
    int64_t ne = 0;
    for (int i = 0; i < len; ++i) {
        const uint32_t b = rng_below(&r, 4);
        if (mode == 1) seq[i] = "ACGT"[b];
        rank = ((rank << 2) | b) & 4095u;                 /* k-mer ending at base i: first base is the most significant */
        if (i < 5) continue;
        int cnt = 0;
        if (rng_below(&r, 100) >= 3) { cnt = 1; while (rng_unif(&r) < 0.45 && cnt < 12) ++cnt; }
        for (int c = 0; c < cnt; ++c) {
            const double noise = rng_norm(&r);
            if (mode == 1) ev[ne] = (float)((double)sc * level_mean[rank] + (double)sh + noise * level_stdv[rank]);
            ++ne;
        }
    }
    if (ne == 0) { if (mode == 1) ev[0] = sc * level_mean[rank] + sh; ne = 1; }
    *n_events = ne;
}

void gbx_gen_abea_counts_many(uint64_t seed, int64_t first, int64_t n_reads, int32_t *seq_len, int64_t *n_events)
{
#pragma omp parallel for schedule(dynamic, 8)
    for (int64_t k = 0; k < n_reads; ++k) abea_read(seed, first + k, 0, 0, 0, seq_len + k, n_events + k, 0, 0, 0, 0);
}

void gbx_gen_abea_fill_many(uint64_t seed, int64_t first, int64_t n_reads, const float *level_mean, const float *level_stdv,
                            const int64_t *seq_off, const int64_t *event_off, char *seq, float *ev, float *scale, float *shift)
{
#pragma omp parallel for schedule(dynamic, 8)
    for (int64_t k = 0; k < n_reads; ++k) {
        int32_t l; int64_t ne;
        abea_read(seed, first + k, 1, level_mean, level_stdv, &l, &ne, seq + seq_off[k], ev + event_off[k], scale + k, shift + k);
    }
}

/* ------------------------------------------------------------------- fmi
 * A synthetic genome and short reads for the FM-index seeding benchmark (the reference runs 151-bp reads of
 * SRR7733443 against the index of a human reference, R/scripts/run-cpu.sh:27,58; neither is in the image).
 * Genome: uniform bases, then `n_fam` repeat families: a family is a random element of 150-3000 bases copied to
 * 3-30 random places (either strand) with 0-8 % divergence per copy - the source of SMEMs with more than one hit
 * and of the re-seeding round; 0.2 % of the genome is low-complexity (dinucleotide runs of 30-200 bases).
 * Reads: `read_len` bases from a uniform position and strand, 1 % substitutions, 0.02 % of the bases read 'N' (4),
 * 2 % of the reads carry one 1-3 base deletion; 1 % of the reads are random sequence (unmappable).
 * Deterministic per (seed, read index).
 */
void gbx_gen_fmi_genome(uint64_t seed, int64_t len, uint8_t *ref)
{
    const int64_t chunk = 1 << 20;
#pragma omp parallel for schedule(static)
    for (int64_t c = 0; c < (len + chunk - 1) / chunk; ++c) {
        rng_t r;
        rng_seed(&r, seed, 0xf31000000ULL + (uint64_t)c);
        const int64_t hi = (c + 1) * chunk < len ? (c + 1) * chunk : len;
        for (int64_t i = c * chunk; i < hi;) {
            uint64_t w = rng_u64(&r);
            for (int k = 0; k < 32 && i < hi; ++k, ++i, w >>= 2) ref[i] = (uint8_t)(w & 3);
        }
    }
    rng_t r;
    rng_seed(&r, seed, 0xf32ULL);
    const int64_t n_fam = len / 40000 + 1;                 /* ~ 4 % of the genome in repeats */
    uint8_t elem[3000];
    for (int64_t f = 0; f < n_fam; ++f) {
        const int el = 150 + (int)rng_below(&r, 2851);
        if (el + 16 >= len) break;
        for (int k = 0; k < el; ++k) elem[k] = (uint8_t)rng_below(&r, 4);
        const int copies = 3 + (int)rng_below(&r, 28);
        for (int c = 0; c < copies; ++c) {
            const int64_t pos = (int64_t)(rng_unif(&r) * (double)(len - el));
            const double div = 0.08 * rng_unif(&r) * rng_unif(&r);
            const int rev = (int)rng_below(&r, 2);
            for (int k = 0; k < el; ++k) {
                uint8_t b = rev ? (uint8_t)(3 - elem[el - 1 - k]) : elem[k];
                if (rng_unif(&r) < div) b = (uint8_t)((b + 1 + rng_below(&r, 3)) & 3);
                ref[pos + k] = b;
            }
        }
    }
    const int64_t n_low = len / 50000;
    for (int64_t f = 0; f < n_low; ++f) {
        const int el = 30 + (int)rng_below(&r, 171);
        if (el + 16 >= len) break;
        const int64_t pos = (int64_t)(rng_unif(&r) * (double)(len - el));
        const uint8_t a = (uint8_t)rng_below(&r, 4), b = (uint8_t)rng_below(&r, 4);
        for (int k = 0; k < el; ++k) ref[pos + k] = (k & 1) ? b : a;
    }
}

void gbx_gen_fmi_reads(uint64_t seed, int64_t first, int64_t n_reads, const uint8_t *ref, int64_t len, int32_t read_len, uint8_t *enc)
{
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n_reads; ++i) {
        rng_t r;
        rng_seed(&r, seed, 0xf33000000000ULL + (uint64_t)(first + i));
        uint8_t *q = enc + i * (int64_t)read_len;
        if (rng_below(&r, 100) == 0 || len < read_len + 8) {
            for (int k = 0; k < read_len; ++k) q[k] = (uint8_t)rng_below(&r, 4);
            continue;
        }
        const int64_t pos = (int64_t)(rng_unif(&r) * (double)(len - read_len - 4));
        const int rev = (int)rng_below(&r, 2);
        const int del_at = rng_below(&r, 50) == 0 ? 10 + (int)rng_below(&r, (uint32_t)(read_len - 20)) : -1;
        const int del_len = 1 + (int)rng_below(&r, 3);
        int64_t src = pos;
        for (int k = 0; k < read_len; ++k, ++src) {
            if (k == del_at) src += del_len;
            uint8_t b = ref[src < len ? src : len - 1];
            const double u = rng_unif(&r);
            if (u < 0.01) b = (uint8_t)((b + 1 + rng_below(&r, 3)) & 3);
            else if (u < 0.0102) b = 4;
            q[k] = b;
        }
        if (rev)
            for (int a = 0, b = read_len - 1; a <= b; ++a, --b) {
                const uint8_t x = q[a], y = q[b];
                q[a] = y > 3 ? y : (uint8_t)(3 - y);
                q[b] = x > 3 ? x : (uint8_t)(3 - x);
            }
    }
}

/* ------------------------------------------------------ input files (bench)
 * The reference's bsw input format (main_banded.cpp:131-141: seed score, target digits, query digits, one line each),
 * written from the generated arrays with one buffered writer: bench.py times the driver end to end on this file.
 * Returns bytes written, or -1.
 */
int64_t gbx_write_bsw_pairs(const char *path, int64_t n, const uint8_t *ref, const int64_t *idr, const int32_t *len1,
                            const uint8_t *qer, const int64_t *idq, const int32_t *len2, const int32_t *h0)
{
    FILE *f = fopen(path, "wb");
    if (!f) return -1;
    static char iobuf[1 << 22];
    setvbuf(f, iobuf, _IOFBF, sizeof(iobuf));
    char line[1 << 16];
    int64_t total = 0;
    for (int64_t k = 0; k < n; ++k) {
        int o = snprintf(line, 32, "%d\n", h0[k]);
        if (len1[k] + len2[k] + o + 4 > (int)sizeof(line)) { fclose(f); return -1; }
        for (int l = 0; l < len1[k]; ++l) line[o++] = (char)(ref[idr[k] + l] + 48);
        line[o++] = '\n';
        for (int l = 0; l < len2[k]; ++l) line[o++] = (char)(qer[idq[k] + l] + 48);
        line[o++] = '\n';
        if (fwrite(line, 1, (size_t)o, f) != (size_t)o) { fclose(f); return -1; }
        total += o;
    }
    return fclose(f) == 0 ? total : -1;
}
