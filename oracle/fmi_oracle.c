/* fmi_oracle.c — CPU restatement of the SMEM seeding the GenomicsBench `fmi` benchmark times.  TEST INFRASTRUCTURE ONLY.
 *
 * The driver is in the tree: R/benchmarks/fmi/fmi.cpp:180-286 calls, per batch of reads,
 *     FMI_search::getSMEMsAllPosOneThread        (:218)  SMEMs from every start position, min_intv 1
 *     a filter over those SMEMs                   (:230-240) + FMI_search::getSMEMsOnePosOneThread (:243)  re-seeding
 *     FMI_search::bwtSeedStrategyAllPosOneThread (:260)  "LAST" round, max_intv 20, minSeedLen + 1
 *     rid += batch offset (:270-273), FMI_search::sortSMEMs (:274)
 * FMI_search itself lives in tools/bwa-mem2, an EMPTY submodule (R/.gitmodules; bwa-mem2 upstream, commit unknown):
 * its methods are restated from bwa-mem2's published src/FMI_search.cpp / FMI_search.h (v2.x: CP_OCC checkpoints of 64
 * BWT symbols with one-hot words, backwardExt, the prev[] array swept backwards, compare_smem) - UPSTREAM KNOWLEDGE.
 *
 * Parity: UNPINNED by a compiled reference (the source is absent, nothing can be built).  What pins the arithmetic
 * independently of bwa-mem2: every reported (k, l, s) is checked against a brute-force suffix-array range of the
 * matched substring and of its reverse complement, and the round-1 SMEMs against a brute-force enumeration of the
 * super-maximal exact matches (tests/test_fmi_cpu.py).  What stays "as recalled": which SMEMs each round reports
 * (tie rules of the backward sweep, the next start position, the re-seeding filter - that one is in the tree).
 *
 * Index conventions (FMI_search::load_index): text = reference . reverse complement (codes 0..3) + sentinel;
 * ref_seq_len = text length incl. sentinel; SA row 0 is the sentinel suffix; count[c] = first row of base c;
 * cp_occ[i].cp_count[c] = occurrences of c in bwt[0, 64 i), bit 63 - j of one_hot_bwt_str[c] = (bwt[64 i + j] == c).
 */
#include <stdlib.h>
#include <string.h>
#include "gbx_oracle.h"

typedef gbx_fmi_smem SMEM;

/* GET_OCC + backwardExt (FMI_search.cpp): extend the match by base a to the left. */
static inline int64_t occ_of(const gbx_fmi_index *X, int64_t pp, int c)
{
    const int64_t id = pp >> 6, y = pp & 63;
    const uint64_t mask = y ? ~0ull << (64 - y) : 0ull;          /* one_hot_mask_array[y]: the top y bits */
    return X->cp_occ[id].cp_count[c] + (int64_t)__builtin_popcountll(X->cp_occ[id].one_hot_bwt_str[c] & mask);
}

static inline SMEM backward_ext(const gbx_fmi_index *X, SMEM smem, uint8_t a, int64_t *n_ext)
{
    int64_t k[4], l[4], s[4];
    for (int b = 0; b < 4; ++b) {
        const int64_t sp = smem.k, ep = smem.k + smem.s;
        const int64_t occ_sp = occ_of(X, sp, b), occ_ep = occ_of(X, ep, b);
        k[b] = X->count[b] + occ_sp;
        s[b] = occ_ep - occ_sp;
    }
    int64_t sentinel_offset = 0;
    if (smem.k <= X->sentinel_index && smem.k + smem.s > X->sentinel_index) sentinel_offset = 1;
    l[3] = smem.l + sentinel_offset;
    l[2] = l[3] + s[3];
    l[1] = l[2] + s[2];
    l[0] = l[1] + s[1];
    smem.k = k[a]; smem.l = l[a]; smem.s = s[a];
    if (n_ext) ++*n_ext;
    return smem;
}

/* forward extension = backward extension of the reverse complement (k and l swapped, base complemented) */
static inline SMEM forward_ext(const gbx_fmi_index *X, SMEM smem, uint8_t a, int64_t *n_ext)
{
    SMEM t = smem;
    t.k = smem.l; t.l = smem.k;
    SMEM r = backward_ext(X, t, (uint8_t)(3 - a), n_ext);
    SMEM o = r;
    o.k = r.l; o.l = r.k;
    return o;
}

/* getSMEMsOnePosOneThread for one (read, start position x, min_intv): appends to out, returns next_x */
static int smems_one_pos(const gbx_fmi_index *X, const uint8_t *q, int readlength, uint32_t rid, int x, int32_t min_intv,
                         int32_t min_seed_len, SMEM *prev, SMEM *out, int64_t *n_out, int64_t *n_ext)
{
    int next_x = x + 1;
    uint8_t a = q[x];
    if (a < 4) {
        SMEM smem;
        memset(&smem, 0, sizeof smem);
        smem.rid = rid; smem.m = (uint32_t)x; smem.n = (uint32_t)x;
        smem.k = X->count[a]; smem.l = X->count[3 - a]; smem.s = X->count[a + 1] - X->count[a];
        int numPrev = 0, j;
        for (j = x + 1; j < readlength; ++j) {
            a = q[j];
            next_x = j + 1;
            if (a < 4) {
                SMEM newSmem = forward_ext(X, smem, a, n_ext);
                newSmem.n = (uint32_t)j;
                const int s_neq_mask = newSmem.s != smem.s;
                prev[numPrev] = smem;
                numPrev += s_neq_mask;
                if (newSmem.s < min_intv) { next_x = j; break; }
                smem = newSmem;
            } else {
                break;
            }
        }
        if (smem.s >= min_intv) prev[numPrev++] = smem;
        for (int p = 0; p < numPrev / 2; ++p) { SMEM t = prev[p]; prev[p] = prev[numPrev - p - 1]; prev[numPrev - p - 1] = t; }

        /* backward search */
        for (j = x - 1; j >= 0; --j) {
            int numCurr = 0;
            int curr_s = -1;
            a = q[j];
            if (a > 3) break;
            int p;
            for (p = 0; p < numPrev; ++p) {
                SMEM sm = prev[p];
                SMEM newSmem = backward_ext(X, sm, a, n_ext);
                newSmem.m = (uint32_t)j;
                if (newSmem.s < min_intv && (int32_t)(sm.n - sm.m + 1) >= min_seed_len) {
                    out[(*n_out)++] = sm;
                    break;
                }
                if (newSmem.s >= min_intv && newSmem.s != curr_s) {
                    curr_s = (int)newSmem.s;
                    prev[numCurr++] = newSmem;
                    break;
                }
            }
            ++p;
            for (; p < numPrev; ++p) {
                SMEM sm = prev[p];
                SMEM newSmem = backward_ext(X, sm, a, n_ext);
                newSmem.m = (uint32_t)j;
                if (newSmem.s >= min_intv && newSmem.s != curr_s) {
                    curr_s = (int)newSmem.s;
                    prev[numCurr++] = newSmem;
                }
            }
            numPrev = numCurr;
            if (numCurr == 0) break;
        }
        if (numPrev != 0) {
            SMEM sm = prev[0];
            if ((int32_t)(sm.n - sm.m + 1) >= min_seed_len) out[(*n_out)++] = sm;
        }
    }
    return next_x;
}

/* bwtSeedStrategyAllPosOneThread for one read */
static void seed_strategy(const gbx_fmi_index *X, const uint8_t *q, int readlength, uint32_t rid, int32_t max_intv,
                          int32_t min_seed_len, SMEM *out, int64_t *n_out, int64_t *n_ext)
{
    int x = 0;
    while (x < readlength) {
        int next_x = x + 1;
        SMEM smem;
        memset(&smem, 0, sizeof smem);
        smem.rid = rid; smem.m = (uint32_t)x; smem.n = (uint32_t)x;
        uint8_t a = q[x];
        if (a < 4) {
            smem.k = X->count[a]; smem.l = X->count[3 - a]; smem.s = X->count[a + 1] - X->count[a];
            for (int j = x + 1; j < readlength; ++j) {
                next_x = j + 1;
                a = q[j];
                if (a < 4) {
                    SMEM newSmem = forward_ext(X, smem, a, n_ext);
                    newSmem.n = (uint32_t)j;
                    smem = newSmem;
                    if (smem.s < max_intv && (int32_t)(smem.n - smem.m + 1) >= min_seed_len) {
                        if (smem.s > 0) out[(*n_out)++] = smem;
                        break;
                    }
                } else {
                    break;
                }
            }
        }
        x = next_x;
    }
}

static int cmp_smem(const void *a, const void *b)              /* compare_smem: rid, m ascending, n descending */
{
    const SMEM *pa = (const SMEM *)a, *pb = (const SMEM *)b;
    if (pa->rid < pb->rid) return -1;
    if (pa->rid > pb->rid) return 1;
    if (pa->m < pb->m) return -1;
    if (pa->m > pb->m) return 1;
    if (pa->n > pb->n) return -1;
    if (pa->n < pb->n) return 1;
    return 0;
}

/* One read through the three rounds of fmi.cpp:218-278 (the rounds only combine SMEMs of one read, so the batch
 * loop of the driver and the round-major order inside a batch do not show in the sorted result).
 * out must hold 3 * readlength + 8 records; prev readlength + 1.  Returns the count. */
int64_t oracle_fmi_read(const gbx_fmi_index *X, const gbx_fmi_params *P, const uint8_t *q, int32_t readlength, uint32_t rid,
                        gbx_fmi_smem *out, gbx_fmi_smem *prev, int64_t *n_ext, int32_t *round_counts)
{
    int64_t n = 0;
    /* getSMEMsAllPosOneThread: start positions 0, next_x, ... with min_intv 1 */
    for (int x = 0; x < readlength;) x = smems_one_pos(X, q, readlength, rid, x, 1, P->min_seed_len, prev, out, &n, n_ext);
    const int64_t n1 = n;
    /* re-seeding (fmi.cpp:230-254) */
    for (int64_t j = 0; j < n1; ++j) {
        const SMEM *p = &out[j];
        const int start = (int)p->m, end = (int)p->n + 1;
        if (end - start < P->split_len || p->s > P->split_width) continue;
        smems_one_pos(X, q, readlength, rid, (end + start) >> 1, (int32_t)(p->s + 1), P->min_seed_len, prev, out, &n, n_ext);
    }
    const int64_t n2 = n;
    seed_strategy(X, q, readlength, rid, P->max_mem_intv, P->min_seed_len + 1, out, &n, n_ext);
    if (round_counts) { round_counts[0] = (int32_t)n1; round_counts[1] = (int32_t)(n2 - n1); round_counts[2] = (int32_t)(n - n2); }
    qsort(out, (size_t)n, sizeof(SMEM), cmp_smem);
    return n;
}

/* Whole job.  smem_off[n_reads + 1]; out_cap records at `out` (reads whose SMEMs do not fit are counted, not written).
 * Returns the total count. */
int64_t oracle_fmi_smem(const gbx_fmi_index *X, const gbx_fmi_params *P, int64_t n_reads, const uint8_t *enc,
                        const int64_t *read_off, const int32_t *read_len, gbx_fmi_smem *out, int64_t out_cap,
                        int64_t *smem_off, int nthreads, int64_t *n_ext_total, int64_t *round_totals)
{
    int64_t *cnt = (int64_t *)calloc((size_t)n_reads + 1, sizeof(int64_t));
    gbx_fmi_smem **per = (gbx_fmi_smem **)calloc((size_t)n_reads + 1, sizeof(void *));
    int64_t ext = 0, r0 = 0, r1 = 0, r2 = 0;
#pragma omp parallel for schedule(dynamic, 256) num_threads(nthreads > 0 ? nthreads : 1) reduction(+ : ext, r0, r1, r2)
    for (int64_t r = 0; r < n_reads; ++r) {
        const int32_t L = read_len[r];
        gbx_fmi_smem *tmp = (gbx_fmi_smem *)malloc(sizeof(gbx_fmi_smem) * (size_t)(4 * L + 16));
        gbx_fmi_smem *prev = tmp + 3 * L + 8;
        int32_t rc[3] = {0, 0, 0};
        int64_t e = 0;
        const int64_t n = L > 0 ? oracle_fmi_read(X, P, enc + read_off[r], L, (uint32_t)r, tmp, prev, &e, rc) : 0;
        cnt[r] = n;
        per[r] = (gbx_fmi_smem *)malloc(sizeof(gbx_fmi_smem) * (size_t)(n > 0 ? n : 1));
        memcpy(per[r], tmp, sizeof(gbx_fmi_smem) * (size_t)n);
        free(tmp);
        ext += e; r0 += rc[0]; r1 += rc[1]; r2 += rc[2];
    }
    int64_t total = 0;
    for (int64_t r = 0; r < n_reads; ++r) {
        if (smem_off) smem_off[r] = total;
        if (out && total + cnt[r] <= out_cap) memcpy(out + total, per[r], sizeof(gbx_fmi_smem) * (size_t)cnt[r]);
        total += cnt[r];
        free(per[r]);
    }
    if (smem_off) smem_off[n_reads] = total;
    if (n_ext_total) *n_ext_total = ext;
    if (round_totals) { round_totals[0] = r0; round_totals[1] = r1; round_totals[2] = r2; }
    free(cnt); free(per);
    return total;
}

/* FM-index of text[0, n) (codes 0..3; the caller passes reference . reverse complement) from its suffix array
 * (sa[0, n + 1): row 0 = the sentinel suffix n) - the tables FMI_search::load_index reads, built the plain way.
 * cp_occ must hold ((n + 1) >> 6) + 1 checkpoints. */
void oracle_fmi_build_index(const uint8_t *text, int64_t n, const int64_t *sa, gbx_fmi_cp_occ *cp_occ, int64_t *count5,
                            int64_t *sentinel_index)
{
    const int64_t n1 = n + 1, ncp = (n1 >> 6) + 1;
    int64_t c[4] = {0, 0, 0, 0};
    memset(cp_occ, 0, sizeof(gbx_fmi_cp_occ) * (size_t)ncp);
    for (int64_t i = 0; i < ncp * 64; ++i) {
        if ((i & 63) == 0) for (int b = 0; b < 4; ++b) cp_occ[i >> 6].cp_count[b] = c[b];
        if (i >= n1) continue;
        if (sa[i] == 0) { *sentinel_index = i; continue; }
        const int b = text[sa[i] - 1];
        cp_occ[i >> 6].one_hot_bwt_str[b] |= 1ull << (63 - (i & 63));
        ++c[b];
    }
    count5[0] = 1;
    for (int b = 0; b < 4; ++b) count5[b + 1] = count5[b] + c[b];
}
