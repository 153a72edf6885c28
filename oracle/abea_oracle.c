/* abea_oracle.c — CPU restatement of f5c / nanopolish adaptive banded event alignment as the GenomicsBench
 * `abea` benchmark's CPU path runs it.  TEST INFRASTRUCTURE ONLY.
 *
 * Follows align(), R/benchmarks/abea/src/align.c:169-548 (called per read by align_single, f5c.c:1346), with its
 * helpers log_normal_pdf (:99-106), log_probability_match_r9 (:108-148, CACHED_LOG is defined, f5c.h:67) and
 * get_kmer_rank (:27-38).  The arithmetic types are the reference's: band scores and emissions are float, the
 * transition penalties lp_skip / lp_stay / lp_step / lp_trim are double, so every candidate score is a double sum
 * rounded to float (:371-373), and the backtrack sums the emissions in double (:455).
 *
 * Parity: UNPINNED by a compiled reference - align.c includes f5c.h, which needs htslib and HDF5 headers the image
 * lacks (R/benchmarks/abea/src/f5c.h:11-15), so the reference translation unit cannot be built here without
 * stand-ins.  The source it restates is in the tree (unlike phmm / poa), line by line above.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "gbx_oracle.h"

#define BW GBX_ABEA_BANDWIDTH
#define KSZ GBX_ABEA_KMER

static inline uint32_t base_rank(char b) { return b == 'C' ? 1u : b == 'G' ? 2u : b == 'T' ? 3u : 0u; }   /* :10-24, non-ACGT -> 0 */
static inline uint32_t kmer_rank(const char *s)                                                            /* :27-38 */
{
    uint32_t r = 0;
    for (uint32_t i = 0; i < KSZ; ++i) r += base_rank(s[KSZ - i - 1]) << (i << 1);
    return r;
}

static inline float lp_match(float scale, float shift, const gbx_abea_model *m, float x, uint32_t rank)     /* :99-148 */
{
    const float gp_mean = scale * m[rank].level_mean + shift;
    const float gp_stdv = m[rank].level_stdv * 1;
    const float gp_log_stdv = m[rank].level_log_stdv;
    const float log_inv_sqrt_2pi = -0.918938f;
    const float a = (x - gp_mean) / gp_stdv;
    return log_inv_sqrt_2pi - gp_log_stdv + (-0.5f * a * a);
}

int32_t oracle_abea_align_one(const char *sequence, int32_t sequence_len, const float *event_mean, int64_t n_events_in,
                              const gbx_abea_model *models, float scale, float shift, gbx_abea_pair *out, int64_t *cells)
{
    const size_t n_events = (size_t)n_events_in, n_kmers = (size_t)(sequence_len - KSZ + 1);
    const uint8_t FROM_D = 0, FROM_U = 1, FROM_L = 2;
    const double min_average_log_emission = -5.0;
    const int max_gap_threshold = 50;
    const int bandwidth = BW, half_bandwidth = BW / 2;
    const double events_per_kmer = (double)n_events / n_kmers;                                     /* :195-205 */
    const double p_stay = 1 - (1 / (events_per_kmer + 1));
    const double epsilon = 1e-10;
    const double lp_skip = log(epsilon), lp_stay = log(p_stay);
    const double lp_step = log(1.0 - exp(lp_skip) - exp(lp_stay));
    const double lp_trim = log(0.01);
    const size_t n_rows = n_events + 1, n_cols = n_kmers + 1, n_bands = n_rows + n_cols;

    uint32_t *ranks = (uint32_t *)malloc(sizeof(uint32_t) * n_kmers);
    for (size_t i = 0; i < n_kmers; ++i) ranks[i] = kmer_rank(sequence + i);
    float *bands = (float *)malloc(sizeof(float) * n_bands * BW);
    uint8_t *trace = (uint8_t *)malloc(n_bands * BW);
    int *bl_e = (int *)malloc(sizeof(int) * n_bands), *bl_k = (int *)malloc(sizeof(int) * n_bands);
    for (size_t i = 0; i < n_bands * BW; ++i) { bands[i] = -INFINITY; trace[i] = 0; }
#define BAND(r, c) bands[(size_t)(r) * BW + (c)]
#define TRACE(r, c) trace[(size_t)(r) * BW + (c)]
#define VALID(o) ((o) >= 0 && (o) < bandwidth)
    bl_e[0] = half_bandwidth - 1; bl_k[0] = -1 - half_bandwidth;                                    /* :264-266 */
    bl_e[1] = bl_e[0] + 1; bl_k[1] = bl_k[0];
    BAND(0, -1 - bl_k[0]) = 0.0f;                                                                   /* :268-271 */
    {
        const int first_trim_offset = bl_e[1] - 0;                                                  /* :274-278 */
        BAND(1, first_trim_offset) = (float)lp_trim;
        TRACE(1, first_trim_offset) = FROM_U;
    }
    int64_t fills = 0;
    for (size_t b = 2; b < n_bands; ++b) {                                                          /* :287-404 */
        const float ll = BAND(b - 1, 0), ur = BAND(b - 1, bandwidth - 1);
        const int ll_ob = ll == -INFINITY, ur_ob = ur == -INFINITY;
        int right;
        if (ll_ob && ur_ob) right = b % 2 == 1; else right = ll < ur;                               /* Suzuki's rule */
        if (right) { bl_e[b] = bl_e[b - 1]; bl_k[b] = bl_k[b - 1] + 1; }
        else { bl_e[b] = bl_e[b - 1] + 1; bl_k[b] = bl_k[b - 1]; }
        const int trim_offset = -1 - bl_k[b];                                                       /* :310-319 */
        if (VALID(trim_offset)) {
            const int64_t event_idx = bl_e[b] - trim_offset;
            if (event_idx >= 0 && event_idx < (int64_t)n_events) {
                BAND(b, trim_offset) = (float)(lp_trim * (event_idx + 1));
                TRACE(b, trim_offset) = FROM_U;
            } else {
                BAND(b, trim_offset) = -INFINITY;
            }
        }
        const int kmer_min_offset = 0 - bl_k[b], kmer_max_offset = (int)n_kmers - bl_k[b];          /* :323-332 */
        const int event_min_offset = bl_e[b] - ((int)n_events - 1), event_max_offset = bl_e[b] - (-1);
        int min_offset = kmer_min_offset > event_min_offset ? kmer_min_offset : event_min_offset;
        min_offset = min_offset > 0 ? min_offset : 0;
        int max_offset = kmer_max_offset < event_max_offset ? kmer_max_offset : event_max_offset;
        max_offset = max_offset < bandwidth ? max_offset : bandwidth;
        for (int offset = min_offset; offset < max_offset; ++offset) {
            const int event_idx = bl_e[b] - offset, kmer_idx = bl_k[b] + offset;
            const int offset_up = bl_e[b - 1] - (event_idx - 1);
            const int offset_left = (kmer_idx - 1) - bl_k[b - 1];
            const int offset_diag = (kmer_idx - 1) - bl_k[b - 2];
            const float up = VALID(offset_up) ? BAND(b - 1, offset_up) : -INFINITY;
            const float left = VALID(offset_left) ? BAND(b - 1, offset_left) : -INFINITY;
            const float diag = VALID(offset_diag) ? BAND(b - 2, offset_diag) : -INFINITY;
            const float lp_emission = lp_match(scale, shift, models, event_mean[event_idx], ranks[kmer_idx]);
            const float score_d = diag + lp_step + lp_emission;                                      /* double sums, :371-373 */
            const float score_u = up + lp_stay + lp_emission;
            const float score_l = left + lp_skip;
            float max_score = score_d;
            uint8_t from = FROM_D;
            max_score = score_u > max_score ? score_u : max_score;
            from = max_score == score_u ? FROM_U : from;
            max_score = score_l > max_score ? score_l : max_score;
            from = max_score == score_l ? FROM_L : from;
            BAND(b, offset) = max_score;
            TRACE(b, offset) = from;
            ++fills;
        }
    }
    /* backtrack, :409-500 */
    double sum_emission = 0, n_aligned_events = 0;
    int out_index = 0;
    float max_score = -INFINITY;
    int curr_event_idx = 0, curr_kmer_idx = (int)n_kmers - 1;
    for (size_t event_idx = 0; event_idx < n_events; ++event_idx) {
        const int b = ((int)event_idx + 1) + (curr_kmer_idx + 1);
        const int offset = bl_e[b] - (int)event_idx;
        if (VALID(offset)) {
            const float s = BAND(b, offset) + (n_events - event_idx) * lp_trim;
            if (s > max_score) { max_score = s; curr_event_idx = (int)event_idx; }
        }
    }
    int curr_gap = 0, max_gap = 0;
    while (curr_kmer_idx >= 0 && curr_event_idx >= 0) {
        out[out_index].ref_pos = curr_kmer_idx;
        out[out_index].read_pos = curr_event_idx;
        ++out_index;
        const float lp = lp_match(scale, shift, models, event_mean[curr_event_idx], kmer_rank(sequence + curr_kmer_idx));
        sum_emission += lp;
        n_aligned_events += 1;
        const int b = (curr_event_idx + 1) + (curr_kmer_idx + 1);
        const int offset = bl_e[b] - curr_event_idx;
        const uint8_t from = TRACE(b, offset);
        if (from == FROM_D) { curr_kmer_idx -= 1; curr_event_idx -= 1; curr_gap = 0; }
        else if (from == FROM_U) { curr_event_idx -= 1; curr_gap = 0; }
        else { curr_kmer_idx -= 1; curr_gap += 1; max_gap = curr_gap > max_gap ? curr_gap : max_gap; }
    }
    for (int c = 0, end = out_index - 1; c < out_index / 2; ++c, --end) {                           /* std::reverse */
        const gbx_abea_pair t = out[c]; out[c] = out[end]; out[end] = t;
    }
    const double avg_log_emission = sum_emission / n_aligned_events;                                /* QC, :530-541 */
    const int spanned = out[0].ref_pos == 0 && out[out_index - 1].ref_pos == (int)(n_kmers - 1);
    if (avg_log_emission < min_average_log_emission || !spanned || max_gap > max_gap_threshold) out_index = 0;
    if (cells) *cells = fills;
    free(ranks); free(bands); free(trace); free(bl_e); free(bl_k);
    return out_index;
#undef BAND
#undef TRACE
#undef VALID
}

/* reads in parallel (align_db's per-read loop, f5c.c:1350-1370): out pairs of read r start at out + 2 * event_off[r] */
void oracle_abea_align(int64_t n_reads, const int64_t *seq_off, const int32_t *seq_len, const char *seq_arena,
                       const int64_t *event_off, const float *event_mean, const gbx_abea_model *models,
                       const float *scale, const float *shift, gbx_abea_pair *out, int32_t *n_pairs,
                       int nthreads, int64_t *cells)
{
    int64_t total = 0;
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads > 0 ? nthreads : 1) reduction(+ : total)
    for (int64_t r = 0; r < n_reads; ++r) {
        int64_t c = 0;
        n_pairs[r] = oracle_abea_align_one(seq_arena + seq_off[r], seq_len[r], event_mean + event_off[r],
                                           event_off[r + 1] - event_off[r], models, scale[r], shift[r],
                                           out + 2 * event_off[r], &c);
        total += c;
    }
    if (cells) *cells = total;
}
