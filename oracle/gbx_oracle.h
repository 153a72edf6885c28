/* gbx_oracle.h — CPU restatements of the reference algorithms.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may
 * load liboracle.so, and only as the checker / the CPU number beside the GPU
 * one.  libgbx.so never links or calls it.
 *
 * Each function cites the reference lines (R/ = /root/reference/) it restates.
 */
#ifndef GBX_ORACLE_H
#define GBX_ORACLE_H
#include <stdint.h>
#include "../include/gbx.h"

#ifdef __cplusplus
extern "C" {
#endif

/* bsw: scalarBandedSWA, R/benchmarks/bsw/bandedSWA.cpp:128-249 */
int  oracle_bsw_extend_one(const gbx_bsw_params *p, int qlen, const uint8_t *query,
                           int tlen, const uint8_t *target, int h0,
                           gbx_bsw_result *out, int64_t *cells_computed);
/* loops pairs (scalarBandedSWAWrapper, bandedSWA.cpp:254-272); nthreads>1 = OpenMP */
void oracle_bsw_extend(const gbx_bsw_params *p, int64_t n,
                       const uint8_t *ref, const uint8_t *qer,
                       const int64_t *idr, const int64_t *idq,
                       const int32_t *len1, const int32_t *len2, const int32_t *h0,
                       gbx_bsw_result *out, int nthreads, int64_t *cells_computed);

/* chain: chain_dp, R/benchmarks/chain/src/host_kernel.cpp:30-94 */
void oracle_chain_one(int64_t n, const uint64_t *ax, const uint64_t *ay,
                      const gbx_chain_call *hdr, int32_t *score, int32_t *parent,
                      int32_t *target, int32_t *peak, int64_t *pairs_evaluated);
void oracle_chain(int64_t n_calls, const int64_t *anchor_off,
                  const uint64_t *ax, const uint64_t *ay, const gbx_chain_call *hdr,
                  int32_t *score, int32_t *parent, int32_t *target, int32_t *peak,
                  int nthreads, int64_t *pairs_evaluated);

/* phmm: GKL compute_full_prob<float|double> + computelikelihoodsboth policy (parity UNPINNED, see phmm_oracle.c) */
void   oracle_phmm_init(void);
double oracle_phmm_pair(int rslen, int haplen, const char *rs, const char *hap, const char *q,
                        const char *qi, const char *qd, const char *qc, int *used_double);
double oracle_phmm_pair_f64(int rslen, int haplen, const char *rs, const char *hap, const char *q,
                            const char *qi, const char *qd, const char *qc);
void   oracle_phmm_forward(int64_t n_pairs, const int32_t *pair_read, const int32_t *pair_hap,
                           const int64_t *read_off, const int32_t *read_len,
                           const char *rs, const char *q, const char *qi, const char *qd, const char *qc,
                           const int64_t *hap_off, const int32_t *hap_len, const char *hap,
                           double *out, int nthreads, int64_t *n_double);
const float  *oracle_phmm_mm_table_f(void);
const double *oracle_phmm_mm_table_d(void);

/* poa: spoa v3 NW convex/affine alignment + graph + heaviest-bundle consensus (parity UNPINNED, see poa_oracle.c) */
int  oracle_poa_window(const gbx_poa_params *P, int n_seqs, const char *const *seqs, const int32_t *lens,
                       char *cons, int64_t cons_cap, int64_t *stats /* nodes, edges, DP cells */);
void oracle_poa_consensus(const gbx_poa_params *P, int64_t n_windows, const int64_t *win_first_seq,
                          const int64_t *seq_off, const int32_t *seq_len, const char *arena,
                          char *cons, int32_t *cons_len, int64_t cons_stride, int nthreads, int64_t *cells);

/* abea: adaptive banded event alignment, R/benchmarks/abea/src/align.c:169-548 (parity UNPINNED by a compiled
 * reference: f5c.h needs htslib / HDF5 headers; the restated source is in the tree, see abea_oracle.c) */
int32_t oracle_abea_align_one(const char *sequence, int32_t sequence_len, const float *event_mean, int64_t n_events,
                              const gbx_abea_model *models, float scale, float shift, gbx_abea_pair *out, int64_t *cells);
void oracle_abea_align(int64_t n_reads, const int64_t *seq_off, const int32_t *seq_len, const char *seq_arena,
                       const int64_t *event_off, const float *event_mean, const gbx_abea_model *models,
                       const float *scale, const float *shift, gbx_abea_pair *out, int32_t *n_pairs,
                       int nthreads, int64_t *cells);

/* fmi: bwa-mem2 FMI_search SMEM seeding as R/benchmarks/fmi/fmi.cpp:180-286 drives it (parity UNPINNED: tools/bwa-mem2 is
 * an empty submodule; see fmi_oracle.c) */
int64_t oracle_fmi_read(const gbx_fmi_index *X, const gbx_fmi_params *P, const uint8_t *q, int32_t readlength, uint32_t rid,
                        gbx_fmi_smem *out, gbx_fmi_smem *prev, int64_t *n_ext, int32_t *round_counts);
int64_t oracle_fmi_smem(const gbx_fmi_index *X, const gbx_fmi_params *P, int64_t n_reads, const uint8_t *enc,
                        const int64_t *read_off, const int32_t *read_len, gbx_fmi_smem *out, int64_t out_cap,
                        int64_t *smem_off, int nthreads, int64_t *n_ext_total, int64_t *round_totals);
void oracle_fmi_build_index(const uint8_t *text, int64_t n, const int64_t *sa, gbx_fmi_cp_occ *cp_occ, int64_t *count5,
                            int64_t *sentinel_index);

#ifdef __cplusplus
}
#endif
#endif
