/* gbx.h — C-ABI of libgbx.so: MI355X (gfx950) kernels for GenomicsBench's
 * dynamic-programming hot path (bsw, chain, phmm, poa).
 *
 * This is the drop-in boundary.  Every entry point is `extern "C"`, takes
 * plain pointers and sizes, returns an int status (0 = GBX_OK, <0 = error;
 * text via gbx_last_error()), never calls exit() and never falls back to a
 * CPU implementation: with no usable HIP device every compute entry point
 * returns GBX_ERR_NO_DEVICE.
 *
 * Two flavours per kernel:
 *   *_host    host buffers in, host buffers out (H2D + kernels + D2H inside;
 *             this is what a reference driver binds to);
 *   *_device  device-resident buffers on a caller-provided hipStream_t (what
 *             bench.py times: inputs already in HBM).
 *
 * Reference interfaces each entry replaces are cited as
 * R/ = /root/reference/ file:line.
 */
#ifndef GBX_H
#define GBX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ status */
#define GBX_OK               0
#define GBX_ERR_ARG         -1   /* bad argument (null pointer, negative size, bad params) */
#define GBX_ERR_NO_DEVICE   -2   /* no HIP device / HIP runtime unusable                   */
#define GBX_ERR_HIP         -3   /* a HIP call failed (see gbx_last_error)                 */
#define GBX_ERR_NOMEM       -4   /* host or device allocation failed                       */
#define GBX_ERR_UNSUPPORTED -5   /* input exceeds a documented limit                       */

const char *gbx_version(void);
const char *gbx_last_error(void);       /* thread-local, never NULL */
int  gbx_device_count(void);            /* number of HIP devices, 0 if none; never errors  */
int  gbx_set_device(int dev);           /* selects the device later calls of this thread use */
int  gbx_device_name(char *buf, size_t cap);
/* Multi-GPU: how many devices of this node the *_host entries spread one call over.  The units of every
 * kernel here are independent, so a call is cut into n contiguous ranges of equal cost (nominal cells / anchors / bases,
 * the same rule as genomicsbench_amd/shard.py), range k runs on device k through a host lane of its own (its own streams,
 * pinned slabs and upload / download threads: the shard goes from the caller's memory straight to that device and its
 * results come back in place) and the call returns when all are done; results are identical to the one-device call.
 * This is the reference's one-engine-per-OpenMP-thread shape (bsw/main_banded.cpp:253-291, chain/src/host_kernel.cpp:98-107,
 * phmm/PairHMMUnitTest.cpp:224-247, poa/msa_spoa_omp.cpp:184-260) with devices in the place of threads.  A call too small
 * to cut (bsw: under 2 x 128 Ki pairs, ...) runs whole on one device and such calls take the devices in turn, so a driver
 * that hands over small slices from many threads still uses every GPU.
 *   n_gpus = 0: the default, i.e. the environment variable GBX_GPUS if set, else 1 (then the calling thread's current
 *   device is used, see gbx_set_device).  Process-wide; not meant to be changed while calls are in flight.
 *   GBX_DEVICE_MAP="0,0,1" (test aid) maps logical devices 0..n-1 onto physical ones: the multi-device path on one GPU.
 * The *_device entries are single-device by construction (the caller owns the buffers and the stream). */
int  gbx_host_set_devices(int n_gpus);
/* The cut itself, for callers that want to see or reuse it: cuts[0..n_parts] with cuts[k] = the first unit index at which the
 * cost of the units before it reaches k / n_parts of the total (negative costs count as 0). */
int  gbx_split_by_cost(int64_t n_units, const double *cost, int n_parts, int64_t *cuts);
int  gbx_host_devices(void);             /* devices the *_host entries currently use (>= 1; 0 without any HIP device) */
/* Optional: creates the calling thread's streams and the pinned staging buffers the *_host entries use for
 * large inputs (about 144 MB of pinned host memory) and runs one small transfer from each buffer on its stream,
 * so that the first large calls do not pay for them.  The
 * counterpart of constructing the reference's aligner object before its timed region
 * (bsw/main_banded.cpp:262-270).  The *_host entries do this themselves on demand. */
int  gbx_host_prepare(void);
/* Optional: puts one device block of `bytes` into the calling thread's lane cache, where the next *_host call
 * that needs a buffer of about that size finds it (e.g. the poa workspace, gbx_poa_workspace_bytes of the plan:
 * ~10 GB for the 'large' job, whose allocation can take seconds right after another process released its
 * memory).  Like gbx_host_prepare this keeps one-time setup out of a caller's timed region. */
int  gbx_host_reserve(size_t bytes);
/* Frees the device memory the *_host entries keep cached between calls (idle lanes only).  Optional. */
int  gbx_host_release(void);
/* Concurrent small calls of gbx_bsw_extend_host / _seqpairs, gbx_phmm_forward_host and gbx_poa_consensus_host are
 * combined: calls that are pending together (the reference drivers' OpenMP threads, one small call each:
 * bsw/main_banded.cpp:279-291, phmm/PairHMMUnitTest.cpp:224-247, poa/msa_spoa_omp.cpp:230-260) share one upload, one launch
 * set and one download, and every caller gets its own results and status.  Nothing to call: GBX_COMBINE=0 in the
 * environment switches it off.  This reads the counters of one kernel (1 bsw, 3 phmm, 4 poa): out[0] calls that were
 * eligible, out[1] device calls made for them, out[2] calls that shared a device call, out[3] most calls in one. */
int  gbx_host_combine_stats(int kernel, uint64_t out[4], int reset);

/* Timing helpers on a stream (HIP events), so that a Python/ctypes host can
 * time the exact stream the kernels are launched on without touching HIP. */
typedef struct gbx_timer gbx_timer;
int  gbx_timer_create(gbx_timer **t);
int  gbx_timer_start(gbx_timer *t, void *stream);
int  gbx_timer_stop(gbx_timer *t, void *stream);
int  gbx_timer_elapsed_ms(gbx_timer *t, float *ms);   /* synchronises on the stop event */
void gbx_timer_destroy(gbx_timer *t);

/* Per-kernel timing: between gbx_profile_begin() and gbx_profile_end() every
 * kernel launched by this thread through a *_device / *_host entry is bracketed
 * by HIP events on its own stream.  gbx_profile_end synchronises and returns,
 * per distinct kernel name (static strings, at most `cap`), the summed
 * duration in ms and the number of launches. */
int  gbx_profile_begin(void);
int  gbx_profile_end(int cap, const char **names, float *ms_sum, int *launches, int *n_stages);

/* Device memory helpers (used by the C++ drivers; bench.py uses torch tensors). */
int  gbx_malloc_device(void **p, size_t bytes);
int  gbx_free_device(void *p);
int  gbx_memcpy_h2d(void *dst, const void *src, size_t bytes, void *stream);
int  gbx_memcpy_d2h(void *dst, const void *src, size_t bytes, void *stream);
int  gbx_stream_synchronize(void *stream);

/* --------------------------------------------------------------------- bsw
 * Banded Smith-Waterman seed extension (bwa-mem2 BSW).
 * Replaces  BandedPairWiseSW::BandedPairWiseSW(...)        R/benchmarks/bsw/bandedSWA.cpp:51-100
 *           BandedPairWiseSW::getScores16(...)              R/benchmarks/bsw/bandedSWA.cpp:1124-1148
 *           (called at R/benchmarks/bsw/main_banded.cpp:286)
 * Per-pair semantics are those of BandedPairWiseSW::scalarBandedSWA
 *           R/benchmarks/bsw/bandedSWA.cpp:128-249 (bwa ksw_extend2), bit-exact.
 */
typedef struct gbx_bsw_params {
    int32_t o_del, e_del, o_ins, e_ins;  /* gap open / extend penalties (>=0; e_* >= 1)   */
    int32_t zdrop;                       /* 100 in the driver, main_banded.cpp:250        */
    int32_t end_bonus;                   /* 5 in the driver                                */
    int32_t w;                           /* band width, 100 in the driver                  */
    int8_t  mat[25];                     /* 5x5 score matrix, bwa_fill_scmat main_banded.cpp:73-81 */
    int8_t  pad_[3];
} gbx_bsw_params;

/* Fills p with the driver's defaults: a=1 b=4 ambig=-1 o=6 e=1 zdrop=100 end_bonus=5 w=100. */
void gbx_bsw_default_params(gbx_bsw_params *p);
/* Same as bwa_fill_scmat(a, b, ambig, mat) (main_banded.cpp:73-81); b is the positive penalty. */
void gbx_bsw_fill_scmat(int a, int b, int ambig, int8_t mat[25]);

/* Per-pair output record, 6 x int32, same field meaning as SeqPair's outputs
 * (R/benchmarks/bsw/bandedSWA.h:91-100). */
typedef struct gbx_bsw_result {
    int32_t score, tle, gtle, qle, gscore, max_off;
} gbx_bsw_result;

/* Mirror of the reference's 72-byte SeqPair (bandedSWA.h:91-100). */
typedef struct gbx_seqpair {
    int64_t idr, idq, id;
    int32_t len1, len2;
    int32_t h0;
    int32_t seqid, regid;
    int32_t score, tle, gtle, qle;
    int32_t gscore, max_off;
} gbx_seqpair;

/* Limits of the device path: query (len2) and target (len1) lengths. */
#define GBX_BSW_MAX_QLEN  8192
#define GBX_BSW_MAX_TLEN  65535

/* Host flat-array entry.  ref/qer are byte arenas of base codes 0..4; pair k's
 * target is ref[idr[k] .. idr[k]+len1[k]), its query qer[idq[k] .. +len2[k]).
 * Writes out[k] for k in [0,n).  Does not write pad records and does not
 * reorder caller memory (cf. bandedSWA.cpp:1172-1177, 1183-1210). */
int gbx_bsw_extend_host(const gbx_bsw_params *p, int64_t n,
                        const uint8_t *ref, int64_t ref_bytes,
                        const uint8_t *qer, int64_t qer_bytes,
                        const int64_t *idr, const int64_t *idq,
                        const int32_t *len1, const int32_t *len2,
                        const int32_t *h0, gbx_bsw_result *out);

/* Drop-in for getScores16 on the reference's own SeqPair array: reads
 * idr/idq/len1/len2/h0 from pairs[k] and writes score,tle,gtle,qle,gscore,
 * max_off in place.  ref_bytes/qer_bytes bound the two arenas. */
int gbx_bsw_extend_seqpairs(const gbx_bsw_params *p, gbx_seqpair *pairs, int64_t n,
                            const uint8_t *ref, int64_t ref_bytes,
                            const uint8_t *qer, int64_t qer_bytes);

/* Device-resident entry: all pointers are device pointers; the arenas must be
 * readable for 16 bytes past their last base (hipMalloc slack is enough);
 * work = scratch of gbx_bsw_workspace_bytes(n) bytes.  Asynchronous on
 * `stream` (a hipStream_t, may be NULL). */
size_t gbx_bsw_workspace_bytes(int64_t n);
int gbx_bsw_extend_device(const gbx_bsw_params *p, int64_t n,
                          const uint8_t *d_ref, const uint8_t *d_qer,
                          const int64_t *d_idr, const int64_t *d_idq,
                          const int32_t *d_len1, const int32_t *d_len2,
                          const int32_t *d_h0, gbx_bsw_result *d_out,
                          void *d_work, size_t work_bytes, void *stream);

/* ------------------------------------------------------------------- chain
 * minimap2 anchor chaining DP.
 * Replaces  host_chain_kernel(std::vector<call_t>&, std::vector<return_t>&, int)
 *           R/benchmarks/chain/src/host_kernel.cpp:96-108  (chain_dp :30-94)
 * Calls are concatenated: call c owns anchors [anchor_off[c], anchor_off[c+1]).
 */
typedef struct gbx_chain_call {
    float   avg_qspan;                         /* host_data.h:24-29 */
    int32_t max_dist_x, max_dist_y, bw, n_segs;
} gbx_chain_call;

#define GBX_CHAIN_MAX_ITER 5000   /* host_kernel.cpp:37 */
#define GBX_CHAIN_MAX_SKIP 25     /* host_kernel.cpp:38 */

/* target/peak may be NULL (the reference computes them but never prints them). */
int gbx_chain_host(int64_t n_calls, const int64_t *anchor_off,
                   const uint64_t *ax, const uint64_t *ay,
                   const gbx_chain_call *hdr,
                   int32_t *score, int32_t *parent,
                   int32_t *target, int32_t *peak);

size_t gbx_chain_workspace_bytes(int64_t n_calls, int64_t n_anchors);
int gbx_chain_device(int64_t n_calls, int64_t n_anchors, const int64_t *d_anchor_off,
                     const uint64_t *d_ax, const uint64_t *d_ay,
                     const gbx_chain_call *d_hdr,
                     int32_t *d_score, int32_t *d_parent,
                     int32_t *d_target, int32_t *d_peak,
                     void *d_work, size_t work_bytes, void *stream);

/* Work counter of the last gbx_chain_device call on this workspace: predecessor pairs (i,j) visited,
 * `continue`d ones included, those after the max_skip break excluded (the benchmark's "cell"). */
int gbx_chain_evaluated_pairs(const void *d_work, int64_t *pairs, void *stream);

/* How the last gbx_chain_device call on this workspace was scheduled: a call whose anchors are sorted by x falls apart
 * at every anchor that lies further than max_dist_x behind its predecessor (it looks back at nobody and nothing later
 * looks across it, host_kernel.cpp:56) into pieces that are chained independently (`jobs` >= calls; at most one cut per
 * 64 anchors); `longest_job` = anchors of the longest piece, what bounds the kernel's makespan. */
int gbx_chain_job_stats(const void *d_work, int64_t n_calls, int64_t n_anchors, int64_t *jobs, int64_t *longest_job, void *stream);

/* -------------------------------------------------------------------- phmm
 * GATK/GKL Pair-HMM forward log10-likelihoods.
 * Replaces  void initPairHMM()                                     R/benchmarks/phmm/PairHMMUnitTest.cpp:84,193
 *           void computelikelihoodsboth(testcase*, double*, int)   R/benchmarks/phmm/PairHMMUnitTest.cpp:86,245
 *           (C++-mangled symbols of libgkl_pairhmm_c.so; `testcase` = pairhmm_common.h:20-24)
 * Reads live in one arena per track (bases rs and the four quality tracks q,i,d,c share
 * read_off/read_len; qualities already Phred-33 as the driver normalises them,
 * PairHMMUnitTest.cpp:89-93,110-113), haplotypes in another; pair p = (read pair_read[p],
 * haplotype pair_hap[p]).  out[p] = log10 likelihood (fp32 pass, fp64 redo below 1e-28f,
 * pairhmm_common.h:16).  Tolerance vs the CPU path: 1e-5 relative.
 */
#define GBX_PHMM_MAX_HAPLEN 32768     /* MAX_HAP_LENGTH, PairHMMUnitTest.h:36 */

int gbx_phmm_init(void);              /* builds + uploads the probability tables (initPairHMM) */

int gbx_phmm_forward_host(int64_t n_pairs, const int32_t *pair_read, const int32_t *pair_hap,
                          int64_t n_reads, const int64_t *read_off, const int32_t *read_len, int64_t read_bytes,
                          const uint8_t *rs, const uint8_t *q, const uint8_t *i, const uint8_t *d, const uint8_t *c,
                          int64_t n_haps, const int64_t *hap_off, const int32_t *hap_len, int64_t hap_bytes,
                          const uint8_t *hap, double *out);

/* The device entry groups the pairs by read (workspace arrays indexed by read id, hence n_reads) and
 * materialises one haplotype byte stream per read: workspace ~ n_pairs * (max_hap_len + 1) bytes.
 * The haplotype arena must be readable for 16 bytes past its last base (hipMalloc slack is enough). */
size_t gbx_phmm_workspace_bytes(int64_t n_pairs, int64_t n_reads, int32_t max_hap_len);
int gbx_phmm_forward_device(int64_t n_pairs, const int32_t *d_pair_read, const int32_t *d_pair_hap,
                            int64_t n_reads, const int64_t *d_read_off, const int32_t *d_read_len,
                            const uint8_t *d_rs, const uint8_t *d_q, const uint8_t *d_i, const uint8_t *d_d,
                            const uint8_t *d_c,
                            const int64_t *d_hap_off, const int32_t *d_hap_len, const uint8_t *d_hap,
                            int32_t max_hap_len, double *d_out,
                            void *d_work, size_t work_bytes, void *stream);

/* --------------------------------------------------------------------- poa
 * Partial-order-alignment consensus (spoa).
 * Replaces, at whole-window granularity, the driver's per-window loop
 *   spoa::createAlignmentEngine(kNW, m, n, g, e, q, c)   R/benchmarks/poa/msa_spoa_omp.cpp:189-190
 *   spoa::createGraph()                                   :237
 *   AlignmentEngine::align(seq, graph)                    :242
 *   Graph::add_alignment(alignment, seq)                  :247
 *   Graph::generate_consensus()                           :252
 * Window w owns sequences [win_first_seq[w], win_first_seq[w+1]); sequence s is
 * arena[seq_off[s] .. seq_off[s]+seq_len[s]).
 */
typedef struct gbx_poa_params {
    int8_t m, n;      /* match score (2) and mismatch score (-4), msa_spoa_omp.cpp:157-158 */
    int8_t g, e;      /* first gap piece: open(-6 = o1+e1) / extend(-2), :184 */
    int8_t q, c;      /* second gap piece: open(-25 = o2+e2) / extend(-1) */
    int8_t pad_[2];
} gbx_poa_params;

void gbx_poa_default_params(gbx_poa_params *p);

/* Capacities of the device path for a set of windows (computed from host metadata). */
typedef struct gbx_poa_plan {
    int32_t max_seq_len;          /* longest sequence                                    */
    int32_t max_seqs_per_window;  /* bounds the fan-in / fan-out of a graph node          */
    int32_t node_cap;             /* graph nodes per window the workspace can hold       */
    int32_t n_slots;              /* windows processed concurrently (one wavefront each) */
    /* Windows that hold a sequence of more than 512 bases run on a second launch with slots of their own (five int16
     * planes per DP matrix instead of two): a handful of long sequences must not size every slot of the job. */
    int32_t n_long_windows;       /* windows with a sequence longer than 512             */
    int32_t long_slots;           /* slots of the second launch (0 when there is none)   */
    int64_t n_windows;            /* windows the plan was made for                       */
} gbx_poa_plan;

#define GBX_POA_MAX_SEQS_PER_WINDOW 255
#define GBX_POA_MAX_LETTERS_PER_COLUMN 8

/* per-window status bits written by the device path (0 = ok) */
#define GBX_POA_ST_NODES   1   /* node_cap exceeded                      */
#define GBX_POA_ST_DEGREE  2   /* fan-in/out above max_seqs_per_window   */
#define GBX_POA_ST_LETTERS 4   /* more than 8 distinct letters aligned   */
#define GBX_POA_ST_STACK   8
#define GBX_POA_ST_CONS    16  /* consensus longer than cons_stride      */

int gbx_poa_plan_host(int64_t n_windows, const int64_t *win_first_seq, const int32_t *seq_len, gbx_poa_plan *plan);
size_t gbx_poa_workspace_bytes(const gbx_poa_plan *plan);

/* DP cells of the last gbx_poa_consensus_device call on this workspace: sum over alignments of
 * graph nodes x sequence length. */
int gbx_poa_cells(const gbx_poa_plan *plan, const void *d_work, int64_t *cells, void *stream);

/* cons: n_windows rows of cons_stride bytes (not NUL-terminated), cons_len[w] = consensus length.
 * Returns GBX_ERR_UNSUPPORTED (and names the first window) if any window overflowed a capacity. */
int gbx_poa_consensus_host(const gbx_poa_params *p, int64_t n_windows, const int64_t *win_first_seq,
                           int64_t n_seqs, const int64_t *seq_off, const int32_t *seq_len,
                           const char *arena, int64_t arena_bytes,
                           char *cons, int32_t *cons_len, int64_t cons_stride);

/* Device-resident entry; d_status[w] receives the GBX_POA_ST_* bits. */
int gbx_poa_consensus_device(const gbx_poa_params *p, const gbx_poa_plan *plan, int64_t n_windows,
                             const int64_t *d_win_first_seq, const int64_t *d_seq_off, const int32_t *d_seq_len,
                             const char *d_arena, char *d_cons, int32_t *d_cons_len, int32_t *d_status,
                             int64_t cons_stride, void *d_work, size_t work_bytes, void *stream);

/* -------------------------------------------------------------------- abea
 * Adaptive banded event alignment of nanopore events to the k-mers of a read's basecalled sequence (f5c /
 * nanopolish; SURVEY §8f rank 4: the suite's other banded DP).
 * Replaces  int32_t align(AlignedPair *out, char *sequence, int32_t sequence_len, event_table events,
 *                         model_t *models, scalings_t scaling, float sample_rate)
 *           R/benchmarks/abea/src/align.c:169-548, called per read by align_single, f5c.c:1344-1349
 *           (the suite's CUDA path for the same step: align.cu:140-560 - not a template for this code).
 * Semantics are those of the CPU function, bit for bit: float band scores and emissions, double transition
 * penalties (every candidate is a double sum rounded to float), Suzuki's adaptive band placement, the traceback,
 * the double emission sum and the three QC rules that empty an alignment.  `sample_rate` is unused by align()
 * (:108-126) and has no counterpart here.  Of an event only its mean is read (:125).
 */
#define GBX_ABEA_BANDWIDTH 100     /* ALN_BANDWIDTH, f5c.h:28 */
#define GBX_ABEA_KMER      6       /* KMER_SIZE, f5c.h:24 */
#define GBX_ABEA_NMODEL    4096    /* 4^KMER_SIZE model states */

typedef struct gbx_abea_model {   /* model_t with CACHED_LOG, f5c.h:122-136 */
    float level_mean, level_stdv, level_log_stdv;     /* level_log_stdv = log(level_stdv), model.c:53 */
} gbx_abea_model;
typedef struct gbx_abea_event {   /* event_t, f5c.h:104-111 (24 bytes) */
    uint64_t start;
    float length, mean, stdv;
} gbx_abea_event;
typedef struct gbx_abea_pair {    /* AlignedPair, f5c.h:163-166 */
    int32_t ref_pos, read_pos;    /* k-mer index, event index */
} gbx_abea_pair;

/* Read r: bases seq_arena[seq_off[r] .. +seq_len[r]) (A/C/G/T; anything else ranks as A, align.c:10-24), events
 * [event_off[r], event_off[r+1]) of the concatenated event array, scalings scale[r], shift[r] (scalings_t, f5c.h:139-155).
 * Output: pairs of read r at out + 2*event_off[r] (the reference sizes the array 2 x n_events, f5c.c), n_pairs[r] of
 * them in ascending order, 0 when a QC rule failed (align.c:530-541); the slots of a read behind its n_pairs are
 * unspecified.  seq_len >= KMER and >= 1 event per read. */
int gbx_abea_align_host(int64_t n_reads, const int64_t *seq_off, const int32_t *seq_len, const char *seq_arena,
                        int64_t seq_bytes, const int64_t *event_off, const gbx_abea_event *events,
                        const gbx_abea_model *models, const float *scale, const float *shift,
                        gbx_abea_pair *out, int32_t *n_pairs);

/* Device path.  gbx_abea_plan_host computes from host metadata the per-read offsets into the band workspace
 * (band_off[n_reads+1], in bands), the processing order (longest first) and the two read-dependent transition
 * penalties, which are double logarithms (align.c:195-204) and are taken with the host C library so that they are the
 * reference's bits; the device entry takes the compact float array of event means.  n_kmers_total = sum of
 * seq_len - KMER + 1, n_bands_total = band_off[n_reads].
 * Indexing of the device entry: ABSOLUTE - read r's means are d_event_mean[d_event_off[r] ..], its pairs are written at
 * d_out + 2*d_event_off[r]; d_event_off[0] need not be 0 (both arrays must then reach up to d_event_off[n_reads]). */
int gbx_abea_plan_host(int64_t n_reads, const int32_t *seq_len, const int64_t *event_off,
                       int64_t *band_off, int32_t *order, double *lp /* [n_reads][2]: lp_stay, lp_step, align.c:195-204 */);
size_t gbx_abea_workspace_bytes(int64_t n_reads, int64_t n_kmers_total, int64_t n_bands_total);
int gbx_abea_align_device(int64_t n_reads, const int64_t *d_seq_off, const int32_t *d_seq_len, const char *d_seq_arena,
                          const int64_t *d_event_off, const float *d_event_mean, const gbx_abea_model *d_models,
                          const float *d_scale, const float *d_shift, const int64_t *d_band_off, const int32_t *d_order,
                          const double *d_lp, int64_t n_kmers_total, int64_t n_bands_total,
                          gbx_abea_pair *d_out, int32_t *d_n_pairs, void *d_work, size_t work_bytes, void *stream);
/* DP cells filled by the last gbx_abea_align_device call on this workspace (the reference's `fills`, align.c:280,401). */
int gbx_abea_cells(const void *d_work, int64_t *cells, void *stream);

/* --------------------------------------------------------------------- fmi
 * SMEM seeding on the FM-index of reference + reverse complement (second half): the three
 * seeding rounds bwa-mem2 runs per batch of reads and the driver times,
 *   R/benchmarks/fmi/fmi.cpp:218-228  FMI_search::getSMEMsAllPosOneThread   SMEMs from every start position
 *   R/benchmarks/fmi/fmi.cpp:230-254  re-seeding: getSMEMsOnePosOneThread from the middle of every SMEM of at least
 *                                     split_len bases with at most split_width hits, min_intv = hits + 1
 *   R/benchmarks/fmi/fmi.cpp:255-266  FMI_search::bwtSeedStrategyAllPosOneThread (max_intv, minSeedLen + 1)
 *   R/benchmarks/fmi/fmi.cpp:270-278  rid += batch offset, FMI_search::sortSMEMs
 * FMI_search lives in tools/bwa-mem2 (an empty submodule here): the arithmetic follows bwa-mem2's published
 * src/FMI_search.cpp (backwardExt over the CP_OCC checkpoints of 64 BWT symbols) - parity UNPINNED by a compiled
 * reference, see oracle/fmi_oracle.c.
 * The three rounds and the sort only ever combine SMEMs of one read, and batches are contiguous rid ranges sorted by
 * rid first, so the job's result is independent of the batch size: for every read, in rid order, its SMEMs of all
 * three rounds sorted by (m ascending, n descending).  Records equal in (rid, m, n) are equal in every field.
 */
typedef struct gbx_fmi_cp_occ {      /* bwa-mem2 CP_OCC (FMI_search.h): one checkpoint per 64 BWT symbols, 64 bytes */
    int64_t  cp_count[4];            /* occurrences of A, C, G, T in bwt[0, 64 i) */
    uint64_t one_hot_bwt_str[4];     /* bit 63 - j set iff bwt[64 i + j] is that base */
} gbx_fmi_cp_occ;
typedef struct gbx_fmi_index {       /* the fields of FMI_search the search reads (load_index) */
    int64_t ref_seq_len;             /* reference_seq_len: 2 x genome length + 1 (the sentinel) */
    int64_t count[5];                /* first SA row of every base, sentinel row included: count[0] = 1, count[4] = ref_seq_len */
    int64_t sentinel_index;          /* SA row whose BWT symbol is the sentinel */
    const gbx_fmi_cp_occ *cp_occ;    /* (ref_seq_len >> 6) + 1 checkpoints; host pointer for *_host, device pointer for *_device */
} gbx_fmi_index;
typedef struct gbx_fmi_smem {        /* bwa-mem2 SMEM (FMI_search.h), 40 bytes */
    uint32_t rid;                    /* read */
    uint32_t m, n;                   /* query interval [m, n], both inclusive (the driver prints [m, n + 1)) */
    uint32_t pad_;
    int64_t  k, l, s;                /* SA interval of the match, of its reverse complement, and their size */
} gbx_fmi_smem;
typedef struct gbx_fmi_params {      /* fmi.cpp:135-140,178 */
    int32_t min_seed_len;            /* argv[4]; the benchmark scripts pass 19 */
    int32_t split_width;             /* 10 */
    int32_t split_len;               /* (int)(min_seed_len * 1.5 + .499) */
    int32_t max_mem_intv;            /* 20 */
} gbx_fmi_params;
void gbx_fmi_default_params(gbx_fmi_params *p, int32_t min_seed_len);

/* Host-buffer entry: reads as base codes 0..3 (4 = ambiguous, fmi.cpp:113-124; the CONTRACT is codes 0..4 - larger values are not
 * checked and their treatment is unspecified), read r = enc[read_off[r] ..+ read_len[r]).
 * out receives the SMEMs (out_cap records; GBX_ERR_ARG with the needed count in gbx_last_error() when it is too small),
 * smem_off[n_reads + 1] (nullable) where each read's run starts, *n_out the total. */
int gbx_fmi_smem_host(const gbx_fmi_index *idx, const gbx_fmi_params *p, int64_t n_reads, const uint8_t *enc, int64_t enc_bytes,
                      const int64_t *read_off, const int32_t *read_len, gbx_fmi_smem *out, int64_t out_cap,
                      int64_t *smem_off, int64_t *n_out);

/* Device path.  The checkpoints are re-laid for the device once per index (gbx_fmi_index_bytes / gbx_fmi_index_build:
 * the count and the one-hot word of a base side by side, so that the four lanes of a read fetch a checkpoint as one
 * 64-byte line).  d_smem_off[n_reads + 1] and d_n_out (one int64) are written on the device; *d_n_out greater than
 * out_cap means the output did not fit (nothing beyond out_cap is written). */
size_t gbx_fmi_index_bytes(int64_t ref_seq_len);
int gbx_fmi_index_build(const gbx_fmi_index *idx_with_device_cp_occ, void *d_index, size_t index_bytes, void *stream);
size_t gbx_fmi_workspace_bytes(int64_t n_reads, int32_t max_read_len, int32_t min_seed_len);
int gbx_fmi_smem_device(const gbx_fmi_index *idx, const void *d_index, const gbx_fmi_params *p, int64_t n_reads,
                        int32_t max_read_len, const uint8_t *d_enc, const int64_t *d_read_off, const int32_t *d_read_len,
                        gbx_fmi_smem *d_out, int64_t out_cap, int64_t *d_smem_off, int64_t *d_n_out,
                        void *d_work, size_t work_bytes, void *stream);
/* A read's SMEMs wait in a slot of max(48, 4 * max_read_len / min_seed_len + 16) records for the pack pass.  *worst = 0, or
 * a lower bound of the largest count a read of the last gbx_fmi_smem_device call on this workspace asked for when that was
 * more (its surplus records, and what the re-seeding round would have made of them, are missing from the output then:
 * very repetitive text, long reads with short seeds).  The host entry runs such a job again with larger slots until
 * every read fits.  GBX_ERR_ARG when a read was longer than the max_read_len the call was given (such a read gets no SMEMs). */
int gbx_fmi_overflow(const void *d_work, int64_t *worst, void *stream);
/* backwardExt calls (checkpoint look-ups: two 64-byte lines each) of the last gbx_fmi_smem_device call on this workspace. */
int gbx_fmi_extensions(const void *d_work, int64_t *ext, void *stream);
/* gbx_fmi_smem_host keeps the device copy of an index between calls (a caller hands over the same tables for every batch
 * of reads, fmi.cpp:218), found again by content - the index scalars and a fingerprint of 256 checkpoints - not by address;
 * at most four idle copies per device; this frees the ones no call is using. */
int gbx_fmi_host_release(void);

#ifdef __cplusplus
}
#endif
#endif /* GBX_H */
