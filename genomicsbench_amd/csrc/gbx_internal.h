// gbx_internal.h — shared declarations between the C-ABI translation unit and
// the per-kernel HIP files of libgbx.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <mutex>
#include "../../include/gbx.h"

namespace gbx {

// thread-local last-error text (gbx_core.hip)
void set_error(const char *fmt, ...);
int  hip_fail(hipError_t e, const char *what);

#define GBX_HIP(call)                                            \
    do {                                                         \
        hipError_t e_ = (call);                                  \
        if (e_ != hipSuccess) return gbx::hip_fail(e_, #call);   \
    } while (0)

// Optional per-kernel timing with HIP events on the launch stream (gbx_profile_*).
// A Stage brackets one kernel launch; it is a no-op unless profiling is on.
// roctx ranges (the reference's analogue: ITT pause/resume and task markers around its timed regions,
// bsw/main_banded.cpp:203-205,274-276,293-295).  GBX_ROCTX=1 loads librocprofiler-sdk-roctx.so (libroctx64.so as
// a fallback) on first use and brackets host-entry calls, H2D / D2H transfers and kernel launches with
// roctxRangePush/Pop on the calling thread, so that `rocprofv3 --marker-trace` shows them beside the kernels;
// unset, the constructor is one predictable branch.
struct RoctxRange {
    explicit RoctxRange(const char *name);
    ~RoctxRange();
    bool on_;
};

struct Stage {
    Stage(const char *name, hipStream_t s);
    ~Stage();
    int slot_;
    hipStream_t s_;
    RoctxRange range_;
};

// ---- loop guards (-DGBX_LOOP_GUARD: a diagnostic build of libgbx.so, scripts/build_guard.sh; round 5) --------------------------------
// A device loop whose trip count depends on data (traceback extensions, work-list walks, spin-waits between wavefronts) hangs the
// whole process if the data is ever not what the algorithm guarantees.  In the guard build every such loop counts down from a bound
// derived from its input sizes; a loop that hits it records (kernel, loop, unit) in a device word and leaves, and the launch function
// returns GBX_ERR_HIP with the record in the error text instead of the process dying in a "GPU Hang".  The product build carries only
// the bounds that are free (a comparison the loop makes anyway).
//   GBX_GUARD(var, bound)                   declares the counter (nothing in the product build)
//   GBX_GUARD_TRIP(var, kernel, loop, unit) true when the bound is exhausted (constant false in the product build)
//   GBX_GUARD_CHECK(what)                   in a launch function, after its launches: hipDeviceSynchronize() (the whole device, other
//                                           callers' streams included: the guard build is diagnostic only and its timings mean nothing), then a record becomes an error
enum { GBX_GK_BSW = 1, GBX_GK_CHAIN = 2, GBX_GK_PHMM = 3, GBX_GK_POA = 4, GBX_GK_ABEA = 5, GBX_GK_FMI = 6 };
#ifdef GBX_LOOP_GUARD
namespace { __device__ unsigned long long gbx_guard_word; }      // one per translation unit
__device__ inline bool gbx_guard_report(int kernel, int loop, long long unit)
{
    atomicCAS(&gbx_guard_word, 0ull, 1ull << 63 | (unsigned long long)kernel << 56 | (unsigned long long)loop << 48 | ((unsigned long long)unit & 0xffffffffffffull));
    return true;
}
#define GBX_GUARD(var, bound) long long var = (long long)(bound)
#define GBX_GUARD_TRIP(var, kernel, loop, unit) (--(var) < 0 && gbx::gbx_guard_report((kernel), (loop), (long long)(unit)))
static inline int gbx_guard_check(const char *what)
{
    unsigned long long v = 0;
    hipError_t e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpyFromSymbol(&v, HIP_SYMBOL(gbx_guard_word), sizeof(v));
    if (e != hipSuccess) return hip_fail(e, what);
    if (!v) return GBX_OK;
    const unsigned long long zero = 0;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(gbx_guard_word), &zero, sizeof(zero));
    set_error("%s: loop guard hit: kernel %d, loop %d, unit %lld (a data-dependent device loop ran past the bound its inputs allow)", what,
              (int)(v >> 56 & 0x7f), (int)(v >> 48 & 0xff), (long long)(v & 0xffffffffffffull));
    return GBX_ERR_HIP;
}
#define GBX_GUARD_CHECK(what) do { const int grc_ = gbx::gbx_guard_check(what); if (grc_) return grc_; } while (0)
#else
#define GBX_GUARD(var, bound)
#define GBX_GUARD_TRIP(var, kernel, loop, unit) false
#define GBX_GUARD_CHECK(what) do { } while (0)
#endif

// Side streams for independent kernels of one call (the per-class kernels have long single-wave tails that
// overlap well).  fork(): the side streams wait for everything queued on `main`; join(): `main` waits for
// them.  One set per device, created on first use and shared by every caller on that device: the launch
// functions hold `mu` from fork() to join() (host enqueue time only), because the events are shared and two
// host threads recording them in between each other would wait on the wrong record.
struct SideStreams {
    static constexpr int N = 3;
    hipStream_t side[N];
    hipStream_t pre[2];                          // urgent streams for the passes that prepare a launch (bsw's pipelined chunks)
    hipEvent_t ev_fork, ev_join[N], ev_aux;      // ev_aux: a point on one side stream the others wait for (bsw: the lane sort)
    hipEvent_t ev_pre;                           // the other preparing pass (bsw: classify)
    std::mutex mu;
    int fork(hipStream_t main);
    int join(hipStream_t main);
};
int side_streams(SideStreams **out);

// ---- bsw (bsw_kernels.hip)
// A chunk of a pipelined host call (join_events set): what has to happen between its uploads and its kernels.  With it the
// launch puts these passes (unpacking, classify, the lane sort) on the urgent streams, waiting for `uploaded` only, and the
// chunk's kernels wait for them - not, as everything on the caller's stream would, for the previous chunk's kernels.
struct BswChunkPrep {
    hipEvent_t uploaded;                          // recorded on the copy stream behind the chunk's uploads
    const uint8_t *ref_packed, *qer_packed;       // 4-bit arenas to unpack (nullptr: the bytes were uploaded as they are)
    uint8_t *ref_bytes, *qer_bytes;
    int64_t lo_r, hi_r, lo_q, hi_q;
    int64_t rows_pairs;                           // pairs of the chunk that bsw_lane_takes() turns down, or an upper bound; -1: not counted
    // 0: one call does everything.  1: the preparing passes only, behind `uploaded` = the chunk's index arrays (returns 1 and
    // queues nothing if this chunk's launch cannot be split); 2: the kernels, behind `uploaded` = its bases.  ev_pre / ev_aux:
    // the caller's own events that tie the two calls together (the side streams' are shared by all callers of a device).
    int phase;
    hipEvent_t ev_pre, ev_aux;
    // How far the caller's byte arenas have been expanded so far (the call's watermarks, nullptr: [lo, hi) as given).  A chunk
    // the lane kernels take whole reads the packed images and expands nothing, so the next chunk that does need the bytes
    // expands from the watermark, not from its own lo: its pairs may lie in what an earlier, packed chunk brought up.
    int64_t *unp_r = nullptr, *unp_q = nullptr;
    // Pairs of the chunk per lane launch (bsw_lane_class: format x query-length range), or upper bounds; class_known = 0: not
    // counted.  With GBX_BSW_SKIP_EMPTY=1 a launch whose class is empty is left out (it has to be given its LDS before its
    // wavefronts can see that their list is empty: 0.1-0.85 ms on its stream in profiles/r05al_host_timeline.txt) - measured no
    // faster on 'large' (six of ten classes empty; profiles/r06h_bsw_skip_empty_ab.txt), so not the default.
    int class_known = 0;
    int64_t class_pairs[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
};
// Which pairs a launch of n pairs puts on the lane kernels (bsw_kernels.hip: lane_ok), for a host pass that counts the
// others: a launch that knows there are none leaves out the row-kernel classes, twenty-one near-empty launches.
struct BswLaneRule { int on, max_mat, qmax, limit; int compact_limit = 0; int range_hi[2][5] = {{0, 0, 0, 0, 0}, {0, 0, 0, 0, 0}}; };
int bsw_lane_rule(const gbx_bsw_params *p, int64_t n, BswLaneRule *r);
static inline bool bsw_lane_takes(const BswLaneRule &r, int qlen, int tlen, int h0)
{
    return r.on && qlen >= 1 && qlen <= r.qmax && tlen >= 1 && h0 >= 0 && h0 + qlen * r.max_mat < r.limit;
}
// the lane launch that takes a pair bsw_lane_takes() accepts: format (0 compact cells, 1 wide) x 5 + query-length range
static inline int bsw_lane_class(const BswLaneRule &r, int qlen, int h0)
{
    const int fmt = h0 + qlen * r.max_mat < r.compact_limit ? 0 : 1;
    int k = 0;
    while (k < 4 && qlen > r.range_hi[fmt][k]) ++k;
    return fmt * 5 + k;
}
size_t bsw_workspace_bytes(int64_t n);
int bsw_launch(const gbx_bsw_params *p, int64_t n,
               const uint8_t *d_ref, const uint8_t *d_qer,
               const int64_t *d_idr, const int64_t *d_idq,
               const int32_t *d_len1, const int32_t *d_len2, const int32_t *d_h0,
               gbx_bsw_result *d_out, void *d_work, size_t work_bytes, hipStream_t s,
               hipEvent_t *join_events = nullptr, const struct BswChunkPrep *prep = nullptr);

int bsw_unpack4(const uint8_t *d_packed, uint8_t *d_out, int64_t lo, int64_t hi, hipStream_t s);
int bsw_launch_direct(const gbx_bsw_params *p, int64_t n, int max_qlen,
                      const uint8_t *d_ref, const uint8_t *d_qer, const int64_t *d_idr, const int64_t *d_idq,
                      const int32_t *d_len1, const int32_t *d_len2, const int32_t *d_h0, gbx_bsw_result *d_out, hipStream_t s);

// ---- chain (chain_kernels.hip)
size_t chain_workspace_bytes(int64_t n_calls, int64_t n_anchors);
int chain_launch(int64_t n_calls, int64_t n_anchors, const int64_t *d_off,
                 const uint64_t *d_ax, const uint64_t *d_ay, const gbx_chain_call *d_hdr,
                 int32_t *d_score, int32_t *d_parent, int32_t *d_target, int32_t *d_peak,
                 void *d_work, size_t work_bytes, hipStream_t s);

int chain_read_evaluated(const void *d_work, int64_t *pairs, hipStream_t s);
int chain_read_job_stats(const void *d_work, int64_t n_calls, int64_t n_anchors, int64_t *jobs, int64_t *longest, hipStream_t s);
int poa_read_cells(const void *d_work, size_t slots_bytes, int64_t *cells, hipStream_t s);

// ---- fmi (fmi_kernels.hip)
size_t fmi_index_bytes(int64_t ref_seq_len);
int fmi_index_build(const gbx_fmi_index *idx, void *d_index, size_t index_bytes, hipStream_t s);
size_t fmi_workspace_bytes(int64_t n_reads, int32_t max_len, int32_t min_seed_len, int raw_cap = 0);
int fmi_launch(const gbx_fmi_index *idx, const void *d_index, const gbx_fmi_params *p, int64_t n_reads, int32_t max_len,
               const uint8_t *d_enc, const int64_t *d_read_off, const int32_t *d_read_len, gbx_fmi_smem *d_out, int64_t out_cap,
               int64_t *d_smem_off, int64_t *d_n_out, void *d_work, size_t work_bytes, hipStream_t s, int raw_cap = 0);
int fmi_read_extensions(const void *d_work, int64_t *ext, hipStream_t s);
int fmi_read_overflow(const void *d_work, int64_t *worst, hipStream_t s);

// ---- phmm (phmm_kernels.hip)
size_t phmm_workspace_bytes(int64_t n_pairs, int64_t n_reads, int max_hap_len, int64_t stream_syms = -1);
int phmm_init_tables();
const float *phmm_host_mm_table_f(int *n);
int phmm_launch(int64_t n_pairs, const int32_t *pair_read, const int32_t *pair_hap,
                int64_t n_reads, const int64_t *read_off, const int32_t *read_len,
                const uint8_t *rs, const uint8_t *q, const uint8_t *qi, const uint8_t *qd, const uint8_t *qc,
                const int64_t *hap_off, const int32_t *hap_len, const uint8_t *hap, int max_hap_len,
                double *out, void *d_work, size_t work_bytes, hipStream_t s, int64_t stream_syms = -1);

// ---- abea (abea_kernels.hip)
size_t abea_workspace_bytes(int64_t n_reads, int64_t n_kmers_total, int64_t n_bands_total);
int abea_read_cells(const void *d_work, int64_t *cells, hipStream_t s);
// after abea_launch: the aligned pairs of all reads packed back to back (read r's n_pairs[r] pairs at d_prefix[r]) into the
// workspace's trace area, which is free by then; returns where
int abea_pack_pairs(int64_t n_reads, const int64_t *d_event_off, const gbx_abea_pair *d_out, const int32_t *d_n_pairs,
                    const int64_t *d_prefix, void *d_work, int64_t n_kmers_total, gbx_abea_pair **d_packed, hipStream_t s);
int abea_launch(int64_t n_reads, const int64_t *d_seq_off, const int32_t *d_seq_len, const char *d_seq,
                const int64_t *d_event_off, const float *d_event_mean, const gbx_abea_model *d_models,
                const float *d_scale, const float *d_shift, const int64_t *d_band_off, const int32_t *d_order,
                const double *d_lp, int64_t n_kmers_total, int64_t n_bands_total,
                gbx_abea_pair *d_out, int32_t *d_n_pairs, void *d_work, size_t work_bytes, hipStream_t s);

// ---- poa (poa_kernels.hip)
constexpr int POA_PIPE_MAXLEN = 512;     // longest sequence of the pipelined DP (two-plane slots)
size_t poa_slot_bytes(int ncap, int deg, int lmax, bool long_slot);
size_t poa_workspace_bytes(const gbx_poa_plan *plan);
int poa_waves_per_cu(int ncap);
bool poa_lockstep_wanted(int64_t n_main, int64_t resident);      // the lock-step form (a slot per window) pays for this job
bool poa_scores_fit_int16(const gbx_poa_params *p, int64_t ncap, int lmax);
// int32 cells (poa_wide_kernel): windows whose scores may leave the int16 range, or whose graph outgrew what int16 admits
size_t poa_wide_slot_bytes(int ncap, int deg, int lmax);
size_t poa_wide_workspace_bytes(int ncap, int deg, int lmax, int n_slots);
int poa_launch_wide(const gbx_poa_params *p, int64_t n_windows, const int64_t *d_win_first_seq, const int64_t *d_seq_off, const int32_t *d_seq_len,
                    const uint8_t *d_arena, uint8_t *d_cons, int32_t *d_cons_len, int32_t *d_status, int64_t cons_stride,
                    int ncap, int deg, int lmax, int n_slots, void *d_work, size_t work_bytes, hipStream_t s);
int poa_launch(const gbx_poa_params *p, const gbx_poa_plan *plan, int64_t n_windows, const int64_t *d_win_first_seq, const int64_t *d_seq_off,
               const int32_t *d_seq_len, const uint8_t *d_arena,
               uint8_t *d_cons, int32_t *d_cons_len, int32_t *d_status, int64_t cons_stride,
               void *d_work, size_t work_bytes, hipStream_t s);

}  // namespace gbx
