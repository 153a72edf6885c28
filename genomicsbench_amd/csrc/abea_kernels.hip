// abea_kernels.hip — adaptive banded event alignment (f5c / nanopolish) for gfx950 (MI355X).
//
// Semantics: align(), R/benchmarks/abea/src/align.c:169-548 (the suite's CPU path; called per read by
// align_single, f5c.c:1344-1349), bit for bit: float band scores and emissions, double transition penalties
// (every candidate is a double sum rounded to float, :371-373), Suzuki's band placement (:289-307), the trim
// column (:310-319), the traceback with its double emission sum (:409-500) and the three QC rules (:530-541).
// The suite's CUDA path for this step (align.cu) is not a template for this file: it uses one 128-thread block
// per read, a rolling band buffer in shared memory with a block barrier per band, and float penalties.
//
// One read per wavefront, nothing on the way of a band but registers:
//   * a band is an anti-diagonal of 100 cells; lane l owns the cells at offsets 2l and 2l+1.  The two previous bands
//     stay in registers; "up", "left" and "diagonal" are the same registers shifted by -1 / 0 / +1 offsets depending
//     on whether the band moved right or down (own other cell, or the neighbour lane's by DPP wave_shr / wave_shl);
//   * the event means and the scaled model parameters of the lane's cells ride along in registers: a move to the
//     right shifts the k-mer parameters by one offset, a move down the event means; the one value that enters comes
//     from a 64-entry look-ahead block held one per lane and refilled with a coalesced load every 64 moves.  The
//     row loop issues no dependent global load;
//   * per band the wavefront stores 64 trace bytes (two 2-bit back-pointers per lane) and the event index of the
//     band's lower-left corner, which is all the traceback needs; band scores are never stored - the best end cell
//     on the last k-mer's column (:416-432) is tracked on the fly;
//   * the traceback is serial; its trace rows, event means and k-mer parameters come from lane-resident blocks that
//     are refilled with wide loads when the walk leaves them.
// The double penalties lp_stay / lp_step depend on the read (events per k-mer) and are computed on the host with the
// C library's log / exp (gbx_abea_plan_host), exactly as the reference does: device transcendental functions are not
// bit-identical to glibc's.
#include "gbx_internal.h"
#include <cmath>

namespace gbx {
namespace {

constexpr int BW = GBX_ABEA_BANDWIDTH;
constexpr int KSZ = GBX_ABEA_KMER;
constexpr int ROW = 64;                       // trace bytes per band (one per lane)
#define ABEA_NEG_INF (-__builtin_inff())

struct AbeaArgs {
    int n_reads;
    const int64_t *seq_off; const int32_t *seq_len; const char *seq;
    const int64_t *event_off; const float *event_mean;
    const gbx_abea_model *models;
    const float *scale, *shift;
    const int64_t *band_off; const int32_t *order;
    const double *lp;                         // [n_reads][2]: lp_stay, lp_step (host-computed)
    double lp_skip, lp_trim;
    gbx_abea_pair *out; int32_t *n_pairs;
    float *kp_mean, *kp_stdv, *kp_lstd;       // [n_kmers_total] scaled model parameters per k-mer
    uint8_t *trace;                           // [n_bands_total][ROW]
    int32_t *ble;                             // [n_bands_total] event index of the band's lower-left corner
    unsigned *cursor; unsigned long long *cells;
};

template <int CTRL>
__device__ inline float dppf(float old, float x)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, false));
}
__device__ inline float shr1(float x, float fill) { return dppf<0x138>(fill, x); }     // lane l <- lane l-1 (lane 0: fill)
__device__ inline float shl1(float x, float fill) { return dppf<0x130>(fill, x); }     // lane l <- lane l+1 (lane 63: fill)
__device__ inline float rlf(float v, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); }

__device__ inline uint32_t base_rank(char b) { return b == 'C' ? 1u : b == 'G' ? 2u : b == 'T' ? 3u : 0u; }   // align.c:10-24

// log_normal_pdf o log_probability_match_r9, align.c:99-148 (gp_mean = scale*level_mean + shift is precomputed per k-mer)
__device__ inline float lp_match(float x, float gp_mean, float gp_stdv, float gp_log_stdv)
{
    const float log_inv_sqrt_2pi = -0.918938f;
    const float a = (x - gp_mean) / gp_stdv;
    return log_inv_sqrt_2pi - gp_log_stdv + (-0.5f * a * a);
}

__global__ void __launch_bounds__(64) abea_kernel(AbeaArgs A)
{
    const int lane = threadIdx.x;
    const float NINF = ABEA_NEG_INF;
    unsigned long long fills = 0;
    for (;;) {
        unsigned q = 0;
        if (lane == 0) q = atomicAdd(A.cursor, 1u);
        q = (unsigned)__builtin_amdgcn_readfirstlane((int)q);
        if (q >= (unsigned)A.n_reads) break;
        const int r = A.order[q];
        const char *seq = A.seq + A.seq_off[r];
        const int n_kmers = A.seq_len[r] - KSZ + 1;
        const int64_t ev0 = A.event_off[r];
        const int n_events = (int)(A.event_off[r + 1] - ev0);
        const float *evm = A.event_mean + ev0;
        const int64_t boff = A.band_off[r];
        const int64_t koff = boff - (ev0 - A.event_off[0]) - 2 * (int64_t)r;      // bands = events + k-mers + 2 per read
        float *kpm = A.kp_mean + koff, *kps = A.kp_stdv + koff, *kpl = A.kp_lstd + koff;
        uint8_t *trace = A.trace + boff * ROW;
        int32_t *ble = A.ble + boff;
        const int n_bands = n_events + n_kmers + 2;
        const double lp_stay = A.lp[2 * r], lp_step = A.lp[2 * r + 1], lp_skip = A.lp_skip, lp_trim = A.lp_trim;
        const float scale = A.scale[r], shift = A.shift[r];

        // ---- scaled model parameters per k-mer (kmer_ranks, :214-222 + log_probability_match_r9's per-state terms)
        for (int k = lane; k < n_kmers; k += 64) {
            uint32_t rank = 0;
#pragma unroll
            for (int i = 0; i < KSZ; ++i) rank += base_rank(seq[k + KSZ - i - 1]) << (i << 1);
            const gbx_abea_model m = A.models[rank];
            kpm[k] = scale * m.level_mean + shift;
            kps[k] = m.level_stdv * 1;
            kpl[k] = m.level_log_stdv;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        // ---- bands 0 and 1 (:264-279)
        int e_b = BW / 2 - 1, k_b = -1 - BW / 2;                 // lower-left corner of band 0
        const int oA = 2 * lane, oB = 2 * lane + 1;
        float p2a = oA == -1 - k_b ? 0.0f : NINF, p2b = oB == -1 - k_b ? 0.0f : NINF;      // band 0: the start cell
        if (lane == 0) ble[0] = e_b;
        trace[lane] = 0;
        e_b += 1;                                                // band 1 = move_down(band 0)
        float p1a = oA == e_b ? (float)lp_trim : NINF, p1b = oB == e_b ? (float)lp_trim : NINF;   // first event trimmed
        if (lane == 0) ble[1] = e_b;
        trace[ROW + lane] = (uint8_t)((oA == e_b ? 1 : 0) | (oB == e_b ? 4 : 0));       // FROM_U
        // registers that ride along: event means and k-mer parameters of the lane's two cells (all 128 offsets)
        auto ev_at = [&](int idx) -> float { return idx >= 0 && idx < n_events ? evm[idx] : 0.f; };
        float evA = ev_at(e_b - oA), evB = ev_at(e_b - oB);
        auto kidx = [&](int idx) -> int { return idx >= 0 && idx < n_kmers ? idx : 0; };
        float kmA = kpm[kidx(k_b + oA)], kmB = kpm[kidx(k_b + oB)], ksA = kps[kidx(k_b + oA)], ksB = kps[kidx(k_b + oB)];
        float klA = kpl[kidx(k_b + oA)], klB = kpl[kidx(k_b + oB)];
        // look-ahead blocks: lane j holds the (j+1)-th value that will enter on a move down / to the right
        int e_next = e_b + 1, k_next = k_b + 128;                // first event / k-mer index not yet in the registers
        float ebuf = ev_at(e_next + lane);
        float kbm = kpm[kidx(k_next + lane)], kbs = kps[kidx(k_next + lane)], kbl = kpl[kidx(k_next + lane)];
        int ecnt = 0, kcnt = 0;
        bool prev_right = false;                                 // band 1 moved down
        float best_s = NINF;                                     // :416-432, tracked on the fly
        int best_ev = 0;

        for (int b = 2; b < n_bands; ++b) {
            // ---- placement of the band (:289-307)
            const float ll = rlf(p1a, 0), ur = rlf(p1b, BW / 2 - 1);
            const bool ll_ob = ll == NINF, ur_ob = ur == NINF;
            const bool right = (ll_ob && ur_ob) ? (b & 1) == 1 : ll < ur;
            float up_a, up_b, left_a, left_b;
            if (right) {
                k_b += 1;
                // k-mer parameters move one offset down; offset 127 takes the next one of the look-ahead block
                const float im = rlf(kbm, kcnt), is = rlf(kbs, kcnt), il = rlf(kbl, kcnt);
                const float nmB = shl1(kmA, im), nsB = shl1(ksA, is), nlB = shl1(klA, il);
                kmA = kmB; ksA = ksB; klA = klB;
                kmB = nmB; ksB = nsB; klB = nlB;
                if (++kcnt == 64) {
                    k_next += 64; kcnt = 0;
                    kbm = kpm[kidx(k_next + lane)]; kbs = kps[kidx(k_next + lane)]; kbl = kpl[kidx(k_next + lane)];
                }
                up_a = p1b; up_b = shl1(p1a, NINF);              // up = band[b-1][o+1], left = band[b-1][o]
                left_a = p1a; left_b = p1b;
            } else {
                e_b += 1;
                const float ie = rlf(ebuf, ecnt);
                const float neA = shr1(evB, ie);
                evB = evA; evA = neA;
                if (++ecnt == 64) { e_next += 64; ecnt = 0; ebuf = ev_at(e_next + lane); }
                up_a = p1a; up_b = p1b;                          // up = band[b-1][o], left = band[b-1][o-1]
                left_b = p1a; left_a = shr1(p1b, NINF);
            }
            // diagonal = band[b-2][o - 1 + (rights among the last two moves)]
            float dg_a, dg_b;
            const int nr2 = (right ? 1 : 0) + (prev_right ? 1 : 0);
            if (nr2 == 2) { dg_a = p2b; dg_b = shl1(p2a, NINF); }
            else if (nr2 == 1) { dg_a = p2a; dg_b = p2b; }
            else { dg_b = p2a; dg_a = shr1(p2b, NINF); }
            prev_right = right;

            // ---- the cells this band may fill (:323-332) and the trim cell (:310-319)
            const int min_off = max(max(0 - k_b, e_b - (n_events - 1)), 0);
            const int max_off = min(min(n_kmers - k_b, e_b + 1), BW);
            const int trim_off = -1 - k_b;
            const int trim_ev = e_b - trim_off;
            const bool trim_ok = trim_off >= 0 && trim_off < BW && trim_ev >= 0 && trim_ev < n_events;
            const float trim_val = (float)(lp_trim * (double)((int64_t)trim_ev + 1));

            auto cell = [&](int o, float diag, float up, float left, float x, float gm, float gs, float gl, float &val, int &from) {
                const float lpe = lp_match(x, gm, gs, gl);
                const float score_d = (float)(((double)diag + lp_step) + (double)lpe);       // :371-373
                const float score_u = (float)(((double)up + lp_stay) + (double)lpe);
                const float score_l = (float)((double)left + lp_skip);
                float max_score = score_d;
                int f = 0;                                                                    // FROM_D
                max_score = score_u > max_score ? score_u : max_score;
                f = max_score == score_u ? 1 : f;                                             // FROM_U
                max_score = score_l > max_score ? score_l : max_score;
                f = max_score == score_l ? 2 : f;                                             // FROM_L
                const bool fill = o >= min_off && o < max_off;
                const bool trim = trim_ok && o == trim_off;
                val = fill ? max_score : trim ? trim_val : NINF;
                from = fill ? f : trim ? 1 : 0;
            };
            float na, nb;
            int fa, fb;
            cell(oA, dg_a, up_a, left_a, evA, kmA, ksA, klA, na, fa);
            cell(oB, dg_b, up_b, left_b, evB, kmB, ksB, klB, nb, fb);
            trace[(int64_t)b * ROW + lane] = (uint8_t)(fa | (fb << 2));
            if (lane == 0) ble[b] = e_b;
            fills += (unsigned long long)max(min(max_off, BW) - min_off, 0);

            // ---- best end on the last k-mer's column (:416-432): the cell (event, n_kmers-1) of this band, if any
            {
                const int ev = b - n_kmers - 1, ol = n_kmers - 1 - k_b;
                if (ev >= 0 && ev < n_events && ol >= 0 && ol < BW) {                   // e_b - ev == ol on this band
                    const float v = (ol & 1) ? rlf(nb, ol >> 1) : rlf(na, ol >> 1);
                    const float s = (float)((double)v + (double)(size_t)(n_events - ev) * lp_trim);
                    if (s > best_s) { best_s = s; best_ev = ev; }
                }
            }
            p2a = p1a; p2b = p1b; p1a = na; p1b = nb;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        // ---- traceback (:409-500): serial, wave-uniform; the data it walks through comes from lane-resident blocks
        gbx_abea_pair *out = A.out + 2 * ev0;
        const int cap = 2 * n_events;
        int ce = best_ev, ck = n_kmers - 1, n_out = 0, curr_gap = 0, max_gap = 0;
        double sum_emission = 0;
        // trace rows [tb_top-15, tb_top] as 16 bytes per lane: lane j holds bytes (j&3)*16.. of row tb_top - (j>>2)
        int tb_top = -1;
        uint4 tb = make_uint4(0, 0, 0, 0);
        int bl_top = -1; int blv = 0;                           // ble[bl_top - lane]
        int ev_top = -1; float evv = 0.f;                       // evm[ev_top - lane]
        int km_top = -1; float kmv = 0.f, ksv = 0.f, klv = 0.f; // k-mer parameters [km_top - lane]
        while (ck >= 0 && ce >= 0) {
            if (n_out < cap && lane == 0) { out[n_out].ref_pos = ck; out[n_out].read_pos = ce; }
            ++n_out;
            if (ce > ev_top || ce <= ev_top - 64) { ev_top = ce; const int i = ce - lane; evv = i >= 0 ? evm[i] : 0.f; }
            if (ck > km_top || ck <= km_top - 64) { km_top = ck; const int i = max(ck - lane, 0); kmv = kpm[i]; ksv = kps[i]; klv = kpl[i]; }
            const float lpe = lp_match(rlf(evv, ev_top - ce), rlf(kmv, km_top - ck), rlf(ksv, km_top - ck), rlf(klv, km_top - ck));
            sum_emission += (double)lpe;
            const int bi = (ce + 1) + (ck + 1);
            if (bi > bl_top || bi <= bl_top - 64) { bl_top = bi; blv = ble[max(bi - lane, 0)]; }
            const int off = __builtin_amdgcn_readlane(blv, bl_top - bi) - ce;
            if (bi > tb_top || bi <= tb_top - 16) {
                tb_top = bi;
                const int row = max(bi - (lane >> 2), 0);
                tb = *(const uint4 *)(trace + (int64_t)row * ROW + (lane & 3) * 16);
            }
            const int tl = off >> 1, src = ((tb_top - bi) << 2) + (tl >> 4), w = (tl & 15) >> 2;
            const unsigned d0 = (unsigned)__builtin_amdgcn_readlane((int)tb.x, src), d1 = (unsigned)__builtin_amdgcn_readlane((int)tb.y, src);
            const unsigned d2 = (unsigned)__builtin_amdgcn_readlane((int)tb.z, src), d3 = (unsigned)__builtin_amdgcn_readlane((int)tb.w, src);
            const unsigned dw = w == 0 ? d0 : w == 1 ? d1 : w == 2 ? d2 : d3;
            const int from = (int)((dw >> ((tl & 3) * 8 + (off & 1) * 2)) & 3u);
            if (from == 0) { ck -= 1; ce -= 1; curr_gap = 0; }
            else if (from == 1) { ce -= 1; curr_gap = 0; }
            else { ck -= 1; curr_gap += 1; max_gap = max(curr_gap, max_gap); }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // std::reverse, then the QC rules (:530-541).  The reference's array holds 2 x n_events pairs; a walk longer
        // than that (only possible with far more k-mers than events) is reported as failed instead of overrunning it.
        const int n_w = min(n_out, cap);
        for (int c = lane; c < n_w / 2; c += 64) { const gbx_abea_pair t = out[c]; out[c] = out[n_w - 1 - c]; out[n_w - 1 - c] = t; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const double avg_log_emission = sum_emission / (double)n_out;
        bool ok = n_out >= 1 && n_out <= cap;
        // spanned: out.front() sits on k-mer 0 (the walk's last pair) and out.back() on the last k-mer (its first)
        const int first_ref = ok ? out[0].ref_pos : -1, last_ref = ok ? out[n_w - 1].ref_pos : -1;
        ok = ok && first_ref == 0 && last_ref == n_kmers - 1;
        ok = ok && !(avg_log_emission < -5.0) && !(max_gap > 50);
        if (lane == 0) A.n_pairs[r] = ok ? n_out : 0;
    }
    if (lane == 0) atomicAdd(A.cells, fills);
}

}  // namespace

// workspace: [256 B header: cursor, cells] [3 float arrays per k-mer] [trace rows] [band corners]
static size_t abea_align_up(size_t x) { return (x + 255) & ~(size_t)255; }
size_t abea_workspace_bytes(int64_t n_reads, int64_t n_kmers_total, int64_t n_bands_total)
{
    (void)n_reads;
    const size_t nk = (size_t)(n_kmers_total > 0 ? n_kmers_total : 0) + 64, nb = (size_t)(n_bands_total > 0 ? n_bands_total : 0) + 16;
    return 256 + 3 * abea_align_up(nk * 4) + abea_align_up(nb * ROW) + abea_align_up(nb * 4) + abea_align_up((size_t)(n_reads > 0 ? n_reads : 0) * 16 + 16);
}

int abea_read_cells(const void *d_work, int64_t *cells, hipStream_t s)
{
    unsigned long long v = 0;
    GBX_HIP(hipMemcpyAsync(&v, (const char *)d_work + 8, sizeof(v), hipMemcpyDeviceToHost, s));
    GBX_HIP(hipStreamSynchronize(s));
    *cells = (int64_t)v;
    return GBX_OK;
}

int abea_launch(int64_t n_reads, const int64_t *d_seq_off, const int32_t *d_seq_len, const char *d_seq,
                const int64_t *d_event_off, const float *d_event_mean, const gbx_abea_model *d_models,
                const float *d_scale, const float *d_shift, const int64_t *d_band_off, const int32_t *d_order,
                const double *d_lp, int64_t n_kmers_total, int64_t n_bands_total,
                gbx_abea_pair *d_out, int32_t *d_n_pairs, void *d_work, size_t work_bytes, hipStream_t s)
{
    if (n_reads == 0) return GBX_OK;
    if (n_reads > 0x7fffffffLL - 1024) { set_error("abea: more than 2^31 reads in one call"); return GBX_ERR_UNSUPPORTED; }
    if (work_bytes < abea_workspace_bytes(n_reads, n_kmers_total, n_bands_total)) { set_error("abea: workspace too small"); return GBX_ERR_ARG; }
    const size_t nk = (size_t)n_kmers_total + 64, nb = (size_t)n_bands_total + 16;
    char *w = (char *)d_work;
    AbeaArgs A;
    A.n_reads = (int)n_reads;
    A.seq_off = d_seq_off; A.seq_len = d_seq_len; A.seq = d_seq; A.event_off = d_event_off; A.event_mean = d_event_mean;
    A.models = d_models; A.scale = d_scale; A.shift = d_shift; A.band_off = d_band_off; A.order = d_order; A.lp = d_lp;
    A.lp_skip = log(1e-10); A.lp_trim = log(0.01);                            // align.c:201-205, the C library's log
    A.out = d_out; A.n_pairs = d_n_pairs;
    A.cursor = (unsigned *)w; A.cells = (unsigned long long *)(w + 8);
    size_t o = 256;
    A.kp_mean = (float *)(w + o); o += abea_align_up(nk * 4);
    A.kp_stdv = (float *)(w + o); o += abea_align_up(nk * 4);
    A.kp_lstd = (float *)(w + o); o += abea_align_up(nk * 4);
    A.trace = (uint8_t *)(w + o); o += abea_align_up(nb * ROW);
    A.ble = (int32_t *)(w + o);
    GBX_HIP(hipMemsetAsync(d_work, 0, 256, s));
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, abea_kernel, 64, 0) != hipSuccess || per_cu < 1) { (void)hipGetLastError(); per_cu = 8; }
    if (per_cu > 16) per_cu = 16;
    const int64_t slots = (int64_t)cus * per_cu;
    const int grid = (int)(n_reads < slots ? n_reads : slots);
    {
        Stage st("abea_align", s);
        hipLaunchKernelGGL(abea_kernel, dim3(grid), dim3(64), 0, s, A);
    }
    GBX_HIP(hipGetLastError());
    return GBX_OK;
}

}  // namespace gbx
