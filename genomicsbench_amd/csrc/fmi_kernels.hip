// fmi_kernels.hip — bwa-mem2 SMEM seeding on the FM-index for gfx950 (MI355X).
//
// Semantics: the three seeding rounds R/benchmarks/fmi/fmi.cpp:218-278 runs per batch of reads
// (FMI_search::getSMEMsAllPosOneThread, the re-seeding call of getSMEMsOnePosOneThread, bwtSeedStrategyAllPosOneThread,
// sortSMEMs), restated in oracle/fmi_oracle.c from bwa-mem2's published FMI_search.cpp; bit-exact on every field.
//
// What bounds it: one backwardExt is two dependent look-ups in the checkpoint table (rows k and k + s), each a
// random 64-byte line of a table far larger than the caches, and a read is a chain of ~700 of them.  Nothing to
// compute, everything to wait for: the design keeps as many independent line requests in flight as the chip holds
// wavefronts.
//   * four lanes per read, lane b = base b: the device index stores a checkpoint as four {count, one-hot word}
//     pairs, so a look-up is ONE 16-byte load per lane and the quad's four loads are one 64-byte line.  16 reads per
//     wavefront, 8 wavefronts per SIMD.
//   * the rounds of a read are a state machine with a single extension site: every trip of the main loop each
//     read of the wavefront performs exactly one backwardExt (forward extension = backward extension of the reverse
//     complement), whatever round and phase it is in, so the wavefront's 32 line requests are always issued together;
//     the bookkeeping between two extensions (start a position, close a forward sweep, output an SMEM, next read)
//     is scalar-per-quad code that runs without memory waits.
//   * the prev[] array of the backward sweep lives in a per-quad slab in HBM, 32 bytes per entry = one 8-byte word per
//     lane of the quad (k, l, s, n); it is short (the interval size changes ~15 times along a forward sweep) and
//     stays in L2.  Vector memory operations of one wavefront are performed in order, so a quad reads back what it
//     wrote without a fence.
//   * reads are drawn from a cursor; SMEMs go to a fixed-capacity slot per read, a second kernel sorts each read's
//     few records by (m ascending, n descending) and packs them behind a prefix sum of the counts.
#include <algorithm>
#include "gbx_internal.h"

namespace gbx {
namespace {

// SMEMs of one read before the pack pass: 48 for short reads (the reference sizes its array at 20 per read and the
// bench's 151-bp reads give 8.5 on average, 18 at most), 3/8 of the read length for long ones (a 700-base read: 56)
__host__ __device__ inline int raw_cap_for(int max_len) { const int c = (max_len * 3 + 7) / 8; return c > 48 ? c : 48; }
constexpr int SCAN_BLOCK = 1024;       // reads per block of the count scan

struct FmiArgs {
    const uint4 *index;                // device layout: checkpoint i, base b -> {count lo, count hi, one-hot lo, one-hot hi} at 4 i + b
    long long count[5];
    long long sentinel;
    int min_seed_len, split_width, split_len, max_intv;
    long long n_reads;                 // of this launch (a chunk)
    long long read_base;               // first read of the chunk (rid = read_base + r)
    const uint8_t *enc;
    const int64_t *read_off;
    const int32_t *read_len;
    int max_len, raw_cap;
    uint2 *raw;                        // [chunk reads][raw_cap][5] 8-byte words: {rid, m} {n, 0} k l s
    int32_t *raw_count;                // [chunk reads]
    uint2 *prev;                       // [resident quads][max_len + 1][4] 8-byte words
    unsigned long long *counters;      // [0] read cursor, [1] extensions, [2] overflow flag, [3] running output total
};

template <int SEL> __device__ inline unsigned quad_bcast(unsigned v)
{
    return (unsigned)__builtin_amdgcn_mov_dpp((int)v, SEL * 0x55, 0xf, 0xf, true);      // quad_perm:[SEL,SEL,SEL,SEL]
}
template <int SEL> __device__ inline long long quad_bcast64(long long v)
{
    const unsigned lo = quad_bcast<SEL>((unsigned)v), hi = quad_bcast<SEL>((unsigned)((unsigned long long)v >> 32));
    return (long long)(((unsigned long long)hi << 32) | lo);
}
__device__ inline long long pick4(int a, long long v0, long long v1, long long v2, long long v3)
{
    return a == 0 ? v0 : a == 1 ? v1 : a == 2 ? v2 : v3;
}
__device__ inline long long u2ll(uint2 v) { return (long long)(((unsigned long long)v.y << 32) | v.x); }
__device__ inline uint2 ll2u(long long v) { return make_uint2((unsigned)v, (unsigned)((unsigned long long)v >> 32)); }

enum : int { ST_FWD = 0, ST_BWD = 1, ST_SEED = 2 };
// transitions between two extensions (T_EXT: the read needs one now; T_IDLE: no reads left for this quad)
enum : int { T_EXT, T_IDLE, T_NEXT_READ, T_ONEPOS_INIT, T_FWD_CHECK, T_FWD_END, T_BWD_J, T_BWD_P, T_BWD_JEND, T_BWD_END, T_ONEPOS_DONE,
             T_P2_NEXT, T_SEED_INIT, T_SEED_CHECK, T_READ_DONE };

typedef __attribute__((address_space(3))) unsigned char lds_u8;

// LDSQ: the quad's read is staged in LDS (qstride bytes per quad) so that the base behind every decision is a
// ds_read_u8 instead of a global load on the dependent chain; reads too long for that are read in place.
template <bool LDSQ>
__global__ void __launch_bounds__(64, 8) fmi_smem_kernel(FmiArgs A, int qstride)
{
    extern __shared__ unsigned char q_lds[];
    const int lane = threadIdx.x, b = lane & 3;
    const long long quad_id = (long long)blockIdx.x * 16 + (lane >> 2);
    uint2 *const prev = A.prev + (size_t)quad_id * (size_t)(A.max_len + 1) * 4 + b;        // entry e: prev[4 e]
    unsigned char *const ql = q_lds + (size_t)(lane >> 2) * (size_t)qstride;
    const long long c0 = A.count[0], c1 = A.count[1], c2 = A.count[2], c3 = A.count[3], c4 = A.count[4];
    const long long count_b = b == 0 ? c0 : b == 1 ? c1 : b == 2 ? c2 : c3;
    auto cnt_of = [&](int c) -> long long { return c == 0 ? c0 : c == 1 ? c1 : c == 2 ? c2 : c == 3 ? c3 : c4; };
    unsigned long long n_ext = 0;
    const int RAW_CAP = A.raw_cap;

    // quad-uniform state
    int t = T_NEXT_READ, state = ST_FWD;
    long long rid_local = 0;                 // read of this chunk
    const uint8_t *q = nullptr;
    int len = 0, n_out = 0;
    int phase = 0;                           // 0: every start position, 1: re-seeding, 2: LAST round
    int x = 0, j = 0, next_x = 0, min_intv = 1;
    long long k = 0, l = 0, s = 0;           // the match being extended forwards, [x, n]
    int n = 0;
    int num_prev = 0, vbase = 0, p = 0, num_curr = 0, curr_s = -1, m_cur = 0, a = 0;
    bool first = true;
    long long pk = 0, pl = 0, ps = 0;        // prev[p] of the backward sweep
    int pn = 0;
    int idx2 = 0, n1 = 0;
    uint2 *raw = nullptr;

    auto base_at = [&](int i) -> int { return LDSQ ? (int)((const volatile lds_u8 *)ql)[i] : (int)q[i]; };
    auto emit = [&](int m_, int n_, long long k_, long long l_, long long s_) {
        if (n_out < RAW_CAP) {
            uint2 *e = raw + (size_t)n_out * 5;
            const long long rid = A.read_base + rid_local;
            // the quad writes the record's five words: lanes 0..3 the first four, lane 0 the fifth
            const uint2 w = b == 0 ? make_uint2((unsigned)rid, (unsigned)m_) : b == 1 ? make_uint2((unsigned)n_, 0u) : b == 2 ? ll2u(k_) : ll2u(l_);
            e[b] = w;
            if (b == 0) e[4] = ll2u(s_);
        }
        ++n_out;
    };
    auto load_prev = [&](int view) {          // view[p] = arr[vbase - p]: the forward sweep's array read backwards
        const uint2 w = prev[(size_t)(vbase - view) * 4];
        const long long v = u2ll(w);
        pk = quad_bcast64<0>(v); pl = quad_bcast64<1>(v); ps = quad_bcast64<2>(v); pn = (int)quad_bcast<3>(w.x);
    };
    auto store_prev_arr = [&](int arr_index, long long k_, long long l_, long long s_, int n_) {
        prev[(size_t)arr_index * 4] = b == 0 ? ll2u(k_) : b == 1 ? ll2u(l_) : b == 2 ? ll2u(s_) : make_uint2((unsigned)n_, 0u);
    };
    auto seed_interval = [&](int c) { k = cnt_of(c); l = cnt_of(3 - c); s = cnt_of(c + 1) - cnt_of(c); };

    for (;;) {
        // ---- bookkeeping until this read needs an extension (or there is nothing left to do)
        while (t != T_EXT && t != T_IDLE) {
            switch (t) {
            case T_NEXT_READ: {
                long long r = 0;
                if (b == 0) r = (long long)atomicAdd(&A.counters[0], 1ull);
                r = quad_bcast64<0>(r);
                if (r >= A.n_reads) { t = T_IDLE; break; }
                rid_local = r;
                len = A.read_len[A.read_base + r];
                q = A.enc + A.read_off[A.read_base + r];
                raw = A.raw + (size_t)r * RAW_CAP * 5;
                n_out = 0;
                if (len <= 0) { t = T_READ_DONE; break; }
                if (LDSQ)
                    for (int i = b; i < len; i += 4) ql[i] = q[i];
                phase = 0; x = 0; min_intv = 1;
                t = T_ONEPOS_INIT;
                break;
            }
            case T_ONEPOS_INIT:                                   // getSMEMsOnePosOneThread, one (read, x, min_intv)
                a = base_at(x);
                next_x = x + 1;
                if (a >= 4) { t = T_ONEPOS_DONE; break; }
                seed_interval(a);
                n = x; num_prev = 0; j = x + 1;
                t = T_FWD_CHECK;
                break;
            case T_FWD_CHECK:
                if (j >= len) { t = T_FWD_END; break; }
                a = base_at(j);
                next_x = j + 1;
                if (a >= 4) { t = T_FWD_END; break; }
                state = ST_FWD; t = T_EXT;
                break;
            case T_FWD_END:
                if (s >= min_intv) { store_prev_arr(num_prev, k, l, s, n); ++num_prev; }
                vbase = num_prev - 1; j = x - 1; m_cur = x;
                t = T_BWD_J;
                break;
            case T_BWD_J:
                if (j < 0) { t = T_BWD_END; break; }
                a = base_at(j);
                if (a > 3) { t = T_BWD_END; break; }
                num_curr = 0; curr_s = -1; p = 0; first = true;
                t = T_BWD_P;
                break;
            case T_BWD_P:
                if (p >= num_prev) { t = T_BWD_JEND; break; }
                load_prev(p);
                state = ST_BWD; t = T_EXT;
                break;
            case T_BWD_JEND:
                num_prev = num_curr;
                if (num_curr == 0) { t = T_ONEPOS_DONE; break; }
                m_cur = j; --j;
                t = T_BWD_J;
                break;
            case T_BWD_END:
                if (num_prev != 0) {
                    load_prev(0);
                    if (pn - m_cur + 1 >= A.min_seed_len) emit(m_cur, pn, pk, pl, ps);
                }
                t = T_ONEPOS_DONE;
                break;
            case T_ONEPOS_DONE:
                if (phase == 0) {
                    x = next_x;                                   // getSMEMsAllPosOneThread: on to the next start position
                    if (x < len) { t = T_ONEPOS_INIT; break; }
                    n1 = n_out < RAW_CAP ? n_out : RAW_CAP; idx2 = 0; phase = 1;
                }
                t = T_P2_NEXT;
                break;
            case T_P2_NEXT: {                                     // fmi.cpp:230-254: re-seed from the middle of long, rare SMEMs
                if (idx2 >= n1) { phase = 2; x = 0; t = T_SEED_INIT; break; }
                const uint2 *e = raw + (size_t)idx2 * 5;
                const uint2 w = e[b == 0 ? 0 : b == 1 ? 1 : 4];
                const int m_ = (int)quad_bcast<0>(w.y), n_ = (int)quad_bcast<1>(w.x);
                const long long s_ = quad_bcast64<2>(u2ll(w));
                ++idx2;
                const int start = m_, end = n_ + 1;
                if (end - start < A.split_len || s_ > A.split_width) break;
                x = (end + start) >> 1; min_intv = (int)(s_ + 1);
                t = T_ONEPOS_INIT;
                break;
            }
            case T_SEED_INIT:                                     // bwtSeedStrategyAllPosOneThread
                if (x >= len) { t = T_READ_DONE; break; }
                a = base_at(x);
                next_x = x + 1;
                if (a >= 4) { x = next_x; break; }
                seed_interval(a);
                n = x; j = x + 1;
                t = T_SEED_CHECK;
                break;
            case T_SEED_CHECK:
                if (j >= len) { x = next_x; t = T_SEED_INIT; break; }
                next_x = j + 1;
                a = base_at(j);
                if (a >= 4) { x = next_x; t = T_SEED_INIT; break; }
                state = ST_SEED; t = T_EXT;
                break;
            case T_READ_DONE:
                if (b == 0) {
                    A.raw_count[rid_local] = n_out < RAW_CAP ? n_out : RAW_CAP;
                    if (n_out > RAW_CAP) atomicMax(&A.counters[2], (unsigned long long)n_out);
                }
                t = T_NEXT_READ;
                break;
            }
        }
        if (__ballot(t == T_EXT) == 0) break;

        // ---- one backwardExt per read (FMI_search.cpp): rows sp = k and ep = k + s of the checkpoint table
        const bool act = t == T_EXT;
        long long ek = 0, el = 0, es = 0;
        int ea = 0;
        if (act) {
            if (state == ST_BWD) { ek = pk; el = pl; es = ps; ea = a; }
            else { ek = l; el = k; es = s; ea = 3 - a; }            // forwards = backwards on the reverse complement
        }
        const long long sp = ek, ep = ek + es;
        const uint4 csp = A.index[(size_t)(sp >> 6) * 4 + b], cep = A.index[(size_t)(ep >> 6) * 4 + b];
        const int ysp = (int)(sp & 63), yep = (int)(ep & 63);
        const unsigned long long msp = ysp ? ~0ull << (64 - ysp) : 0ull, mep = yep ? ~0ull << (64 - yep) : 0ull;
        const long long occ_sp = (long long)(((unsigned long long)csp.y << 32) | csp.x) +
                                 __builtin_popcountll((((unsigned long long)csp.w << 32) | csp.z) & msp);
        const long long occ_ep = (long long)(((unsigned long long)cep.y << 32) | cep.x) +
                                 __builtin_popcountll((((unsigned long long)cep.w << 32) | cep.z) & mep);
        const long long kb = count_b + occ_sp, sb = occ_ep - occ_sp;
        const long long s1 = quad_bcast64<1>(sb), s2 = quad_bcast64<2>(sb), s3 = quad_bcast64<3>(sb), s0 = quad_bcast64<0>(sb);
        const long long l3 = el + ((ek <= A.sentinel && ek + es > A.sentinel) ? 1 : 0);
        const long long l2 = l3 + s3, l1 = l2 + s2, l0 = l1 + s1;
        const long long rl = pick4(ea, l0, l1, l2, l3), rs = pick4(ea, s0, s1, s2, s3);
        const int src = (lane & ~3) | ea;
        const long long rk = (long long)(((unsigned long long)(unsigned)__shfl((int)((unsigned long long)kb >> 32), src) << 32) |
                                         (unsigned)__shfl((int)(unsigned)kb, src));
        if (act) {
            ++n_ext;
            if (state == ST_FWD) {
                const long long nk = rl, nl = rk, ns = rs;
                if (ns != s) { store_prev_arr(num_prev, k, l, s, n); ++num_prev; }
                if (ns < min_intv) { next_x = j; t = T_FWD_END; }
                else { k = nk; l = nl; s = ns; n = j; ++j; t = T_FWD_CHECK; }
            } else if (state == ST_BWD) {
                const long long ns = rs;
                if (first && ns < min_intv && pn - m_cur + 1 >= A.min_seed_len) {
                    emit(m_cur, pn, pk, pl, ps);
                    first = false;
                } else if (ns >= min_intv && ns != (long long)curr_s) {
                    curr_s = (int)ns;
                    store_prev_arr(vbase - num_curr, rk, rl, ns, pn);
                    ++num_curr;
                    first = false;
                }
                ++p;
                t = T_BWD_P;
            } else {
                k = rl; l = rk; s = rs; n = j;
                if (s < A.max_intv && n - x + 1 >= A.min_seed_len + 1) {
                    if (s > 0) emit(x, n, k, l, s);
                    x = next_x;
                    t = T_SEED_INIT;
                } else {
                    ++j;
                    t = T_SEED_CHECK;
                }
            }
        }
    }
    // extensions of the wavefront: one count per quad
    if (b != 0) n_ext = 0;
    for (int d = 32; d; d >>= 1) n_ext += __shfl_xor((unsigned long long)n_ext, d);
    if (lane == 0 && n_ext) atomicAdd(&A.counters[1], n_ext);
}

// ---- counts -> offsets -> sorted, packed records
__global__ void __launch_bounds__(SCAN_BLOCK) fmi_scan1_kernel(const int32_t *cnt, long long n, long long *block_sum)
{
    __shared__ long long sh[SCAN_BLOCK / 64];
    const long long i = (long long)blockIdx.x * SCAN_BLOCK + threadIdx.x;
    long long v = i < n ? cnt[i] : 0;
    for (int d = 32; d; d >>= 1) v += __shfl_xor(v, d);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        long long t = 0;
        for (int w = 0; w < SCAN_BLOCK / 64; ++w) t += sh[w];
        block_sum[blockIdx.x] = t;
    }
}

// one block: exclusive scan of the block sums; counters[4] = records before this chunk, counters[3] += this chunk's
__global__ void __launch_bounds__(1024) fmi_scan2_kernel(long long *block_sum, int n_blocks, unsigned long long *counters, int64_t *d_n_out)
{
    __shared__ long long sh[1024];
    __shared__ long long carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int b0 = 0; b0 < n_blocks; b0 += 1024) {
        const int i = b0 + threadIdx.x;
        const long long v = i < n_blocks ? block_sum[i] : 0;
        sh[threadIdx.x] = v;
        __syncthreads();
        for (int d = 1; d < 1024; d <<= 1) {
            const long long u = threadIdx.x >= d ? sh[threadIdx.x - d] : 0;
            __syncthreads();
            sh[threadIdx.x] += u;
            __syncthreads();
        }
        if (i < n_blocks) block_sum[i] = carry + sh[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry += sh[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        counters[4] = counters[3];
        counters[3] += (unsigned long long)carry;
        *d_n_out = (int64_t)counters[3];
    }
}

// one thread per read: its offset, then its records in (m ascending, n descending) order (sortSMEMs' compare_smem
// within one rid; records equal in (m, n) are equal in every field)
__global__ void __launch_bounds__(SCAN_BLOCK) fmi_pack_kernel(const int32_t *cnt, long long n, long long read_base, const long long *block_off,
                                                              const unsigned long long *counters, const uint2 *raw, int RAW_CAP,
                                                              gbx_fmi_smem *out, long long out_cap, int64_t *smem_off)
{
    __shared__ long long sh[SCAN_BLOCK / 64];
    const long long i = (long long)blockIdx.x * SCAN_BLOCK + threadIdx.x;
    const int c = i < n ? cnt[i] : 0;
    // exclusive prefix inside the block
    long long v = c;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int d = 1; d < 64; d <<= 1) { const long long u = __shfl_up(v, d); if (lane >= d) v += u; }
    if (lane == 63) sh[wv] = v;
    __syncthreads();
    long long before = 0;
    for (int w = 0; w < wv; ++w) before += sh[w];
    const long long off = (long long)counters[4] + block_off[blockIdx.x] + before + v - c;
    if (i >= n) return;
    smem_off[read_base + i] = off;
    if (i == n - 1) smem_off[read_base + n] = off + c;
    if (off + c > out_cap) return;
    const uint2 *e = raw + (size_t)i * RAW_CAP * 5;
    for (int a = 0; a < c; ++a) {
        const unsigned ma = e[a * 5].y, na = e[a * 5 + 1].x;
        int rank = 0;
        for (int o = 0; o < c; ++o) {
            const unsigned mo = e[o * 5].y, no = e[o * 5 + 1].x;
            rank += (mo < ma) || (mo == ma && (no > na || (no == na && o < a)));
        }
        uint2 *dst = (uint2 *)(out + off + rank);
        for (int w = 0; w < 5; ++w) dst[w] = e[a * 5 + w];
    }
}

// host layout (bwa-mem2 CP_OCC: four counts, then four one-hot words) -> device layout ({count, one-hot} per base)
__global__ void __launch_bounds__(256) fmi_index_kernel(const gbx_fmi_cp_occ *src, long long n_cp, uint4 *dst)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_cp * 4) return;
    const long long cp = i >> 2;
    const int b = (int)(i & 3);
    const unsigned long long c = (unsigned long long)src[cp].cp_count[b], o = src[cp].one_hot_bwt_str[b];
    dst[i] = make_uint4((unsigned)c, (unsigned)(c >> 32), (unsigned)o, (unsigned)(o >> 32));
}

constexpr long long FMI_CHUNK = 4ll << 20;         // reads per launch: bounds the raw slots (4 Mi x 48 x 40 B = 7.7 GB; fewer for long reads)
constexpr long long FMI_MAX_QUADS = 256ll * 32 * 16;   // resident quads on 256 CUs at 8 wavefronts per SIMD

struct FmiLayout { size_t o_raw, o_cnt, o_prev, o_bsum, total; long long chunk, quads; };
static FmiLayout fmi_layout(int64_t n_reads, int32_t max_len)
{
    FmiLayout L;
    const int RAW_CAP = raw_cap_for(max_len);
    long long cap_chunk = FMI_CHUNK * 48 / RAW_CAP;
    if (cap_chunk < 16) cap_chunk = 16;
    L.chunk = n_reads < cap_chunk ? (n_reads > 0 ? n_reads : 1) : cap_chunk;
    const long long want = (L.chunk + 15) / 16 * 16;
    L.quads = want < FMI_MAX_QUADS ? want : FMI_MAX_QUADS;
    size_t at = 64;                                                        // counters: 8 x u64
    auto take = [&](size_t bytes) { const size_t o = at; at += (bytes + 255) & ~(size_t)255; return o; };
    L.o_raw = take((size_t)L.chunk * RAW_CAP * 40);
    L.o_cnt = take((size_t)L.chunk * 4);
    L.o_prev = take((size_t)L.quads * (size_t)(max_len + 1) * 32);
    L.o_bsum = take((size_t)((L.chunk + SCAN_BLOCK - 1) / SCAN_BLOCK) * 8);
    L.total = at;
    return L;
}

}  // namespace

size_t fmi_index_bytes(int64_t ref_seq_len) { return ref_seq_len > 0 ? (size_t)((ref_seq_len >> 6) + 1) * 64 : 0; }

int fmi_index_build(const gbx_fmi_index *idx, void *d_index, size_t index_bytes, hipStream_t s)
{
    if (index_bytes < fmi_index_bytes(idx->ref_seq_len)) { set_error("fmi: device index buffer too small"); return GBX_ERR_ARG; }
    const long long n_cp = (idx->ref_seq_len >> 6) + 1;
    Stage st("fmi_index", s);
    hipLaunchKernelGGL(fmi_index_kernel, dim3((unsigned)((n_cp * 4 + 255) / 256)), dim3(256), 0, s, idx->cp_occ, n_cp, (uint4 *)d_index);
    GBX_HIP(hipGetLastError());
    return GBX_OK;
}

size_t fmi_workspace_bytes(int64_t n_reads, int32_t max_len) { return fmi_layout(n_reads, max_len).total; }

int fmi_read_extensions(const void *d_work, int64_t *ext, hipStream_t s)
{
    unsigned long long v = 0;
    GBX_HIP(hipMemcpyAsync(&v, (const char *)d_work + 8, sizeof(v), hipMemcpyDeviceToHost, s));
    GBX_HIP(hipStreamSynchronize(s));
    *ext = (int64_t)v;
    return GBX_OK;
}

int fmi_launch(const gbx_fmi_index *idx, const void *d_index, const gbx_fmi_params *p, int64_t n_reads, int32_t max_len,
               const uint8_t *d_enc, const int64_t *d_read_off, const int32_t *d_read_len, gbx_fmi_smem *d_out, int64_t out_cap,
               int64_t *d_smem_off, int64_t *d_n_out, void *d_work, size_t work_bytes, hipStream_t s)
{
    if (max_len < 0 || max_len > 65535) { set_error("fmi: reads of up to 65535 bases (the reference asserts 10000, fmi.cpp:93)"); return GBX_ERR_UNSUPPORTED; }
    if (idx->ref_seq_len < 2 || idx->ref_seq_len >= (1ll << 40)) { set_error("fmi: bad reference length"); return GBX_ERR_ARG; }
    const FmiLayout L = fmi_layout(n_reads, max_len);
    if (work_bytes < L.total) { set_error("fmi: workspace too small"); return GBX_ERR_ARG; }
    char *wb = (char *)d_work;
    unsigned long long *counters = (unsigned long long *)wb;
    GBX_HIP(hipMemsetAsync(counters, 0, 64, s));
    if (n_reads == 0) {
        GBX_HIP(hipMemsetAsync(d_smem_off, 0, 8, s));
        GBX_HIP(hipMemsetAsync(d_n_out, 0, 8, s));
        return GBX_OK;
    }
    FmiArgs A;
    A.index = (const uint4 *)d_index;
    for (int c = 0; c < 5; ++c) A.count[c] = idx->count[c];
    A.sentinel = idx->sentinel_index;
    A.min_seed_len = p->min_seed_len; A.split_width = p->split_width; A.split_len = p->split_len; A.max_intv = p->max_mem_intv;
    A.enc = d_enc; A.read_off = d_read_off; A.read_len = d_read_len; A.max_len = max_len; A.raw_cap = raw_cap_for(max_len);
    A.raw = (uint2 *)(wb + L.o_raw); A.raw_count = (int32_t *)(wb + L.o_cnt); A.prev = (uint2 *)(wb + L.o_prev);
    A.counters = counters;
    long long *bsum = (long long *)(wb + L.o_bsum);
    // reads of up to 320 bases are staged in LDS at full occupancy (16 quads x 320 B x 32 wavefronts = the CU's 160 KB);
    // up to 2048 bases at lower occupancy, longer ones are read in place
    const int qstride = (max_len + 3) & ~3;
    const bool ldsq = qstride <= 2048;
    for (long long base = 0; base < n_reads; base += L.chunk) {
        const long long m = std::min<long long>(L.chunk, n_reads - base);
        A.n_reads = m; A.read_base = base;
        GBX_HIP(hipMemsetAsync(counters, 0, 8, s));                  // the read cursor
        const long long quads = std::min<long long>((m + 15) / 16 * 16, L.quads);
        {
            Stage st("fmi_smem", s);
            if (ldsq) hipLaunchKernelGGL(fmi_smem_kernel<true>, dim3((unsigned)(quads / 16)), dim3(64), (size_t)qstride * 16, s, A, qstride);
            else hipLaunchKernelGGL(fmi_smem_kernel<false>, dim3((unsigned)(quads / 16)), dim3(64), 0, s, A, 0);
        }
        const int nb = (int)((m + SCAN_BLOCK - 1) / SCAN_BLOCK);
        {
            Stage st("fmi_pack", s);
            hipLaunchKernelGGL(fmi_scan1_kernel, dim3(nb), dim3(SCAN_BLOCK), 0, s, A.raw_count, m, bsum);
            hipLaunchKernelGGL(fmi_scan2_kernel, dim3(1), dim3(1024), 0, s, bsum, nb, counters, d_n_out);
            hipLaunchKernelGGL(fmi_pack_kernel, dim3(nb), dim3(SCAN_BLOCK), 0, s, A.raw_count, m, base, bsum, counters, A.raw, A.raw_cap, d_out,
                               (long long)out_cap, d_smem_off);
        }
    }
    GBX_HIP(hipGetLastError());
    return GBX_OK;
}

int fmi_read_overflow(const void *d_work, int64_t *worst, hipStream_t s)
{
    unsigned long long v = 0;
    GBX_HIP(hipMemcpyAsync(&v, (const char *)d_work + 16, sizeof(v), hipMemcpyDeviceToHost, s));
    GBX_HIP(hipStreamSynchronize(s));
    *worst = (int64_t)v;
    return GBX_OK;
}

}  // namespace gbx
