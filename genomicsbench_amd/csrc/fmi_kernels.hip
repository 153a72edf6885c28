// fmi_kernels.hip — bwa-mem2 SMEM seeding on the FM-index for gfx950 (MI355X).
//
// Semantics: the three seeding rounds R/benchmarks/fmi/fmi.cpp:218-278 runs per batch of reads
// (FMI_search::getSMEMsAllPosOneThread, the re-seeding call of getSMEMsOnePosOneThread, bwtSeedStrategyAllPosOneThread,
// sortSMEMs), restated in oracle/fmi_oracle.c from bwa-mem2's published FMI_search.cpp; bit-exact on every field.
//
// What bounds it: one backwardExt is two dependent look-ups in the checkpoint table (rows k and k + s), each a
// random 64-byte line of a table far larger than the caches, and a read is a chain of ~800 of them.  Two limits were
// measured: instruction issue (the bookkeeping of a read is wave-uniform-per-read code that is paid per
// wavefront trip, however few lanes take a branch) and the rate at which a CU's vector memory path takes lane
// requests that do not share a line (one read per lane: a third of the instructions and yet slower at every occupancy).
//   * four lanes per read, lane b = base b: the device index stores a checkpoint as four {count, one-hot word}
//     pairs, so a look-up is ONE 16-byte load per lane and the quad's four loads are one 64-byte line: two line
//     requests per extension, the minimum.  16 reads per wavefront.
//   * SA rows and interval sizes in 32 bits when the reference has fewer than 2^32 rows, in 64 otherwise (two
//     instances): arithmetic, selects and cross-lane moves of 64-bit values are two instructions each.
//   * the rounds of a read are a state machine with a single extension site: every trip of the main loop each
//     read of the wavefront performs exactly one backwardExt (forward extension = backward extension of the reverse
//     complement), whatever round and phase it is in, so the wavefront's 32 line requests are always issued together;
//     the usual successor of an extension (next base of the sweep, next record of the backward pass) is decided
//     inline, the rarer transitions (close a forward sweep, next position, next round, next read) go through one
//     dispatch that is skipped when no quad needs it.
//   * the prev[] array of the backward sweep lives in a per-quad slab in HBM, one word per lane of the quad (k, l, s,
//     n); the next entry is requested together with the checkpoint look-ups of the current one and entry 0 is
//     forwarded in registers, so the sweep never waits for its own array.  Vector memory operations of one wavefront
//     are performed in order, so a quad reads back what it wrote without a fence.
//   * the read's bases sit in LDS, four bits each; reads are drawn from a cursor; SMEMs go to a fixed-capacity slot
//     per read, a second kernel sorts each read's few records by (m ascending, n descending) and packs them behind a
//     prefix sum of the counts.
#include <algorithm>
#include "gbx_internal.h"

namespace gbx {
namespace {

// SMEMs of one read before the pack pass: the LAST round reports up to one seed per minSeedLen + 1 bases and the first
// two rounds about as many again: 48 for 151-bp reads at minSeedLen 19 (8.6 on average, 18 at most on the bench's reads;
// the reference sizes its array at 20 per read), 4 per minSeedLen bases for long reads or short seeds
__host__ __device__ inline int raw_cap_for(int max_len, int min_seed_len)
{
    const int msl = min_seed_len > 1 ? min_seed_len : 1;
    const int c = (4 * max_len + msl - 1) / msl + 16;
    return c > 48 ? c : 48;
}
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
    uint2 *prev;                       // [resident quads][max_len + 1][4 lanes] words of 4 or 8 bytes: k, l, s, n of a backward-sweep record
    unsigned long long *counters;      // [0] read cursor, [1] extensions, [2] overflow flag, [3] running output total, [4] total before
                                       // this chunk, [5] longest read that exceeded max_read_len
};

__device__ inline long long u2ll(uint2 v) { return (long long)(((unsigned long long)v.y << 32) | v.x); }
__device__ inline uint2 ll2u(long long v) { return make_uint2((unsigned)v, (unsigned)((unsigned long long)v >> 32)); }

// quad_perm:[SEL,SEL,SEL,SEL]: the value of lane SEL of every quad in all four lanes of the quad
template <int SEL> __device__ inline unsigned quad_bcast(unsigned v)
{
    return (unsigned)__builtin_amdgcn_mov_dpp((int)v, SEL * 0x55, 0xf, 0xf, true);
}
template <int SEL> __device__ inline unsigned long long quad_bcast(unsigned long long v)
{
    const unsigned lo = quad_bcast<SEL>((unsigned)v), hi = quad_bcast<SEL>((unsigned)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}
__device__ inline unsigned shfl_iv(unsigned v, int src) { return (unsigned)__shfl((int)v, src); }
__device__ inline unsigned long long shfl_iv(unsigned long long v, int src)
{
    return ((unsigned long long)(unsigned)__shfl((int)(v >> 32), src) << 32) | (unsigned)__shfl((int)(unsigned)v, src);
}
template <class IV> __device__ inline IV pick4(int a, IV v0, IV v1, IV v2, IV v3) { return a == 0 ? v0 : a == 1 ? v1 : a == 2 ? v2 : v3; }

enum : int { ST_FWD = 0, ST_BWD = 1, ST_SEED = 2 };
// transitions between two extensions (T_EXT: the read needs one now; T_IDLE: no reads left for this quad)
enum : int { T_EXT, T_IDLE, T_FWD_END, T_BWD_JEND, T_BWD_END, T_ONEPOS_DONE, /* the rare ones: */ T_NEXT_READ, T_ONEPOS_INIT, T_P2_NEXT, T_SEED_INIT, T_READ_DONE };

typedef __attribute__((address_space(3))) unsigned lds_u32;

#ifndef GBX_FMI_WAVES
#define GBX_FMI_WAVES 6              // wavefronts per SIMD the register budget of the 64-bit instance is cut for (80 VGPRs)
#endif
#ifndef GBX_FMI_LDS_PREV
#define GBX_FMI_LDS_PREV 16          // records of a quad's prev[] array that live in LDS (32-bit instance; half as many in the 64-bit one)
#endif
#ifndef GBX_FMI_WAVES32
#define GBX_FMI_WAVES32 7            // ... and of the 32-bit instance (72 VGPRs; at 64 it spills: 134 ms instead of 107 for 3 M reads)
#endif
// IV: the type SA rows and interval sizes are held in - unsigned when the reference (both strands + sentinel) has
// fewer than 2^32 rows, else unsigned long long (a human genome: 6.2 G rows): the kernel is bound by instruction issue
// and 64-bit arithmetic, selects and cross-lane moves are two instructions each.
// LDSQ: the quad's read is staged in LDS, four bits per base, so that the base behind every decision is a ds_read_b32
// instead of a global load on the dependent chain; reads too long for that are read in place.
template <bool LDSQ, class IV>
__global__ void __launch_bounds__(64, sizeof(IV) == 4 ? GBX_FMI_WAVES32 : GBX_FMI_WAVES) fmi_smem_kernel(FmiArgs A, int qwords)
{
    extern __shared__ unsigned q_lds[];
    __shared__ IV cnt_lds[8];                         // count[] by a run-time base (a chain of selects becomes a table in scratch memory)
    const int lane = threadIdx.x, b = lane & 3;
    if (lane < 5) cnt_lds[lane] = (IV)A.count[lane];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_barrier();
    // prev[] of this quad: entry e = one IV word per lane (k, l, s, n) at (quad * (max_len + 1) + e) * 4 + b
    // its first EL records live in LDS (a forward sweep changes the interval size ~15 times for a 1-Gbase text, so the
    // backward passes mostly stay there: the array was 30 of the 105 HBM bytes per extension), the rest in the HBM slab
    IV *const prev = (IV *)A.prev + ((size_t)blockIdx.x * 16 + (lane >> 2)) * (size_t)(A.max_len + 1) * 4 + b;
    constexpr int EL = GBX_FMI_LDS_PREV * 4 / (int)sizeof(IV);
    unsigned *const ql = q_lds + (size_t)(lane >> 2) * (size_t)qwords;
    IV *const lprev = (IV *)(q_lds + (LDSQ ? 16 * (size_t)qwords : 0)) + (size_t)(lane >> 2) * (EL * 4) + b;
    auto prev_load = [&](int e) -> IV { return (EL > 0 && e < EL) ? lprev[e * 4] : prev[(size_t)e * 4]; };
    typedef __attribute__((address_space(3))) const IV lds_iv;        // typed LDS pointer: ds_read, not a FLAT load
    auto cnt_of = [&](int c) -> IV { return ((lds_iv *)cnt_lds)[c]; };
    const IV sentinel = (IV)A.sentinel;
    unsigned n_ext = 0;
    const int RAW_CAP = A.raw_cap;

    // quad-uniform state
    int t = T_NEXT_READ, state = ST_FWD;
    int rid_local = 0;                       // read of this chunk
    const uint8_t *q = nullptr;
    int len = 0, n_out = 0;
    int phase = 0;                           // 0: every start position, 1: re-seeding, 2: LAST round
    int x = 0, j = 0, next_x = 0, min_intv = 1;
    IV k = 0, l = 0, s = 0;                  // the match being extended forwards, [x, n]
    int n = 0;
    int num_prev = 0, vbase = 0, p = 0, num_curr = 0, curr_s = -1, m_cur = 0, a = 0;
    bool first = true;
    IV pk = 0, pl = 0, ps = 0;               // prev[p] of the backward sweep
    int pn = 0;
    // the backward sweep never waits for its own array: entry p + 1 is requested together with the checkpoint look-ups
    // of entry p (nxt, one word per lane), and entry 0 of a position - the last record of the forward sweep, later the
    // first record the previous position kept - is still in registers: (k, l, s, n), which the forward sweep no longer needs
    IV nxt = 0;
    int idx2 = 0, n1 = 0;
    GBX_GUARD(gd_trips, 0);                  // extensions the quad's current read may still take (set when the read is fetched)
    GBX_GUARD(gd_disp, 1 << 22);             // passes of the dispatch loop of one trip

    auto raw_of = [&]() -> uint2 * { return A.raw + (size_t)rid_local * (size_t)(RAW_CAP * 5); };
    auto base_at = [&](int i) -> int {
        if (LDSQ) return (int)((((const lds_u32 *)ql)[i >> 3] >> ((i & 7) << 2)) & 15u);
        return (int)q[i];
    };
    auto emit = [&](int m_, int n_, IV k_, IV l_, IV s_) {
        if (n_out < RAW_CAP) {
            uint2 *e = raw_of() + (size_t)n_out * 5;
            // the quad writes the record's five words: lanes 0..3 the first four, lane 0 the fifth
            const uint2 w = b == 0 ? make_uint2((unsigned)(A.read_base + rid_local), (unsigned)m_) : b == 1 ? make_uint2((unsigned)n_, 0u)
                          : b == 2 ? ll2u((long long)k_) : ll2u((long long)l_);
            e[b] = w;
            if (b == 0) e[4] = ll2u((long long)s_);
        }
        ++n_out;
    };
    auto take_prev = [&](IV w) {              // one word per lane -> the record in every lane
        pk = quad_bcast<0>(w); pl = quad_bcast<1>(w); ps = quad_bcast<2>(w); pn = (int)quad_bcast<3>((unsigned)w);
    };
    auto store_prev_arr = [&](int e, IV k_, IV l_, IV s_, int n_) {
        const IV v = b == 0 ? k_ : b == 1 ? l_ : b == 2 ? s_ : (IV)(unsigned)n_;
        if (EL > 0 && e < EL) lprev[e * 4] = v;
        else prev[(size_t)e * 4] = v;
    };
    auto seed_interval = [&](int c) { k = cnt_of(c); l = cnt_of(3 - c); s = cnt_of(c + 1) - cnt_of(c); };

    // steps that several transitions share (a transition is paid by the whole wavefront whenever one quad takes it, so the
    // usual successions are taken in one go instead of one trip through the dispatch each)
    // the next base of a forward sweep (getSMEMsOnePosOneThread's j loop head, the LAST round's alike): true = extend
    auto fwd_next = [&]() -> bool {
        if (j >= len) return false;
        a = base_at(j);
        next_x = j + 1;
        return a < 4;
    };
    // one start position (getSMEMsOnePosOneThread for (read, x, min_intv)) up to its first forward extension
    auto onepos_init = [&]() {
        a = base_at(x);
        next_x = x + 1;
        if (a >= 4) { t = T_ONEPOS_DONE; return; }
        seed_interval(a);
        n = x; num_prev = 0; j = x + 1;
        if (fwd_next()) { state = ST_FWD; t = T_EXT; } else t = T_FWD_END;
    };
    // the backward sweep moves on to position j: its first record is (k, l, s, n)
    auto bwd_enter = [&]() {
        if (j < 0) { t = T_BWD_END; return; }
        a = base_at(j);
        if (a > 3) { t = T_BWD_END; return; }
        num_curr = 0; curr_s = -1; p = 0; first = true;
        if (num_prev == 0) { t = T_BWD_JEND; return; }
        pk = k; pl = l; ps = s; pn = n;
        state = ST_BWD; t = T_EXT;
    };
    // the LAST round from start position x (bwtSeedStrategyAllPosOneThread): skips ambiguous bases
    auto seed_init = [&]() {
        for (;;) {
            if (x >= len) { t = T_READ_DONE; return; }
            a = base_at(x);
            next_x = x + 1;
            if (a >= 4) { x = next_x; continue; }
            seed_interval(a);
            n = x; j = x + 1;
            if (fwd_next()) { state = ST_SEED; t = T_EXT; return; }
            x = next_x;
        }
    };

    for (;;) {
        // ---- bookkeeping until this read needs an extension (or there is nothing left to do)
        if (__ballot(t != T_EXT) != 0) {
            // the three transitions of the backward sweep come up in three trips out of four (some quad of the sixteen is
            // at one), the others a few times per read: the frequent ones are tested directly, the rest behind one test
            do {
                if (t == T_BWD_JEND) {
                    num_prev = num_curr;
                    if (num_curr == 0) t = T_ONEPOS_DONE;
                    else { m_cur = j; --j; bwd_enter(); }
                } else if (t == T_FWD_END) {
                    if (s >= (IV)min_intv) {
                        store_prev_arr(num_prev, k, l, s, n); ++num_prev;          // ... and (k, l, s, n) is entry 0 of the backward view
                    } else if (num_prev > 0) {                                      // (a seed interval below min_intv: the last record pushed)
                        take_prev(prev_load(num_prev - 1));
                        k = pk; l = pl; s = ps; n = pn;
                    }
                    vbase = num_prev - 1; j = x - 1; m_cur = x;
                    bwd_enter();
                }
                if (t == T_BWD_END) {
                    if (num_prev != 0 && n - m_cur + 1 >= A.min_seed_len) emit(m_cur, n, k, l, s);      // view[0]
                    t = T_ONEPOS_DONE;
                }
                if (t == T_ONEPOS_DONE) {
                    if (phase == 0) {
                        x = next_x;                                   // getSMEMsAllPosOneThread: on to the next start position
                        if (x < len) onepos_init();
                        else { n1 = n_out < RAW_CAP ? n_out : RAW_CAP; idx2 = 0; phase = 1; t = T_P2_NEXT; }
                    } else t = T_P2_NEXT;
                }
                if (__ballot(t >= T_NEXT_READ) != 0) {
                    switch (t) {
                case T_NEXT_READ: {
                    unsigned long long r = 0;
                    if (b == 0) r = atomicAdd(&A.counters[0], 1ull);
                    r = quad_bcast<0>(r);
                    if ((long long)r >= A.n_reads) { t = T_IDLE; break; }
                    rid_local = (int)r;
                    len = A.read_len[A.read_base + r];
                    q = A.enc + A.read_off[A.read_base + r];
                    n_out = 0;
                    if (len > A.max_len) {                            // the caller's max_read_len is too small: no SMEMs, flagged
                        if (b == 0) atomicMax(&A.counters[5], (unsigned long long)len);
                        len = 0;
                    }
                    if (len <= 0) { t = T_READ_DONE; break; }
#ifdef GBX_LOOP_GUARD
                    gd_trips = 64ll * (len + 2) * (len + 2);           // the three rounds of a read take a few extensions per base
#endif
                    if (LDSQ) {
                        // eight bases per dword, the quad's lanes take turns; nothing is read behind the read (the last
                        // group base by base)
#pragma unroll 1
                        for (int i = b * 8; i < len; i += 32) {
                            unsigned w = 0;
                            if (i + 8 <= len) {
                                unsigned lo, hi;
                                __builtin_memcpy(&lo, q + i, 4);
                                __builtin_memcpy(&hi, q + i + 4, 4);
                                lo = (lo & 0x0f0f0f0fu); lo = (lo | (lo >> 4)) & 0x00ff00ffu; lo = (lo | (lo >> 8)) & 0xffffu;
                                hi = (hi & 0x0f0f0f0fu); hi = (hi | (hi >> 4)) & 0x00ff00ffu; hi = (hi | (hi >> 8)) & 0xffffu;
                                w = lo | (hi << 16);
                            } else {
                                for (int c = 0; i + c < len; ++c) w |= (unsigned)(q[i + c] & 15) << (c << 2);
                            }
                            ql[i >> 3] = w;
                        }
                    }
                    phase = 0; x = 0; min_intv = 1;
                    onepos_init();
                    break;
                }
                case T_ONEPOS_INIT:
                    onepos_init();
                    break;
                case T_P2_NEXT: {                                     // fmi.cpp:230-254: re-seed from the middle of long, rare SMEMs
                    if (idx2 >= n1) { phase = 2; x = 0; seed_init(); break; }
                    const uint2 *e = raw_of() + (size_t)idx2 * 5;
                    const uint2 w = e[b == 0 ? 0 : b == 1 ? 1 : 4];
                    const int m_ = (int)quad_bcast<0>(w.y), n_ = (int)quad_bcast<1>(w.x);
                    const long long s_ = (long long)quad_bcast<2>((unsigned long long)u2ll(w));
                    ++idx2;
                    const int start = m_, end = n_ + 1;
                    if (end - start < A.split_len || s_ > A.split_width) break;
                    x = (end + start) >> 1; min_intv = (int)(s_ + 1);
                    onepos_init();
                    break;
                }
                case T_SEED_INIT:
                    seed_init();
                    break;
                case T_READ_DONE:
                    if (b == 0) {
                        A.raw_count[rid_local] = n_out < RAW_CAP ? n_out : RAW_CAP;
                        if (n_out > RAW_CAP) atomicMax(&A.counters[2], (unsigned long long)n_out);
                    }
                    t = T_NEXT_READ;
                    break;
                default:
                    break;
                    }
                }
                if (GBX_GUARD_TRIP(gd_disp, GBX_GK_FMI, 2, A.read_base + rid_local)) { if (t != T_EXT) t = T_IDLE; }
            } while (__ballot(t != T_EXT && t != T_IDLE) != 0);
            if (__ballot(t == T_EXT) == 0) break;
        }

        // ---- one backwardExt per read (FMI_search.cpp): rows sp = k and ep = k + s of the checkpoint table; lane b of the
        // quad looks up base b: one 16-byte load per row and lane, one 64-byte line per row and quad
        if (t == T_EXT && GBX_GUARD_TRIP(gd_trips, GBX_GK_FMI, 1, A.read_base + rid_local)) t = T_IDLE;      // (the read is abandoned: the call fails)
        const bool act = t == T_EXT;
        IV ek = 0, el = 0, es = 0;
        int ea = 0;
        if (act) {
            if (state == ST_BWD) { ek = pk; el = pl; es = ps; ea = a; }
            else { ek = l; el = k; es = s; ea = 3 - a; }            // forwards = backwards on the reverse complement
        }
        const IV sp = ek, ep = ek + es;
        uint4 csp = A.index[(size_t)(sp >> 6) * 4 + b], cep = A.index[(size_t)(ep >> 6) * 4 + b];
        // one 16-byte request per row and lane also in the 32-bit instance, which has no use for the counts' upper words
        // (left alone the compiler splits the load into a dword and a dwordx2: twice the requests on the vector memory path)
        asm volatile("" : "+v"(csp.x), "+v"(csp.y), "+v"(csp.z), "+v"(csp.w));
        asm volatile("" : "+v"(cep.x), "+v"(cep.y), "+v"(cep.z), "+v"(cep.w));
        if (act && state == ST_BWD && p + 1 < num_prev) nxt = prev_load(vbase - (p + 1));
        const int ysp = (int)(sp & 63), yep = (int)(ep & 63);
        const unsigned long long msp = ysp ? ~0ull << (64 - ysp) : 0ull, mep = yep ? ~0ull << (64 - yep) : 0ull;
        auto occ = [](uint4 c, unsigned long long m) -> IV {
            const int pc = __builtin_popcountll((((unsigned long long)c.w << 32) | c.z) & m);
            if (sizeof(IV) == 4) return (IV)(c.x + (unsigned)pc);
            return (IV)((((unsigned long long)c.y << 32) | c.x) + (unsigned long long)pc);
        };
        const IV occ_sp = occ(csp, msp), occ_ep = occ(cep, mep);
        const IV kb = cnt_of(b) + occ_sp, sb = occ_ep - occ_sp;
        const IV s1 = quad_bcast<1>(sb), s2 = quad_bcast<2>(sb), s3 = quad_bcast<3>(sb), s0 = quad_bcast<0>(sb);
        const IV l3 = el + ((ek <= sentinel && ek + es > sentinel) ? 1 : 0);
        const IV l2 = l3 + s3, l1 = l2 + s2, l0 = l1 + s1;
        const IV rl = pick4<IV>(ea, l0, l1, l2, l3), rs = pick4<IV>(ea, s0, s1, s2, s3);
        const IV rk = shfl_iv(kb, (lane & ~3) | ea);
        if (act) {
            ++n_ext;
            if (state == ST_BWD) {
                const IV ns = rs;
                if (first && ns < (IV)min_intv && pn - m_cur + 1 >= A.min_seed_len) {
                    emit(m_cur, pn, pk, pl, ps);
                    first = false;
                } else if (ns >= (IV)min_intv && (long long)ns != (long long)curr_s) {
                    curr_s = (int)ns;
                    store_prev_arr(vbase - num_curr, rk, rl, ns, pn);
                    if (num_curr == 0) { k = rk; l = rl; s = ns; n = pn; }        // entry 0 of the next position's view
                    ++num_curr;
                    first = false;
                }
                ++p;
                if (p < num_prev) take_prev(nxt);                                 // stays T_EXT / ST_BWD
                else t = T_BWD_JEND;
            } else {
                // a forward step of round 1 / 2 (ST_FWD) or of the LAST round (ST_SEED): one code path, the rounds differ in
                // when they stop and in what they do then
                const bool fwd = state == ST_FWD;
                const IV ns = rs;
                if (fwd && ns != s) { store_prev_arr(num_prev, k, l, s, n); ++num_prev; }
                const bool stop = fwd ? ns < (IV)min_intv : (ns < (IV)A.max_intv && j - x + 1 >= A.min_seed_len + 1);
                if (!(fwd && stop)) { k = rl; l = rk; s = ns; n = j; }             // round 1 keeps the match it could not extend
                if (stop) {
                    if (fwd) { next_x = j; t = T_FWD_END; }
                    else {
                        if (s > 0) emit(x, n, k, l, s);
                        x = next_x;
                        t = T_SEED_INIT;
                    }
                } else {
                    ++j;
                    if (!fwd_next()) {                                           // end of the read or an ambiguous base
                        if (fwd) t = T_FWD_END;
                        else { x = next_x; t = T_SEED_INIT; }
                    }
                }
            }
        }
    }
    // extensions of the wavefront: one count per quad
    unsigned long long tot = b == 0 ? n_ext : 0;
    for (int d = 32; d; d >>= 1) tot += __shfl_xor(tot, d);
    if (lane == 0 && tot) atomicAdd(&A.counters[1], tot);
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

// A block takes SCAN_BLOCK reads: their offsets (one thread per read), then their records in (m ascending, n descending)
// order (sortSMEMs' compare_smem within one rid; records equal in (m, n) are equal in every field) - eight lanes per
// read, a lane per record: a record's rank is the number of records with a smaller key (ties by slot), found by passing
// the keys round the eight lanes; the lane then copies its 40 bytes to its place.
__global__ void __launch_bounds__(SCAN_BLOCK) fmi_pack_kernel(const int32_t *cnt, long long n, long long read_base, const long long *block_off,
                                                              const unsigned long long *counters, const uint2 *raw, int RAW_CAP,
                                                              gbx_fmi_smem *out, long long out_cap, int64_t *smem_off)
{
    __shared__ long long sh[SCAN_BLOCK / 64];
    __shared__ long long soff[SCAN_BLOCK];
    __shared__ int scnt[SCAN_BLOCK];
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
    soff[threadIdx.x] = off; scnt[threadIdx.x] = c;
    if (i < n) {
        smem_off[read_base + i] = off;
        if (i == n - 1) smem_off[read_base + n] = off + c;
    }
    __syncthreads();
    const int l8 = threadIdx.x & 7, grp = threadIdx.x >> 3;                    // 128 groups of eight lanes
    for (int r = grp; r < SCAN_BLOCK; r += SCAN_BLOCK / 8) {
        const long long rd = (long long)blockIdx.x * SCAN_BLOCK + r;
        const int cr = scnt[r];
        const long long o0 = soff[r];
        if (rd >= n || cr == 0 || o0 + cr > out_cap) continue;                  // (group-uniform)
        const uint2 *e = raw + (size_t)rd * (size_t)RAW_CAP * 5;
        for (int a0 = 0; a0 < cr; a0 += 8) {
            const int a = a0 + l8;
            const bool mine = a < cr;
            // key: m ascending, then n descending (reads have fewer than 65 536 bases)
            const unsigned ka = mine ? (e[a * 5].y << 16) | (0xffffu - e[a * 5 + 1].x) : 0xffffffffu;
            int rank = 0;
            for (int b0 = 0; b0 < cr; b0 += 8) {
                const int bb = b0 + l8;
                const unsigned kb = b0 == a0 ? ka : (bb < cr ? (e[bb * 5].y << 16) | (0xffffu - e[bb * 5 + 1].x) : 0xffffffffu);
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const unsigned ko = (unsigned)__shfl((int)kb, (threadIdx.x & ~7) | t, 64);
                    const int oidx = b0 + t;
                    rank += (oidx < cr) && (ko < ka || (ko == ka && oidx < a));
                }
            }
            if (mine) {
                uint2 *dst = (uint2 *)(out + o0 + rank);
#pragma unroll
                for (int w = 0; w < 5; ++w) dst[w] = e[a * 5 + w];
            }
        }
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

constexpr long long FMI_CHUNK = 12ll << 20;        // reads per launch: bounds the raw slots (12 Mi x 48 x 40 B = 24 GB of the 288; fewer for long
                                                   // reads).  Every launch ends in a tail of a few reads per wavefront: 10 M reads as one launch
                                                   // instead of three are 5 % faster
constexpr long long FMI_MAX_BLOCKS = 256ll * 4 * (GBX_FMI_WAVES32 > GBX_FMI_WAVES ? GBX_FMI_WAVES32 : GBX_FMI_WAVES);   // resident wavefronts on 256 CUs (more CUs: the grid stays this size)

struct FmiLayout { size_t o_raw, o_cnt, o_prev, o_bsum, total; long long chunk, blocks; };
static FmiLayout fmi_layout(int64_t n_reads, int32_t max_len, int RAW_CAP)
{
    FmiLayout L;
    long long cap_chunk = FMI_CHUNK * 48 / RAW_CAP;
    if (cap_chunk < 16) cap_chunk = 16;
    L.chunk = n_reads < cap_chunk ? (n_reads > 0 ? n_reads : 1) : cap_chunk;
    const long long want = (L.chunk + 15) / 16;
    L.blocks = want < FMI_MAX_BLOCKS ? want : FMI_MAX_BLOCKS;
    size_t at = 64;                                                        // counters: 8 x u64
    auto take = [&](size_t bytes) { const size_t o = at; at += (bytes + 255) & ~(size_t)255; return o; };
    L.o_raw = take((size_t)L.chunk * RAW_CAP * 40);
    L.o_cnt = take((size_t)L.chunk * 4);
    L.o_prev = take((size_t)L.blocks * 16 * (size_t)(max_len + 1) * 32);
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

// raw_cap = records a read may produce before the pack pass; 0: the default for these parameters (raw_cap_for)
size_t fmi_workspace_bytes(int64_t n_reads, int32_t max_len, int32_t min_seed_len, int raw_cap)
{
    return fmi_layout(n_reads, max_len, raw_cap > 0 ? raw_cap : raw_cap_for(max_len, min_seed_len)).total;
}

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
               int64_t *d_smem_off, int64_t *d_n_out, void *d_work, size_t work_bytes, hipStream_t s, int raw_cap)
{
    if (raw_cap <= 0) raw_cap = raw_cap_for(max_len, p->min_seed_len);
    if (max_len < 0 || max_len > 65535) { set_error("fmi: reads of up to 65535 bases (the reference asserts 10000, fmi.cpp:93)"); return GBX_ERR_UNSUPPORTED; }
    if (idx->ref_seq_len < 2 || idx->ref_seq_len >= (1ll << 40)) { set_error("fmi: bad reference length"); return GBX_ERR_ARG; }
    const FmiLayout L = fmi_layout(n_reads, max_len, raw_cap);
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
    A.enc = d_enc; A.read_off = d_read_off; A.read_len = d_read_len; A.max_len = max_len; A.raw_cap = raw_cap;
    A.raw = (uint2 *)(wb + L.o_raw); A.raw_count = (int32_t *)(wb + L.o_cnt); A.prev = (uint2 *)(wb + L.o_prev);
    A.counters = counters;
    long long *bsum = (long long *)(wb + L.o_bsum);
    // the read of a quad is staged in LDS, four bits per base (an odd dword count per quad: the copies start in different
    // banks): 16 x 19 dwords per wavefront for 151-bp reads; reads of more than ~8000 bases are read in place
    const int qwords = ((max_len + 7) / 8) | 1;
    static const size_t pad = getenv("GBX_FMI_LDS_PAD") ? (size_t)atol(getenv("GBX_FMI_LDS_PAD")) : 0;
    // the whole request must stay inside the 64 KB a workgroup gets without raising the function's limit: the staged reads,
    // the prev[] slab, the tuning pad and the kernel's static LDS (cnt_lds: 64 B)
    const size_t lds_fixed = (size_t)GBX_FMI_LDS_PREV * 16 * 16 + pad + 64;
    const char *ipenv = getenv("GBX_FMI_INPLACE");                    // test aid: 1 = read the bases in place whatever their length
    const bool ldsq = (size_t)qwords * 64 + lds_fixed <= 65536 && !(ipenv && atoi(ipenv));
    const bool wide = idx->ref_seq_len >= (1ll << 32);                // SA rows and interval sizes need 64 bits
    // GBX_FMI_LDS_PAD (tuning aid): extra dynamic LDS per wavefront, i.e. fewer wavefronts per CU; GBX_FMI_WIDE=1 (test aid):
    // the 64-bit instance whatever the reference length
    const char *wenv = getenv("GBX_FMI_WIDE");
    const bool w64 = wide || (wenv && atoi(wenv));
    for (long long base = 0; base < n_reads; base += L.chunk) {
        const long long m = std::min<long long>(L.chunk, n_reads - base);
        A.n_reads = m; A.read_base = base;
        GBX_HIP(hipMemsetAsync(counters, 0, 8, s));                  // the read cursor
        const long long blocks = std::min<long long>((m + 15) / 16, L.blocks);
        {
            Stage st("fmi_smem", s);
            const dim3 g((unsigned)blocks), tb(64);
            const size_t lds = (ldsq ? (size_t)qwords * 64 : 0) + (size_t)GBX_FMI_LDS_PREV * 16 * 16 + pad;
            if (ldsq && !w64) hipLaunchKernelGGL((fmi_smem_kernel<true, unsigned>), g, tb, lds, s, A, qwords);
            else if (ldsq) hipLaunchKernelGGL((fmi_smem_kernel<true, unsigned long long>), g, tb, lds, s, A, qwords);
            else if (!w64) hipLaunchKernelGGL((fmi_smem_kernel<false, unsigned>), g, tb, lds, s, A, 0);
            else hipLaunchKernelGGL((fmi_smem_kernel<false, unsigned long long>), g, tb, lds, s, A, 0);
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
    GBX_GUARD_CHECK("fmi");
    return GBX_OK;
}

int fmi_read_overflow(const void *d_work, int64_t *worst, hipStream_t s)
{
    unsigned long long v[6] = {0, 0, 0, 0, 0, 0};
    GBX_HIP(hipMemcpyAsync(v, d_work, sizeof(v), hipMemcpyDeviceToHost, s));
    GBX_HIP(hipStreamSynchronize(s));
    *worst = (int64_t)v[2];
    if (v[5]) { set_error("fmi: a read of %llu bases exceeds max_read_len (it got no SMEMs)", v[5]); return GBX_ERR_ARG; }
    return GBX_OK;
}

}  // namespace gbx
