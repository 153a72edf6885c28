// capi_poa.hip — poa entries of the C-ABI (include/gbx.h).
#include "capi_common.h"

using namespace gbx;

extern "C" {

/* --------------------------------------------------------------------- poa */
void gbx_poa_default_params(gbx_poa_params *p)
{
    memset(p, 0, sizeof(*p));
    p->m = 2; p->n = -4; p->g = -6; p->e = -2; p->q = -25; p->c = -1;   /* msa_spoa_omp.cpp:156-162,184 */
}

int gbx_poa_plan_host(int64_t n_windows, const int64_t *win_first_seq, const int32_t *seq_len, gbx_poa_plan *plan)
{
    if (n_windows < 0 || !plan || (n_windows > 0 && (!win_first_seq || !seq_len))) {
        set_error("gbx_poa_plan_host: bad argument");
        return GBX_ERR_ARG;
    }
    int lmax = 1, smax = 1;
    int64_t bmax = 1, n_long = 0;
    for (int64_t w = 0; w < n_windows; ++w) {
        const int64_t a = win_first_seq[w], b = win_first_seq[w + 1];
        if (b < a) { set_error("gbx_poa_plan_host: win_first_seq not monotone at window %lld", (long long)w); return GBX_ERR_ARG; }
        if (b - a > GBX_POA_MAX_SEQS_PER_WINDOW) {
            set_error("gbx_poa_plan_host: window %lld has more than %d sequences", (long long)w, GBX_POA_MAX_SEQS_PER_WINDOW);
            return GBX_ERR_UNSUPPORTED;
        }
        int64_t bases = 0;
        bool is_long = false;
        for (int64_t s = a; s < b; ++s) {
            if (seq_len[s] < 0) { set_error("gbx_poa_plan_host: negative sequence length"); return GBX_ERR_ARG; }
            if (seq_len[s] > lmax) lmax = seq_len[s];
            if (seq_len[s] > POA_PIPE_MAXLEN) is_long = true;
            bases += seq_len[s];
        }
        if (is_long) ++n_long;
        if (b - a > smax) smax = (int)(b - a);
        if (bases > bmax) bmax = bases;
    }
    if (n_long > 0x7fffffff) { set_error("gbx_poa_plan_host: too many windows"); return GBX_ERR_UNSUPPORTED; }
    plan->max_seq_len = lmax;
    plan->max_seqs_per_window = smax < 4 ? 4 : ((smax + 3) & ~3);     /* multiple of 4: 16-byte aligned edge rows */
    int nf = 6;                                            /* typical windows stay below ~3.6x the read length */
    if (const char *e = getenv("GBX_POA_NODE_FACTOR")) { const int v = atoi(e); if (v >= 2 && v <= 16) nf = v; }   /* tuning aid */
    int64_t cap = (int64_t)nf * lmax + 256;
    if (bmax + 8 < cap) cap = bmax + 8;
    plan->node_cap = (int32_t)cap;
    int cus = 256, dev = 0;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    else (void)hipGetLastError();
    const int64_t resident = (int64_t)cus * poa_waves_per_cu(plan->node_cap);   /* one window per resident wavefront */
    const int64_t n_main = n_windows - n_long;
    plan->n_slots = (int32_t)(n_main < resident ? n_main : resident);
    // more windows than resident wavefronts: the lock-step form (poa_kernels.hip), a slot per window, while that fits the
    // budget (GBX_POA_LOCKSTEP_MAX_GB, default 96 of the 288 GB: 'large' takes 55)
    if (poa_lockstep_wanted(n_main, resident)) {
        const char *e = getenv("GBX_POA_LOCKSTEP_MAX_GB");
        const double budget = (e && atof(e) > 0 ? atof(e) : 96.0) * 1e9;
        if ((double)poa_slot_bytes(plan->node_cap, plan->max_seqs_per_window, plan->max_seq_len, false) * (double)n_main <= budget)
            plan->n_slots = (int32_t)n_main;
    }
    plan->n_long_windows = (int32_t)n_long;
    plan->long_slots = (int32_t)(n_long < resident ? n_long : resident);
    plan->n_windows = n_windows;
    return GBX_OK;
}

size_t gbx_poa_workspace_bytes(const gbx_poa_plan *plan)
{
    if (!plan) return 0;
    return poa_workspace_bytes(plan);
}

int gbx_poa_cells(const gbx_poa_plan *plan, const void *d_work, int64_t *cells, void *stream)
{
    if (!plan || !d_work || !cells) { set_error("gbx_poa_cells: null pointer"); return GBX_ERR_ARG; }
    return poa_read_cells(d_work, poa_slot_bytes(plan->node_cap, plan->max_seqs_per_window, plan->max_seq_len, false) * (size_t)(plan->n_slots > 0 ? plan->n_slots : 0),
                          cells, (hipStream_t)stream);
}

/* development aid (scripts/dbg_poa_phases.py): byte offset of the 32-counter block inside a poa workspace */
size_t gbx_debug_poa_counter_offset(const gbx_poa_plan *plan)
{
    return poa_slot_bytes(plan->node_cap, plan->max_seqs_per_window, plan->max_seq_len, false) * (size_t)(plan->n_slots > 0 ? plan->n_slots : 0);
}

int gbx_poa_consensus_device(const gbx_poa_params *p, const gbx_poa_plan *plan, int64_t n_windows,
                             const int64_t *d_win_first_seq, const int64_t *d_seq_off, const int32_t *d_seq_len,
                             const char *d_arena, char *d_cons, int32_t *d_cons_len, int32_t *d_status,
                             int64_t cons_stride, void *d_work, size_t work_bytes, void *stream)
{
    if (!p || !plan || n_windows < 0 || cons_stride <= 0) { set_error("gbx_poa_consensus_device: bad argument"); return GBX_ERR_ARG; }
    if (n_windows == 0) return GBX_OK;
    if (!d_win_first_seq || !d_seq_off || !d_seq_len || !d_arena || !d_cons || !d_cons_len || !d_status || !d_work) {
        set_error("gbx_poa_consensus_device: null pointer");
        return GBX_ERR_ARG;
    }
    int rc = require_device();
    if (rc) return rc;
    return poa_launch(p, plan, n_windows, d_win_first_seq, d_seq_off, d_seq_len, (const uint8_t *)d_arena, (uint8_t *)d_cons, d_cons_len,
                      d_status, cons_stride, d_work, work_bytes, (hipStream_t)stream);
}

// The int16 paths (poa_launch: window kernel, team kernel, long-window launch) over the windows of one validated job on the
// calling thread's device; status[w] = GBX_POA_ST_* bits of window w afterwards (0: its consensus is in place).  Windows whose
// graph outgrew the plan's node capacity have been run again with the largest capacity int16 cells admit; what still carries
// GBX_POA_ST_NODES then belongs to the wide path (poa_host_wide).
static int poa_host_narrow(const gbx_poa_params *p, int64_t n_windows, const int64_t *win_first_seq,
                           int64_t n_seqs, const int64_t *seq_off, const int32_t *seq_len,
                           const char *arena, int64_t arena_bytes,
                           char *cons, int32_t *cons_len, int64_t cons_stride, std::vector<int32_t> &status)
{
    int rc;
    gbx_poa_plan plan;
    if ((rc = gbx_poa_plan_host(n_windows, win_first_seq, seq_len, &plan))) return rc;
    const size_t wb = gbx_poa_workspace_bytes(&plan);
    const bool trace = getenv("GBX_HOST_TRACE") != nullptr;
    const double t_begin = wall_s();
    auto mark = [&](const char *what) { if (trace) fprintf(stderr, "[gbx poa host] %9.3f ms %s\n", (wall_s() - t_begin) * 1e3, what); };
    HostLane lane;
    if ((rc = lane.acquire())) return rc;
    Lane *L = lane.l;
    DevBuf dwf(L), doff(L), dlen(L), dar(L), dcons(L), dcl(L), dst(L), dw(L);
    if ((rc = dwf.alloc((n_windows + 1) * 8)) || (rc = doff.alloc(n_seqs * 8)) || (rc = dlen.alloc(n_seqs * 4)) ||
        (rc = dar.alloc(arena_bytes)) || (rc = dcons.alloc(n_windows * cons_stride)) || (rc = dcl.alloc(n_windows * 4)) ||
        (rc = dst.alloc(n_windows * 4)) || (rc = dw.alloc(wb)))
        return rc;
    mark("allocated");
    status.assign((size_t)n_windows, 0);
    {
        HostPipe pipe(lane.l, (size_t)arena_bytes + (size_t)n_seqs * 12 + (size_t)n_windows * 8, false);
        if ((rc = pipe.prepare(1))) return rc;
        pipe.stage(0, dwf.p, win_first_seq, (n_windows + 1) * 8);
        pipe.stage(0, doff.p, seq_off, n_seqs * 8);
        pipe.stage(0, dlen.p, seq_len, n_seqs * 4);
        pipe.stage(0, dar.p, arena, arena_bytes);
        pipe.start();
        if ((rc = pipe.wait_stage(0))) return pipe.finish(rc);
        rc = poa_launch(p, &plan, n_windows, dwf.as<int64_t>(), doff.as<int64_t>(), dlen.as<int32_t>(), dar.as<uint8_t>(),
                        dcons.as<uint8_t>(), dcl.as<int32_t>(), dst.as<int32_t>(), cons_stride, dw.p, wb, lane.l->compute);
            if (rc) return pipe.finish(rc);
        pipe.fetch(0, cons, dcons.p, n_windows * cons_stride);
        pipe.fetch(0, cons_len, dcl.p, n_windows * 4);
        pipe.fetch(0, status.data(), dst.p, n_windows * 4);
        if ((rc = pipe.chunk_launched(0))) return pipe.finish(rc);
        mark("kernels queued");
        if ((rc = pipe.finish())) return rc;
        mark("results fetched");
    }
    // Windows whose graph outgrew the first pass's node capacity (deep or noisy windows; the plan sizes it for the
    // typical case so that the 'large' job's slots stay small) run again with room for the worst case of exactly
    // those windows: every base its own node, bounded by what int16 scores admit.  spoa has no such limit
    // (msa_spoa_omp.cpp:237-252), so only a window that cannot be represented at all fails the call.
    std::vector<int64_t> redo;
    for (int64_t w = 0; w < n_windows; ++w)
        if (status[w] & GBX_POA_ST_NODES) redo.push_back(w);      // (other bits set next to it are re-decided by the second pass)
    if (!redo.empty()) {
        std::vector<int64_t> wf(redo.size() + 1, 0), off;
        std::vector<int32_t> len;
        int64_t bmax = 1, n_long = 0;
        int lmax = 1, smax = 1;
        for (size_t k = 0; k < redo.size(); ++k) {
            const int64_t a = win_first_seq[redo[k]], b = win_first_seq[redo[k] + 1];
            int64_t bases = 0;
            bool is_long = false;
            for (int64_t sidx = a; sidx < b; ++sidx) {
                off.push_back(seq_off[sidx]); len.push_back(seq_len[sidx]);
                bases += seq_len[sidx];
                if (seq_len[sidx] > lmax) lmax = seq_len[sidx];
                if (seq_len[sidx] > POA_PIPE_MAXLEN) is_long = true;
            }
            if (is_long) ++n_long;
            if (b - a > smax) smax = (int)(b - a);
            if (bases > bmax) bmax = bases;
            wf[k + 1] = (int64_t)off.size();
        }
        int64_t cap = bmax + 8;
        while (cap > plan.node_cap && !poa_scores_fit_int16(p, cap, lmax)) cap -= (cap - plan.node_cap + 1) / 2;
        if (cap > plan.node_cap) {
            const int64_t nr = (int64_t)redo.size(), ns = (int64_t)off.size();
            gbx_poa_plan big = plan;
            big.max_seq_len = lmax;
            big.max_seqs_per_window = smax < 4 ? 4 : ((smax + 3) & ~3);
            big.node_cap = (int32_t)cap;
            big.n_windows = nr;
            big.n_long_windows = (int32_t)n_long;
            // slots: what the device has room for beside the first pass's buffers (still held), at most 16 GB, at most one per window
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); free_b = (size_t)16 << 30; }
            size_t budget = free_b / 2 < ((size_t)16 << 30) ? free_b / 2 : ((size_t)16 << 30);
            DevBuf dwf2(L), doff2(L), dlen2(L), dcons2(L), dcl2(L), dst2(L), dw2(L);
            if ((rc = dwf2.alloc((nr + 1) * 8)) || (rc = doff2.alloc(ns * 8)) || (rc = dlen2.alloc(ns * 4)) ||
                (rc = dcons2.alloc(nr * cons_stride)) || (rc = dcl2.alloc(nr * 4)) || (rc = dst2.alloc(nr * 4)))
                return rc;
            size_t wb2 = 0;
            for (;;) {                                      // fewer slots when the allocation fails
                const size_t ms = poa_slot_bytes(big.node_cap, big.max_seqs_per_window, big.max_seq_len, false);
                const size_t ls = poa_slot_bytes(big.node_cap, big.max_seqs_per_window, big.max_seq_len, true);
                int64_t slots = (int64_t)(budget / (ls ? ls : 1));
                if (slots < 1) slots = 1;
                if (slots > plan.n_slots + plan.long_slots && plan.n_slots + plan.long_slots > 0) slots = plan.n_slots + plan.long_slots;
                const int64_t n_main = nr - n_long;
                big.n_slots = (int32_t)(n_main < slots ? n_main : slots);
                big.long_slots = (int32_t)(n_long < slots ? n_long : slots);
                (void)ms;
                wb2 = gbx_poa_workspace_bytes(&big);
                if (dw2.alloc(wb2) == GBX_OK) break;
                if (budget <= ls) { set_error("gbx_poa_consensus_host: no device memory for the second pass of %lld oversized window(s)", (long long)nr); return GBX_ERR_NOMEM; }
                budget /= 2;
            }
            hipStream_t st = lane.l->compute;
            GBX_HIP(hipMemcpyAsync(dwf2.p, wf.data(), (size_t)(nr + 1) * 8, hipMemcpyHostToDevice, st));
            GBX_HIP(hipMemcpyAsync(doff2.p, off.data(), (size_t)ns * 8, hipMemcpyHostToDevice, st));
            GBX_HIP(hipMemcpyAsync(dlen2.p, len.data(), (size_t)ns * 4, hipMemcpyHostToDevice, st));
            if ((rc = poa_launch(p, &big, nr, dwf2.as<int64_t>(), doff2.as<int64_t>(), dlen2.as<int32_t>(), dar.as<uint8_t>(),
                                 dcons2.as<uint8_t>(), dcl2.as<int32_t>(), dst2.as<int32_t>(), cons_stride, dw2.p, wb2, st)))
                return rc;
            std::vector<char> c2((size_t)nr * (size_t)cons_stride);
            std::vector<int32_t> l2((size_t)nr), s2((size_t)nr);
            GBX_HIP(hipMemcpyAsync(c2.data(), dcons2.p, c2.size(), hipMemcpyDeviceToHost, st));
            GBX_HIP(hipMemcpyAsync(l2.data(), dcl2.p, (size_t)nr * 4, hipMemcpyDeviceToHost, st));
            GBX_HIP(hipMemcpyAsync(s2.data(), dst2.p, (size_t)nr * 4, hipMemcpyDeviceToHost, st));
            GBX_HIP(hipStreamSynchronize(st));
            for (int64_t k = 0; k < nr; ++k) {
                status[(size_t)redo[(size_t)k]] = s2[(size_t)k];
                if (s2[(size_t)k]) continue;
                cons_len[redo[(size_t)k]] = l2[(size_t)k];
                memcpy(cons + redo[(size_t)k] * cons_stride, c2.data() + (size_t)k * (size_t)cons_stride, (size_t)cons_stride);
            }
            mark("oversized windows redone");
        }
    }
    return GBX_OK;
}

// The wide path (poa_launch_wide: int32 cells) over the windows `list` of a validated job: windows whose scores may leave the
// int16 range (long reads) and windows whose graph outgrew every capacity int16 admits - where spoa switches to 32-bit lanes.
// First with the plan's usual capacity (six times the longest read), then, for a graph that outgrew that too, with a node
// per base, which cannot overflow.  Slots: what the device has room for, at most one per window.
static int poa_host_wide(const gbx_poa_params *p, const std::vector<int64_t> &list, const int64_t *win_first_seq,
                         const int64_t *seq_off, const int32_t *seq_len, const char *arena, int64_t arena_bytes,
                         char *cons, int32_t *cons_len, int64_t cons_stride, std::vector<int32_t> &status)
{
    if (list.empty()) return GBX_OK;
    int rc;
    HostLane lane;
    if ((rc = lane.acquire())) return rc;
    Lane *L = lane.l;
    hipStream_t st = L->compute;
    DevBuf dar(L);
    if ((rc = dar.alloc((size_t)arena_bytes + 64))) return rc;
    GBX_HIP(hipMemcpyAsync(dar.p, arena, (size_t)arena_bytes, hipMemcpyHostToDevice, st));
    std::vector<int64_t> todo = list;
    for (int pass = 0; pass < 2 && !todo.empty(); ++pass) {
        const int64_t nr = (int64_t)todo.size();
        std::vector<int64_t> wf((size_t)nr + 1, 0), off;
        std::vector<int32_t> len;
        int64_t bmax = 1;
        int lmax = 1, smax = 1;
        for (int64_t k = 0; k < nr; ++k) {
            const int64_t a = win_first_seq[todo[(size_t)k]], b = win_first_seq[todo[(size_t)k] + 1];
            int64_t bases = 0;
            for (int64_t sidx = a; sidx < b; ++sidx) {
                off.push_back(seq_off[sidx]); len.push_back(seq_len[sidx]);
                bases += seq_len[sidx];
                if (seq_len[sidx] > lmax) lmax = seq_len[sidx];
            }
            if (b - a > smax) smax = (int)(b - a);
            if (bases > bmax) bmax = bases;
            wf[(size_t)k + 1] = (int64_t)off.size();
        }
        int64_t cap = pass == 0 ? (int64_t)6 * lmax + 256 : bmax + 8;
        if (bmax + 8 < cap) cap = bmax + 8;
        if (cap > 0x7fffff00LL) { set_error("gbx_poa_consensus_host: a window of %lld bases is beyond the device path", (long long)bmax); return GBX_ERR_UNSUPPORTED; }
        const int deg = smax < 4 ? 4 : ((smax + 3) & ~3);
        const int64_t ns = (int64_t)off.size();
        const size_t slot = poa_wide_slot_bytes((int)cap, deg, lmax);
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); free_b = (size_t)64 << 30; }
        int cus = 256, dev = 0;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        int64_t slots = (int64_t)((double)free_b * 0.8 / (double)slot);
        if (slots > nr) slots = nr;
        if (slots > (int64_t)cus * 4) slots = (int64_t)cus * 4;
        if (const char *e = getenv("GBX_POA_WIDE_SLOTS")) { const long long v = atoll(e); if (v >= 1 && v < slots) slots = v; }      /* test aid */
        if (slots < 1) {
            set_error("gbx_poa_consensus_host: a window of %lld nodes x %d columns needs %.1f GB of 32-bit DP planes, the device has %.1f GB free",
                      (long long)cap, lmax, (double)slot / 1e9, (double)free_b / 1e9);
            return GBX_ERR_NOMEM;
        }
        DevBuf dwf(L), doff(L), dlen(L), dcons(L), dcl(L), dst(L), dw(L);
        size_t wb = 0;
        for (;;) {                                          // fewer slots when the allocation fails
            wb = poa_wide_workspace_bytes((int)cap, deg, lmax, (int)slots);
            if (dw.alloc(wb) == GBX_OK) break;
            if (slots == 1) { set_error("gbx_poa_consensus_host: no device memory for the 32-bit DP planes of a window (%.1f GB)", (double)slot / 1e9); return GBX_ERR_NOMEM; }
            slots = (slots + 1) / 2;
        }
        if ((rc = dwf.alloc((size_t)(nr + 1) * 8)) || (rc = doff.alloc((size_t)ns * 8)) || (rc = dlen.alloc((size_t)ns * 4)) ||
            (rc = dcons.alloc((size_t)nr * (size_t)cons_stride)) || (rc = dcl.alloc((size_t)nr * 4)) || (rc = dst.alloc((size_t)nr * 4)))
            return rc;
        std::vector<char> c2((size_t)nr * (size_t)cons_stride);
        std::vector<int32_t> l2((size_t)nr), s2((size_t)nr);
        struct SyncOnExit { hipStream_t s; ~SyncOnExit() { (void)hipStreamSynchronize(s); } } sync_on_exit{st};      // whatever path leaves this pass
        GBX_HIP(hipMemcpyAsync(dwf.p, wf.data(), (size_t)(nr + 1) * 8, hipMemcpyHostToDevice, st));
        GBX_HIP(hipMemcpyAsync(doff.p, off.data(), (size_t)ns * 8, hipMemcpyHostToDevice, st));
        GBX_HIP(hipMemcpyAsync(dlen.p, len.data(), (size_t)ns * 4, hipMemcpyHostToDevice, st));
        if ((rc = poa_launch_wide(p, nr, dwf.as<int64_t>(), doff.as<int64_t>(), dlen.as<int32_t>(), dar.as<uint8_t>(), dcons.as<uint8_t>(),
                                  dcl.as<int32_t>(), dst.as<int32_t>(), cons_stride, (int)cap, deg, lmax, (int)slots, dw.p, wb, st))) {
            (void)hipStreamSynchronize(st);                  // (the index arrays above are this scope's: nothing may still be reading them)
            return rc;
        }
        GBX_HIP(hipMemcpyAsync(c2.data(), dcons.p, c2.size(), hipMemcpyDeviceToHost, st));
        GBX_HIP(hipMemcpyAsync(l2.data(), dcl.p, (size_t)nr * 4, hipMemcpyDeviceToHost, st));
        GBX_HIP(hipMemcpyAsync(s2.data(), dst.p, (size_t)nr * 4, hipMemcpyDeviceToHost, st));
        GBX_HIP(hipStreamSynchronize(st));
        std::vector<int64_t> again;
        for (int64_t k = 0; k < nr; ++k) {
            const int64_t w = todo[(size_t)k];
            status[(size_t)w] = s2[(size_t)k];
            if (s2[(size_t)k] & GBX_POA_ST_NODES) { again.push_back(w); continue; }
            if (s2[(size_t)k]) continue;
            cons_len[w] = l2[(size_t)k];
            memcpy(cons + w * cons_stride, c2.data() + (size_t)k * (size_t)cons_stride, (size_t)cons_stride);
        }
        todo.swap(again);
    }
    return GBX_OK;
}

// One device (the calling thread's current one).  win_base / seq_base = indices of window 0 / sequence 0 in the caller's job
// (error texts only).
static int poa_host_one(const gbx_poa_params *p, int64_t n_windows, const int64_t *win_first_seq,
                        int64_t n_seqs, const int64_t *seq_off, const int32_t *seq_len,
                        const char *arena, int64_t arena_bytes,
                        char *cons, int32_t *cons_len, int64_t cons_stride, int64_t win_base = 0, int64_t seq_base = 0)
{
    RoctxRange range_("gbx_poa_consensus_host");
    if (!p || n_windows < 0 || n_seqs < 0 || arena_bytes < 0 || cons_stride <= 0) {
        set_error("gbx_poa_consensus_host: bad argument");
        return GBX_ERR_ARG;
    }
    if (n_windows == 0) return GBX_OK;
    if (!win_first_seq || !seq_off || !seq_len || !arena || !cons || !cons_len) {
        set_error("gbx_poa_consensus_host: null pointer");
        return GBX_ERR_ARG;
    }
    if (win_first_seq[0] != 0 || win_first_seq[n_windows] != n_seqs) {
        set_error("gbx_poa_consensus_host: win_first_seq must span [0, n_seqs]");
        return GBX_ERR_ARG;
    }
    for (int64_t s = 0; s < n_seqs; ++s)
        if (seq_len[s] < 0 || seq_off[s] < 0 || seq_off[s] + seq_len[s] > arena_bytes) {
            set_error("gbx_poa_consensus_host: sequence %lld lies outside the arena", (long long)(seq_base + s));
            return GBX_ERR_ARG;
        }
    for (int64_t w = 0; w < n_windows; ++w)
        if (win_first_seq[w + 1] < win_first_seq[w]) { set_error("gbx_poa_consensus_host: win_first_seq not monotone at window %lld", (long long)(win_base + w)); return GBX_ERR_ARG; }
    int rc = require_device();
    if (rc) return rc;
    // Which windows int16 cells can hold: the plan of the job (longest sequence, node capacity) must pass poa_scores_fit_int16;
    // while it does not, the windows with the longest sequences leave for the wide path (int32 cells), as spoa's engine
    // switches to 32-bit lanes for them.  GBX_POA_FORCE_WIDE=1 (a test aid) sends everything there.
    std::vector<int> wl((size_t)n_windows, 0);
    std::vector<int64_t> wbases((size_t)n_windows, 0);
    for (int64_t w = 0; w < n_windows; ++w)
        for (int64_t sidx = win_first_seq[w]; sidx < win_first_seq[w + 1]; ++sidx) {
            wl[(size_t)w] = seq_len[sidx] > wl[(size_t)w] ? seq_len[sidx] : wl[(size_t)w];
            wbases[(size_t)w] += seq_len[sidx];
        }
    std::vector<char> is_wide((size_t)n_windows, 0);
    int64_t n_wide = 0;
    if (getenv("GBX_POA_FORCE_WIDE") && atoi(getenv("GBX_POA_FORCE_WIDE")) != 0) { is_wide.assign((size_t)n_windows, 1); n_wide = n_windows; }
    while (n_wide < n_windows) {
        int lmax = 1;
        int64_t bmax = 1;
        for (int64_t w = 0; w < n_windows; ++w)
            if (!is_wide[(size_t)w]) { lmax = wl[(size_t)w] > lmax ? wl[(size_t)w] : lmax; bmax = wbases[(size_t)w] > bmax ? wbases[(size_t)w] : bmax; }
        int64_t cap = (int64_t)6 * lmax + 256;             // (gbx_poa_plan_host's rule)
        if (bmax + 8 < cap) cap = bmax + 8;
        if (poa_scores_fit_int16(p, cap, lmax)) break;
        for (int64_t w = 0; w < n_windows; ++w)
            if (!is_wide[(size_t)w] && wl[(size_t)w] == lmax) { is_wide[(size_t)w] = 1; ++n_wide; }
    }
    std::vector<int32_t> status((size_t)n_windows, 0);
    std::vector<int64_t> wide;
    if (n_wide == 0) {
        if ((rc = poa_host_narrow(p, n_windows, win_first_seq, n_seqs, seq_off, seq_len, arena, arena_bytes, cons, cons_len, cons_stride, status))) return rc;
    } else {
        // the windows that stay: a job of their own over the same arena, results handed back to the caller's rows
        std::vector<int64_t> keep, wf(1, 0), off;
        std::vector<int32_t> len;
        for (int64_t w = 0; w < n_windows; ++w) {
            if (is_wide[(size_t)w]) { wide.push_back(w); continue; }
            keep.push_back(w);
            for (int64_t sidx = win_first_seq[w]; sidx < win_first_seq[w + 1]; ++sidx) { off.push_back(seq_off[sidx]); len.push_back(seq_len[sidx]); }
            wf.push_back((int64_t)off.size());
        }
        if (!keep.empty()) {
            const int64_t nk = (int64_t)keep.size();
            std::vector<char> c2((size_t)nk * (size_t)cons_stride);
            std::vector<int32_t> l2((size_t)nk, 0), s2;
            if (off.empty()) { off.push_back(0); len.push_back(0); }      // (arrays must not be null)
            if ((rc = poa_host_narrow(p, nk, wf.data(), wf.back(), off.data(), len.data(), arena, arena_bytes, c2.data(), l2.data(), cons_stride, s2))) return rc;
            for (int64_t k = 0; k < nk; ++k) {
                status[(size_t)keep[(size_t)k]] = s2[(size_t)k];
                if (s2[(size_t)k]) continue;
                cons_len[keep[(size_t)k]] = l2[(size_t)k];
                memcpy(cons + keep[(size_t)k] * cons_stride, c2.data() + (size_t)k * (size_t)cons_stride, (size_t)cons_stride);
            }
        }
    }
    // a graph that outgrew every node capacity int16 scores admit: the wide path has room (spoa has no such limit)
    for (int64_t w = 0; w < n_windows; ++w)
        if (!is_wide[(size_t)w] && (status[(size_t)w] & GBX_POA_ST_NODES)) wide.push_back(w);
    if ((rc = poa_host_wide(p, wide, win_first_seq, seq_off, seq_len, arena, arena_bytes, cons, cons_len, cons_stride, status))) return rc;
    int64_t n_bad = 0, first_bad = -1;
    for (int64_t w = 0; w < n_windows; ++w)
        if (status[(size_t)w]) { if (first_bad < 0) first_bad = w; ++n_bad; }
    if (n_bad) {
        set_error("gbx_poa_consensus_host: %lld window(s) exceeded a device capacity, first is window %lld (status bits 0x%x, "
                  "see GBX_POA_ST_*); the others' results are valid", (long long)n_bad, (long long)(win_base + first_bad), status[(size_t)first_bad]);
        return GBX_ERR_UNSUPPORTED;
    }
    return GBX_OK;
}


// The host entry: one device, or the windows cut into contiguous ranges of equal estimated cells over the devices of
// gbx_host_set_devices / GBX_GPUS - the driver's OpenMP loop over windows, one engine per thread
// (msa_spoa_omp.cpp:184-196,230-260), as a loop over devices.  A window's cells ~ (bases) x (mean length) x (1 + depth / 20)
// (shard.py:poa_cost: the graph starts as the first sequence and grows by about a tenth of every later one).
static int poa_host_entry(const gbx_poa_params *p, int64_t n_windows, const int64_t *win_first_seq,
                          int64_t n_seqs, const int64_t *seq_off, const int32_t *seq_len,
                          const char *arena, int64_t arena_bytes,
                          char *cons, int32_t *cons_len, int64_t cons_stride)
{
    auto one = [&]() { return poa_host_one(p, n_windows, win_first_seq, n_seqs, seq_off, seq_len, arena, arena_bytes, cons, cons_len, cons_stride); };
    if (!host_multi_wanted() || !p || n_windows <= 0 || n_seqs < 0 || arena_bytes < 0 || cons_stride <= 0 || !win_first_seq || !seq_off ||
        !seq_len || !arena || !cons || !cons_len || win_first_seq[0] != 0 || win_first_seq[n_windows] != n_seqs)
        return one();
    for (int64_t s = 0; s < n_seqs; ++s)
        if (seq_len[s] < 0 || seq_off[s] < 0 || seq_off[s] + seq_len[s] > arena_bytes) return one();
    for (int64_t w = 0; w < n_windows; ++w)
        if (win_first_seq[w + 1] < win_first_seq[w] || win_first_seq[w + 1] - win_first_seq[w] > GBX_POA_MAX_SEQS_PER_WINDOW) return one();
    int map[MAX_HOST_DEVICES];
    const int n_dev = host_device_set(map);
    if (n_dev < 0) return n_dev;
    const int parts = shard_parts(n_dev, n_windows, 256);
    if (parts == 1) {
        DeviceGuard g;
        int rc = g.set(map[host_next_small_call_device(n_dev)]);
        return rc ? rc : one();
    }
    const std::vector<int64_t> cuts = split_by_cost(n_windows, parts, [&](int64_t w) {
        double tot = 0.0;
        for (int64_t s = win_first_seq[w]; s < win_first_seq[w + 1]; ++s) tot += seq_len[s];
        const double n = win_first_seq[w + 1] > win_first_seq[w] ? (double)(win_first_seq[w + 1] - win_first_seq[w]) : 1.0;
        return tot * (tot / n) * (1.0 + n / 20.0);
    });
    return run_on_devices(parts, map, "gbx_poa_consensus_host", [&](int k) -> int {
        const int64_t lo = cuts[(size_t)k], hi = cuts[(size_t)k + 1], m = hi - lo;
        if (m == 0) return GBX_OK;
        const int64_t a = win_first_seq[lo], b = win_first_seq[hi];
        std::vector<int64_t> wf((size_t)m + 1), so((size_t)(b - a));
        for (int64_t w = 0; w <= m; ++w) wf[(size_t)w] = win_first_seq[lo + w] - a;
        int64_t a0 = arena_bytes, a1 = 0;
        for (int64_t s = a; s < b; ++s) { a0 = seq_off[s] < a0 ? seq_off[s] : a0; a1 = seq_off[s] + seq_len[s] > a1 ? seq_off[s] + seq_len[s] : a1; }
        if (a1 < a0) a0 = a1 = 0;
        for (int64_t s = a; s < b; ++s) so[(size_t)(s - a)] = seq_off[s] - a0;
        return poa_host_one(p, m, wf.data(), b - a, so.data(), seq_len + a, arena + a0, a1 - a0, cons + lo * cons_stride, cons_len + lo, cons_stride, lo, a);
    });
}

}  // extern "C"

// ---- small concurrent calls combined (host_combine.h).  The reference's driver builds one window per OpenMP thread and asks
// for its consensus (msa_spoa_omp.cpp:230-260; through include/spoa/spoa.hpp that is one one-window call per thread): the
// windows that are pending together become one job.  A request's consensus rows have its own stride; the combined call
// uses the widest and a consensus that would not have fitted its caller's rows sends that caller back through its own call,
// which reports it as it always did.
namespace {
struct PoaReq : CombineReq {
    const gbx_poa_params *p; int64_t n_windows; const int64_t *win_first_seq; int64_t n_seqs; const int64_t *seq_off; const int32_t *seq_len;
    const char *arena; int64_t arena_bytes; char *cons; int32_t *cons_len; int64_t cons_stride;
    int64_t cb;                           // bytes of its sequences laid end to end
};
struct PoaScratch { Scratch<int64_t> wf, off; Scratch<int32_t> len, clen; Scratch<char> arena, cons; };
constexpr int64_t POA_COMBINE_MAX_CALL = 64, POA_COMBINE_MAX_JOB = 16384;
}
namespace gbx { Combiner &combiner_poa() { static Combiner *c = new Combiner(); return *c; } }

static void poa_run_alone(PoaReq *r)
{
    r->rc = poa_host_entry(r->p, r->n_windows, r->win_first_seq, r->n_seqs, r->seq_off, r->seq_len, r->arena, r->arena_bytes, r->cons, r->cons_len,
                           r->cons_stride);
    if (r->rc) r->err = gbx_last_error();
}

static void poa_run_combined(const std::vector<CombineReq *> &batch, int slot)
{
    if (batch.size() == 1) { poa_run_alone((PoaReq *)batch[0]); return; }
    static PoaScratch *slots = new PoaScratch[Combiner::MAX_LEADERS];      // one per leader in flight (Combiner::submit)
    PoaScratch *S = slots + slot;
    const size_t nb = batch.size();
    std::vector<int64_t> w0(nb + 1, 0), s0(nb + 1, 0), b0(nb + 1, 0);
    int64_t stride = 1;
    for (size_t k = 0; k < nb; ++k) {
        const PoaReq *r = (const PoaReq *)batch[k];
        w0[k + 1] = w0[k] + r->n_windows; s0[k + 1] = s0[k] + r->n_seqs; b0[k + 1] = b0[k] + r->cb;
        stride = r->cons_stride > stride ? r->cons_stride : stride;
    }
    const int64_t NW = w0[nb], NS = s0[nb], NB = b0[nb];
    int64_t *mwf = S->wf.get((size_t)NW + 1), *moff = S->off.get((size_t)NS);
    int32_t *mlen = S->len.get((size_t)NS), *mcl = S->clen.get((size_t)NW);
    char *mar = S->arena.get((size_t)NB + 32), *mcons = S->cons.get((size_t)(NW * stride));
    for (size_t k = 0; k < nb; ++k) {
        const PoaReq *r = (const PoaReq *)batch[k];
        int64_t at = b0[k];
        for (int64_t j = 0; j < r->n_seqs; ++j) {
            memcpy(mar + at, r->arena + r->seq_off[j], (size_t)r->seq_len[j]);
            moff[s0[k] + j] = at; mlen[s0[k] + j] = r->seq_len[j];
            at += r->seq_len[j];
        }
        for (int64_t w = 0; w < r->n_windows; ++w) mwf[w0[k] + w] = s0[k] + r->win_first_seq[w];
    }
    mwf[NW] = NS;
    memset(mar + NB, 0, 32);
    const PoaReq *lead = (const PoaReq *)batch[0];
    const int rc = poa_host_entry(lead->p, NW, mwf, NS, moff, mlen, mar, NB, mcons, mcl, stride);
    if (rc) { for (CombineReq *q : batch) poa_run_alone((PoaReq *)q); return; }      // (a window over a capacity: its own call names it)
    for (size_t k = 0; k < nb; ++k) {
        PoaReq *r = (PoaReq *)batch[k];
        bool fits = true;
        for (int64_t w = 0; w < r->n_windows; ++w) fits = fits && mcl[w0[k] + w] <= r->cons_stride;
        if (!fits) { poa_run_alone(r); continue; }
        for (int64_t w = 0; w < r->n_windows; ++w) {
            r->cons_len[w] = mcl[w0[k] + w];
            memcpy(r->cons + w * r->cons_stride, mcons + (w0[k] + w) * stride, (size_t)mcl[w0[k] + w]);
        }
        r->rc = GBX_OK;
    }
}

extern "C" {

int gbx_poa_consensus_host(const gbx_poa_params *p, int64_t n_windows, const int64_t *win_first_seq,
                           int64_t n_seqs, const int64_t *seq_off, const int32_t *seq_len,
                           const char *arena, int64_t arena_bytes,
                           char *cons, int32_t *cons_len, int64_t cons_stride)
{
    auto plain = [&] { return poa_host_entry(p, n_windows, win_first_seq, n_seqs, seq_off, seq_len, arena, arena_bytes, cons, cons_len, cons_stride); };
    if (!p || n_windows <= 0 || n_windows > POA_COMBINE_MAX_CALL || n_seqs <= 0 || arena_bytes < 0 || cons_stride <= 0 || !win_first_seq || !seq_off ||
        !seq_len || !arena || !cons || !cons_len || win_first_seq[0] != 0 || win_first_seq[n_windows] != n_seqs || !combine_enabled() ||
        profile_active())
        return plain();
    PoaReq r;
    r.p = p; r.n_windows = n_windows; r.win_first_seq = win_first_seq; r.n_seqs = n_seqs; r.seq_off = seq_off; r.seq_len = seq_len;
    r.arena = arena; r.arena_bytes = arena_bytes; r.cons = cons; r.cons_len = cons_len; r.cons_stride = cons_stride; r.units = n_windows; r.cb = 0;
    for (int64_t w = 0; w < n_windows; ++w)
        if (win_first_seq[w + 1] < win_first_seq[w] || win_first_seq[w + 1] > n_seqs) return plain();
    for (int64_t k = 0; k < n_seqs; ++k) {
        if (seq_len[k] < 0 || seq_off[k] < 0 || seq_off[k] + seq_len[k] > arena_bytes) return plain();
        r.cb += seq_len[k];
    }
    if (hipGetDevice(&r.dev) != hipSuccess) { (void)hipGetLastError(); return plain(); }
    return combiner_poa().submit(&r, POA_COMBINE_MAX_JOB, Combiner::max_leaders(1),
        [](const CombineReq *a, const CombineReq *b) { return memcmp(((const PoaReq *)a)->p, ((const PoaReq *)b)->p, offsetof(gbx_poa_params, pad_)) == 0; },
        poa_run_combined);
}

}  // extern "C"
