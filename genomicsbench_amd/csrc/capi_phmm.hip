// capi_phmm.hip — phmm entries of the C-ABI (include/gbx.h).
#include "capi_common.h"
#include "phmm_split.h"

using namespace gbx;

extern "C" {

/* -------------------------------------------------------------------- phmm */
int gbx_phmm_init(void)
{
    int rc = require_device();
    if (rc) return rc;
    return phmm_init_tables();
}

size_t gbx_phmm_workspace_bytes(int64_t n_pairs, int64_t n_reads, int32_t max_hap_len)
{
    return phmm_workspace_bytes(n_pairs, n_reads, max_hap_len);
}

int gbx_phmm_forward_device(int64_t n_pairs, const int32_t *d_pair_read, const int32_t *d_pair_hap,
                            int64_t n_reads, const int64_t *d_read_off, const int32_t *d_read_len,
                            const uint8_t *d_rs, const uint8_t *d_q, const uint8_t *d_i, const uint8_t *d_d,
                            const uint8_t *d_c,
                            const int64_t *d_hap_off, const int32_t *d_hap_len, const uint8_t *d_hap,
                            int32_t max_hap_len, double *d_out, void *d_work, size_t work_bytes, void *stream)
{
    if (n_pairs < 0 || n_reads < 0 || max_hap_len < 0 || max_hap_len > GBX_PHMM_MAX_HAPLEN) {
        set_error("gbx_phmm_forward_device: bad argument");
        return GBX_ERR_ARG;
    }
    if (n_pairs == 0) return GBX_OK;
    if (!d_pair_read || !d_pair_hap || !d_read_off || !d_read_len || !d_rs || !d_q || !d_i || !d_d || !d_c ||
        !d_hap_off || !d_hap_len || !d_hap || !d_out || !d_work) {
        set_error("gbx_phmm_forward_device: null pointer");
        return GBX_ERR_ARG;
    }
    int rc = require_device();
    if (rc) return rc;
    return phmm_launch(n_pairs, d_pair_read, d_pair_hap, n_reads, d_read_off, d_read_len, d_rs, d_q, d_i, d_d, d_c,
                       d_hap_off, d_hap_len, d_hap, max_hap_len, d_out, d_work, work_bytes, (hipStream_t)stream);
}

// One device (the calling thread's current one).  pair_base / read_base / hap_base = indices of pair 0 / read 0 / haplotype 0
// in the caller's job (error texts only).
static int phmm_host_one(int64_t n_pairs, const int32_t *pair_read, const int32_t *pair_hap,
                         int64_t n_reads, const int64_t *read_off, const int32_t *read_len, int64_t read_bytes,
                         const uint8_t *rs, const uint8_t *q, const uint8_t *i, const uint8_t *d, const uint8_t *c,
                         int64_t n_haps, const int64_t *hap_off, const int32_t *hap_len, int64_t hap_bytes,
                         const uint8_t *hap, double *out, int64_t pair_base = 0, int64_t read_base = 0, int64_t hap_base = 0)
{
    RoctxRange range_("gbx_phmm_forward_host");
    const bool trace = getenv("GBX_HOST_TRACE") != nullptr;
    const double t_begin = wall_s();
    auto mark = [&](const char *what) { if (trace) fprintf(stderr, "[gbx phmm host] %9.3f ms %s\n", (wall_s() - t_begin) * 1e3, what); };
    if (n_pairs < 0 || n_reads < 0 || n_haps < 0 || read_bytes < 0 || hap_bytes < 0) {
        set_error("gbx_phmm_forward_host: bad argument");
        return GBX_ERR_ARG;
    }
    if (n_pairs == 0) return GBX_OK;
    if (!pair_read || !pair_hap || !read_off || !read_len || !rs || !q || !i || !d || !c || !hap_off || !hap_len ||
        !hap || !out) {
        set_error("gbx_phmm_forward_host: null pointer");
        return GBX_ERR_ARG;
    }
    int max_h = 1;
    // exact size of the haplotype streams (sum over the pairs of haplen+1): one long haplotype must not size the
    // workspace of every pair
    int64_t stream_syms = 0;
    for (int64_t k = 0; k < n_reads; ++k)
        if (read_len[k] < 0 || read_off[k] < 0 || read_off[k] + read_len[k] > read_bytes) {
            set_error("gbx_phmm_forward_host: read %lld lies outside the arena", (long long)(read_base + k));
            return GBX_ERR_ARG;
        }
    for (int64_t k = 0; k < n_haps; ++k) {
        if (hap_len[k] < 0 || hap_off[k] < 0 || hap_off[k] + hap_len[k] > hap_bytes) {
            set_error("gbx_phmm_forward_host: haplotype %lld lies outside the arena", (long long)(hap_base + k));
            return GBX_ERR_ARG;
        }
        if (hap_len[k] > GBX_PHMM_MAX_HAPLEN) {
            set_error("gbx_phmm_forward_host: haplotype %lld longer than GBX_PHMM_MAX_HAPLEN", (long long)(hap_base + k));
            return GBX_ERR_UNSUPPORTED;
        }
        if (hap_len[k] > max_h) max_h = hap_len[k];
    }
    {
        // the pair list is the long one (10.9 M entries in the 'large' job): a few threads, lowest bad index reported
        const int T = host_workers();
        std::vector<int64_t> bad((size_t)T, -1), syms((size_t)T, 0);
        parallel_ranges(n_pairs, T, [&](int t, int64_t lo, int64_t hi) {
            int64_t sum = 0;
            for (int64_t k = lo; k < hi; ++k) {
                if (pair_read[k] < 0 || pair_read[k] >= n_reads || pair_hap[k] < 0 || pair_hap[k] >= n_haps) { bad[(size_t)t] = k; return; }
                sum += (int64_t)hap_len[pair_hap[k]] + 1;
            }
            syms[(size_t)t] = sum;
        });
        for (int t = 0; t < T; ++t) stream_syms += syms[(size_t)t];
        for (int t = 0; t < T; ++t)
            if (bad[(size_t)t] >= 0) {
                set_error("gbx_phmm_forward_host: pair %lld names a read/haplotype out of range", (long long)(pair_base + bad[(size_t)t]));
                return GBX_ERR_ARG;
            }
    }
    mark("validated");
    int rc = require_device();
    if (rc) return rc;
    HostLane lane;
    if ((rc = lane.acquire())) return rc;
    Lane *L = lane.l;
    DevBuf dpr(L), dph(L), dro(L), drl(L), drs(L), dq(L), di(L), dd(L), dc(L), dho(L), dhl(L), dh(L), dout(L), dw(L);
    const size_t wb = phmm_workspace_bytes(n_pairs, n_reads, max_h, stream_syms);
    if ((rc = dpr.alloc(n_pairs * 4)) || (rc = dph.alloc(n_pairs * 4)) || (rc = dro.alloc(n_reads * 8)) ||
        (rc = drl.alloc(n_reads * 4)) || (rc = drs.alloc(read_bytes)) || (rc = dq.alloc(read_bytes)) ||
        (rc = di.alloc(read_bytes)) || (rc = dd.alloc(read_bytes)) || (rc = dc.alloc(read_bytes)) ||
        (rc = dho.alloc(n_haps * 8)) || (rc = dhl.alloc(n_haps * 4)) || (rc = dh.alloc(hap_bytes)) ||
        (rc = dout.alloc(n_pairs * 8)) || (rc = dw.alloc(wb)))
        return rc;
    // A large staged call uploads in two stages: what the grouping passes of the launch read (pair lists, length tables, haplotypes:
    // a fifth of the bytes) first, the reads' bases and four quality tracks behind - and the launch makes the stream wait for the
    // second stage only where its first kernel needs it (phmm_split.h), so that the grouping (5 ms on 'large') runs under the upload.
    // GBX_PHMM_UPLOAD_STAGES=1: one stage, as before.
    const size_t up_bytes = (size_t)read_bytes * 5 + (size_t)hap_bytes + (size_t)n_pairs * 8 + (size_t)(n_reads + n_haps) * 12;
    const bool two = up_bytes >= ((size_t)64 << 20) && !(getenv("GBX_PHMM_UPLOAD_STAGES") && atoi(getenv("GBX_PHMM_UPLOAD_STAGES")) == 1) &&
                     !getenv("GBX_HOST_PAGEABLE");
    HostPipe pipe(lane.l, up_bytes, two);
    const bool staged2 = two && pipe.staged;
    if ((rc = pipe.prepare(1))) return rc;
    if (staged2) pipe.upload_stages(2);
    const int64_t s1 = staged2 ? 1 : 0;
    pipe.stage(0, dpr.p, pair_read, n_pairs * 4); pipe.stage(0, dph.p, pair_hap, n_pairs * 4);
    pipe.stage(0, dro.p, read_off, n_reads * 8); pipe.stage(0, drl.p, read_len, n_reads * 4);
    pipe.stage(0, dho.p, hap_off, n_haps * 8); pipe.stage(0, dhl.p, hap_len, n_haps * 4);
    pipe.stage(0, dh.p, hap, hap_bytes);
    pipe.stage(s1, drs.p, rs, read_bytes); pipe.stage(s1, dq.p, q, read_bytes); pipe.stage(s1, di.p, i, read_bytes);
    pipe.stage(s1, dd.p, d, read_bytes); pipe.stage(s1, dc.p, c, read_bytes);
    mark("device buffers ready");
    pipe.start();
    if ((rc = pipe.wait_stage(0))) return pipe.finish(rc);
    mark(staged2 ? "first upload stage queued" : "uploads queued");
    const std::function<int()> between = [&]() -> int {
        const int r = pipe.wait_stage(1);
        mark("uploads queued");
        return r;
    };
    if (staged2) phmm_set_between(&between);
    rc = phmm_launch(n_pairs, dpr.as<int32_t>(), dph.as<int32_t>(), n_reads, dro.as<int64_t>(), drl.as<int32_t>(),
                     drs.as<uint8_t>(), dq.as<uint8_t>(), di.as<uint8_t>(), dd.as<uint8_t>(), dc.as<uint8_t>(),
                     dho.as<int64_t>(), dhl.as<int32_t>(), dh.as<uint8_t>(), max_h, dout.as<double>(), dw.p, wb,
                     lane.l->compute, stream_syms);
    if (rc) return pipe.finish(rc);
    pipe.fetch(0, out, dout.p, n_pairs * 8);
    if ((rc = pipe.chunk_launched(0))) return pipe.finish(rc);
    mark("kernels queued");
    rc = pipe.finish();
    mark("results downloaded");
    return rc;
}


// The host entry: one device, or the pair list cut into contiguous ranges of equal cells (read length x haplotype length)
// over the devices of gbx_host_set_devices / GBX_GPUS - the driver's OpenMP loop over slices of the testcase array
// (PairHMMUnitTest.cpp:224-247) as a loop over devices.  A shard takes the reads / haplotypes its pairs name (the id
// ranges they span: the testcase array is read-major, so these are the shard's own batches) and the byte ranges of the
// arenas those occupy.
static int phmm_host_entry(int64_t n_pairs, const int32_t *pair_read, const int32_t *pair_hap,
                           int64_t n_reads, const int64_t *read_off, const int32_t *read_len, int64_t read_bytes,
                           const uint8_t *rs, const uint8_t *q, const uint8_t *i, const uint8_t *d, const uint8_t *c,
                           int64_t n_haps, const int64_t *hap_off, const int32_t *hap_len, int64_t hap_bytes,
                           const uint8_t *hap, double *out)
{
    auto one = [&]() { return phmm_host_one(n_pairs, pair_read, pair_hap, n_reads, read_off, read_len, read_bytes, rs, q, i, d, c,
                                            n_haps, hap_off, hap_len, hap_bytes, hap, out); };
    if (!host_multi_wanted() || n_pairs <= 0 || n_reads <= 0 || n_haps <= 0 || read_bytes < 0 || hap_bytes < 0 || !pair_read || !pair_hap ||
        !read_off || !read_len || !rs || !q || !i || !d || !c || !hap_off || !hap_len || !hap || !out)
        return one();
    // the whole job's tables are checked once, as the one-device path does (same texts, same order); the shards then
    // only see valid input
    for (int64_t k = 0; k < n_reads; ++k)
        if (read_len[k] < 0 || read_off[k] < 0 || read_off[k] + read_len[k] > read_bytes) return one();
    for (int64_t k = 0; k < n_haps; ++k)
        if (hap_len[k] < 0 || hap_off[k] < 0 || hap_off[k] + hap_len[k] > hap_bytes || hap_len[k] > GBX_PHMM_MAX_HAPLEN) return one();
    {
        const int T = host_workers();
        std::vector<char> bad((size_t)T, 0);
        parallel_ranges(n_pairs, T, [&](int t, int64_t lo, int64_t hi) {
            for (int64_t j = lo; j < hi; ++j)
                if (pair_read[j] < 0 || pair_read[j] >= n_reads || pair_hap[j] < 0 || pair_hap[j] >= n_haps) { bad[(size_t)t] = 1; return; }
        });
        for (char b : bad) if (b) return one();
    }
    int map[MAX_HOST_DEVICES];
    const int n_dev = host_device_set(map);
    if (n_dev < 0) return n_dev;
    const int parts = shard_parts(n_dev, n_pairs, 262144);
    if (parts == 1) {
        DeviceGuard g;
        int rc = g.set(map[host_next_small_call_device(n_dev)]);
        return rc ? rc : one();
    }
    const std::vector<int64_t> cuts = split_by_cost(n_pairs, parts, [&](int64_t k) {
        const int64_t r = pair_read[k], h = pair_hap[k];
        return r >= 0 && r < n_reads && h >= 0 && h < n_haps ? (double)read_len[r] * (double)hap_len[h] : 0.0; });
    return run_on_devices(parts, map, "gbx_phmm_forward_host", [&](int k) -> int {
        const int64_t lo = cuts[(size_t)k], hi = cuts[(size_t)k + 1], m = hi - lo;
        if (m == 0) return GBX_OK;
        int64_t r0 = n_reads, r1 = -1, h0 = n_haps, h1 = -1;
        for (int64_t j = lo; j < hi; ++j) {
            const int64_t r = pair_read[j], h = pair_hap[j];
            r0 = r < r0 ? r : r0; r1 = r > r1 ? r : r1; h0 = h < h0 ? h : h0; h1 = h > h1 ? h : h1;
        }
        const int64_t nr = r1 - r0 + 1, nh = h1 - h0 + 1;
        int64_t ra = read_bytes, rb = 0, ha = hap_bytes, hb = 0;
        for (int64_t r = r0; r <= r1; ++r) { ra = read_off[r] < ra ? read_off[r] : ra; rb = read_off[r] + read_len[r] > rb ? read_off[r] + read_len[r] : rb; }
        for (int64_t h = h0; h <= h1; ++h) { ha = hap_off[h] < ha ? hap_off[h] : ha; hb = hap_off[h] + hap_len[h] > hb ? hap_off[h] + hap_len[h] : hb; }
        std::vector<int32_t> pr((size_t)m), ph((size_t)m);
        for (int64_t j = 0; j < m; ++j) { pr[(size_t)j] = (int32_t)(pair_read[lo + j] - r0); ph[(size_t)j] = (int32_t)(pair_hap[lo + j] - h0); }
        std::vector<int64_t> ro((size_t)nr), ho((size_t)nh);
        for (int64_t r = 0; r < nr; ++r) ro[(size_t)r] = read_off[r0 + r] - ra;
        for (int64_t h = 0; h < nh; ++h) ho[(size_t)h] = hap_off[h0 + h] - ha;
        return phmm_host_one(m, pr.data(), ph.data(), nr, ro.data(), read_len + r0, rb - ra, rs + ra, q + ra, i + ra, d + ra, c + ra,
                             nh, ho.data(), hap_len + h0, hb - ha, hap + ha, out + lo, lo, r0, h0);
    });
}

}  // extern "C"

// ---- small concurrent calls combined (host_combine.h).  The reference's driver calls computelikelihoodsboth once per batch
// (a few hundred pairs) from every OpenMP thread (PairHMMUnitTest.cpp:224-247): the calls that are pending together become
// one job - reads, haplotypes and pairs of all requests end to end, ids and offsets re-based.
namespace {
struct PhmmReq : CombineReq {
    int64_t n_pairs; const int32_t *pair_read, *pair_hap;
    int64_t n_reads; const int64_t *read_off; const int32_t *read_len; int64_t read_bytes;
    const uint8_t *rs, *q, *i, *d, *c;
    int64_t n_haps; const int64_t *hap_off; const int32_t *hap_len; int64_t hap_bytes; const uint8_t *hap; double *out;
    int64_t cr, ch;                       // bytes of its reads / haplotypes laid end to end
};
struct PhmmScratch {
    Scratch<int32_t> pr, ph, rl, hl; Scratch<int64_t> ro, ho; Scratch<uint8_t> rs, q, i, d, c, hap; Scratch<double> out;
};
constexpr int64_t PHMM_COMBINE_MAX_CALL = 131072, PHMM_COMBINE_MAX_JOB = (int64_t)4 << 20;
}
namespace gbx { Combiner &combiner_phmm() { static Combiner *c = new Combiner(); return *c; } }

static void phmm_run_alone(PhmmReq *r)
{
    r->rc = phmm_host_entry(r->n_pairs, r->pair_read, r->pair_hap, r->n_reads, r->read_off, r->read_len, r->read_bytes, r->rs, r->q, r->i, r->d,
                            r->c, r->n_haps, r->hap_off, r->hap_len, r->hap_bytes, r->hap, r->out);
    if (r->rc) r->err = gbx_last_error();
}

static void phmm_run_combined(const std::vector<CombineReq *> &batch, int slot)
{
    if (batch.size() == 1) { phmm_run_alone((PhmmReq *)batch[0]); return; }
    static PhmmScratch *slots = new PhmmScratch[Combiner::MAX_LEADERS];      // one per leader in flight (Combiner::submit)
    PhmmScratch *S = slots + slot;
    const size_t nb = batch.size();
    std::vector<int64_t> p0(nb + 1, 0), r0(nb + 1, 0), h0(nb + 1, 0), rb(nb + 1, 0), hb(nb + 1, 0);
    for (size_t k = 0; k < nb; ++k) {
        const PhmmReq *r = (const PhmmReq *)batch[k];
        p0[k + 1] = p0[k] + r->n_pairs; r0[k + 1] = r0[k] + r->n_reads; h0[k + 1] = h0[k] + r->n_haps;
        rb[k + 1] = rb[k] + r->cr; hb[k + 1] = hb[k] + r->ch;
    }
    const int64_t NP = p0[nb], NR = r0[nb], NH = h0[nb], RB = rb[nb], HB = hb[nb];
    if (NR > 0x7fffffffLL || NH > 0x7fffffffLL) { for (CombineReq *q : batch) phmm_run_alone((PhmmReq *)q); return; }
    int32_t *mpr = S->pr.get((size_t)NP), *mph = S->ph.get((size_t)NP), *mrl = S->rl.get((size_t)NR), *mhl = S->hl.get((size_t)NH);
    int64_t *mro = S->ro.get((size_t)NR), *mho = S->ho.get((size_t)NH);
    uint8_t *mrs = S->rs.get((size_t)RB + 16), *mq = S->q.get((size_t)RB + 16), *mi = S->i.get((size_t)RB + 16), *md = S->d.get((size_t)RB + 16),
            *mc = S->c.get((size_t)RB + 16), *mhap = S->hap.get((size_t)HB + 16);
    double *mout = S->out.get((size_t)NP);
    combine_parallel((int64_t)nb, host_workers(), [&](int64_t k) {
        const PhmmReq *r = (const PhmmReq *)batch[(size_t)k];
        int64_t at = rb[(size_t)k];
        for (int64_t j = 0; j < r->n_reads; ++j) {
            const int64_t o = r->read_off[j]; const size_t l = (size_t)r->read_len[j];
            memcpy(mrs + at, r->rs + o, l); memcpy(mq + at, r->q + o, l); memcpy(mi + at, r->i + o, l); memcpy(md + at, r->d + o, l); memcpy(mc + at, r->c + o, l);
            mro[r0[(size_t)k] + j] = at; mrl[r0[(size_t)k] + j] = r->read_len[j];
            at += (int64_t)l;
        }
        at = hb[(size_t)k];
        for (int64_t j = 0; j < r->n_haps; ++j) {
            memcpy(mhap + at, r->hap + r->hap_off[j], (size_t)r->hap_len[j]);
            mho[h0[(size_t)k] + j] = at; mhl[h0[(size_t)k] + j] = r->hap_len[j];
            at += r->hap_len[j];
        }
        const int32_t dr = (int32_t)r0[(size_t)k], dh = (int32_t)h0[(size_t)k];
        for (int64_t j = 0; j < r->n_pairs; ++j) { mpr[p0[(size_t)k] + j] = r->pair_read[j] + dr; mph[p0[(size_t)k] + j] = r->pair_hap[j] + dh; }
    });
    const int rc = phmm_host_entry(NP, mpr, mph, NR, mro, mrl, RB, mrs, mq, mi, md, mc, NH, mho, mhl, HB, mhap, mout);
    if (rc) { for (CombineReq *q : batch) phmm_run_alone((PhmmReq *)q); return; }
    for (size_t k = 0; k < nb; ++k) {
        PhmmReq *r = (PhmmReq *)batch[k];
        memcpy(r->out, mout + p0[k], (size_t)r->n_pairs * sizeof(double));
        r->rc = GBX_OK;
    }
}

extern "C" {

int gbx_phmm_forward_host(int64_t n_pairs, const int32_t *pair_read, const int32_t *pair_hap,
                          int64_t n_reads, const int64_t *read_off, const int32_t *read_len, int64_t read_bytes,
                          const uint8_t *rs, const uint8_t *q, const uint8_t *i, const uint8_t *d, const uint8_t *c,
                          int64_t n_haps, const int64_t *hap_off, const int32_t *hap_len, int64_t hap_bytes,
                          const uint8_t *hap, double *out)
{
    auto plain = [&] { return phmm_host_entry(n_pairs, pair_read, pair_hap, n_reads, read_off, read_len, read_bytes, rs, q, i, d, c,
                                              n_haps, hap_off, hap_len, hap_bytes, hap, out); };
    if (n_pairs <= 0 || n_pairs > PHMM_COMBINE_MAX_CALL || n_reads <= 0 || n_haps <= 0 || read_bytes < 0 || hap_bytes < 0 || !pair_read || !pair_hap ||
        !read_off || !read_len || !rs || !q || !i || !d || !c || !hap_off || !hap_len || !hap || !out || !combine_enabled() || profile_active())
        return plain();
    PhmmReq r;
    r.n_pairs = n_pairs; r.pair_read = pair_read; r.pair_hap = pair_hap; r.n_reads = n_reads; r.read_off = read_off; r.read_len = read_len;
    r.read_bytes = read_bytes; r.rs = rs; r.q = q; r.i = i; r.d = d; r.c = c; r.n_haps = n_haps; r.hap_off = hap_off; r.hap_len = hap_len;
    r.hap_bytes = hap_bytes; r.hap = hap; r.out = out; r.units = n_pairs; r.cr = r.ch = 0;
    // a call with a bad table entry goes its own way: its error names the entry
    for (int64_t k = 0; k < n_reads; ++k) {
        if (read_len[k] < 0 || read_off[k] < 0 || read_off[k] + read_len[k] > read_bytes) return plain();
        r.cr += read_len[k];
    }
    for (int64_t k = 0; k < n_haps; ++k) {
        if (hap_len[k] < 0 || hap_off[k] < 0 || hap_off[k] + hap_len[k] > hap_bytes || hap_len[k] > GBX_PHMM_MAX_HAPLEN) return plain();
        r.ch += hap_len[k];
    }
    for (int64_t k = 0; k < n_pairs; ++k)
        if (pair_read[k] < 0 || pair_read[k] >= n_reads || pair_hap[k] < 0 || pair_hap[k] >= n_haps) return plain();
    if (hipGetDevice(&r.dev) != hipSuccess) { (void)hipGetLastError(); return plain(); }
    return combiner_phmm().submit(&r, PHMM_COMBINE_MAX_JOB, Combiner::max_leaders(2), [](const CombineReq *, const CombineReq *) { return true; }, phmm_run_combined);
}

}  // extern "C"
