#!/usr/bin/env python3
"""bench.py — headline benchmark: bsw 'large' GCUPS on N MI355X (BASELINE.json), one JSON line on rank 0.

    python bench.py [--gpus N --steps K --warmup W]     # bsw large = the headline; chain, phmm and poa 'large' follow
                                                        # under "kernels" in the same JSON line
    python bench.py --kernel bsw|chain|phmm|poa         # one kernel only, same JSON shape
    python bench.py --mode local                        # every rank generates its own shard (no scatter / gather)

A step = one pass of the kernel's hot path (every launch of the *_device entry point) over this rank's shard of the
synthetic dataset, inputs resident in HBM.  Work units are independent (pairs / calls / batches / windows), so
the job shards over ranks with no data-path collective ("weak": the dataset is N x the 'large' config):

  mode "scatter" (default): rank 0 owns the whole dataset, cuts it into N cost-balanced contiguous shards
  (genomicsbench_amd/shard.py), ships each shard as one packed message over RCCL point-to-point (scatter_ms),
  every rank runs the timed steps on its HBM-resident shard, rank 0 gathers the fixed-stride outputs (gather_ms)
  and checks units from every rank's shard against the CPU oracle.  With N = 1 the scatter is the H2D copy and the
  gather the D2H copy.

value = units of ALL ranks per second of the slowest rank over the K timed steps (barrier + synchronize on both
sides).  Per-kernel durations come from HIP events recorded around every launch on the launch stream inside the
timed region (gbx_profile_begin/end).
"""
import argparse
import re
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md

# query-length range handled by each bsw row kernel (csrc/bsw_kernels.hip: cls_of / class_shapes; the default
# shape of a class has lanes x columns == the longest query of the class)
_BSW_QMAX = [8, 16, 24, 32, 40, 48, 56, 64, 72, 80, 88, 96, 104, 112, 120, 128, 160, 192, 256, 1024]
_BSW_SHAPE = ["2x4", "2x8", "2x12", "2x16", "2x20", "2x24", "4x14", "4x16", "4x18", "4x20", "4x22", "4x24", "8x13", "8x14", "8x15",
              "8x16", "16x10", "16x12", "16x16", "64x16"]
BSW_CLASS = {"bsw_rows_" + sh: ((_BSW_QMAX[k - 1] + 1) if k else 1, _BSW_QMAX[k]) for k, sh in enumerate(_BSW_SHAPE)}
BSW_CLASS["bsw_lds"] = (1025, 1 << 30)
BSW_LANE_HI = {"c": [47, 79, 99, 135, 159], "w": [39, 79, 103, 127, 159]}
# read-length range handled by each phmm kernel (csrc/phmm_kernels.hip: 31 row lanes x K rows per lane on the stream
# path, one pair per wavefront beyond 248 rows)
PHMM_CLASS = {"phmm_stream_rpl%d" % k: (31 * (k - 1) + 1, 31 * k) for k in range(1, 9)}
PHMM_CLASS.update({"phmm_f32_rpl4": (249, 256), "phmm_f32_rpl6": (257, 384), "phmm_f32_rpl8": (385, 1 << 30)})


# ------------------------------------------------------------------------------------------ workloads
def _cpu_quota():
    """CPU time the container may use, in cores (cgroup v2 cpu.max), or None: os.cpu_count() counts the host's hardware
    threads, which a quota does not change - a baseline on "256 cores" of a box with a 16-core quota ran on sixteen."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else round(float(q) / float(per), 2)
    except (OSError, ValueError):
        return None


def _host_timed(fn, reps=4):
    ms = []
    res = None
    for _ in range(reps):
        res = None                                          # the previous call's result arrays are freed outside the timed region
        t0 = time.perf_counter()
        out = fn()
        ms.append((time.perf_counter() - t0) * 1e3)
        res = out
        del out
    return ms, res


class BswWork:
    metric, unit, dtype = "bsw_large_gcups", "GCUPS", "int32"

    large, seed = 2_000_000, 1002

    def __init__(self, args):
        from genomicsbench_amd.bsw import make_params
        self.n = args.size or self.large
        self.params = make_params()
        self.workload = "bsw large: %d synthetic 151-bp seed-extension pairs per GPU (seed 1002), inputs resident in HBM" % self.n
        self.detail = {"cell": "nominal cells = sum len1*len2 (main_banded.cpp:183,323)"}

    # host side (rank 0 in scatter mode, every rank in local mode)
    def generate(self, first, n_units):
        from genomicsbench_amd.datagen import gen_bsw
        return gen_bsw(n_units, self.seed, first=first)

    def shards(self, full, parts):
        return S.bsw_shards(full, parts)

    to_arrays = staticmethod(lambda sh: S.bsw_to_arrays(sh))
    n_units = staticmethod(lambda sh: sh.n)

    def attach(self, tensors, dev, host_shard):
        """tensors: this rank's shard in HBM (from the scatter); host_shard: its host copy where it exists (rank 0)."""
        from genomicsbench_amd.bsw import DeviceBswBatch
        self.d = DeviceBswBatch.from_tensors(tensors, dev)
        self.batch = host_shard
        self.units = float((self.d.len1.long() * self.d.len2.long()).sum().item())
        self.extra = {"pairs_this_gpu": self.d.n, "nominal_cells_this_gpu": int(self.units)}

    def output_tensor(self):
        return self.d.out[:self.d.n]

    def check_gathered(self, full, ranges, parts, sample):
        """Units from the front of every rank's shard: gathered device results == the oracle's, all 6 fields."""
        from oracle import oracle_py as O
        bad = checked = 0
        for (lo, hi), got in zip(ranges, parts):
            m = min(hi - lo, sample)
            if m:
                want = O.bsw_oracle(self.params, full.slice(lo, lo + m), min(os.cpu_count() or 1, 32))
                bad += int(not np.array_equal(got[:m].cpu().numpy(), want))
                checked += m
        return "%d pairs (front of every shard) vs oracle, all 6 fields: %s" % (checked, "identical" if not bad else "DIFFER")

    def run(self, stream):
        self.d.run(self.params, stream)

    def roofline_bytes(self, kernel):
        b = self.batch
        if kernel in ("bsw_lane_sort", "bsw_classify", "bsw_unpack4") or not kernel.startswith("bsw_"):
            return 0, 0.0                                       # pre-passes: no pairs of their own
        if kernel.startswith("bsw_lane_"):
            # lane-per-pair kernels (csrc/bsw_kernels.hip: LANE_RANGE_HI): format c = every score below 256, w = below 8192
            fmt, hi = kernel[9], int(kernel[10:])
            his = BSW_LANE_HI[fmt]
            lo = his[his.index(hi) - 1] + 1 if his.index(hi) else 1
            bound = b.h0.astype(np.int64) + b.len2.astype(np.int64) * max(0, max(int(v) for v in self.params.mat))
            on = b.n >= int(os.environ.get("GBX_BSW_LANE_MIN", 262144)) if "GBX_BSW_LANE" not in os.environ else os.environ["GBX_BSW_LANE"] != "0"
            sel = (b.len2 >= lo) & (b.len2 <= hi) & (b.len1 >= 1) & (b.h0 >= 0) & ((bound < 256) if fmt == "c" else ((bound >= 256) & (bound < 8192))) & on
            units = float((b.len1[sel].astype(np.int64) * b.len2[sel]).sum())
            return int(b.len1[sel].astype(np.int64).sum() + b.len2[sel].astype(np.int64).sum() + 36 * sel.sum()), units
        lo, hi = BSW_CLASS.get(kernel, (1, 1 << 30))
        # small jobs run several query classes on one kernel (bsw_kernels.hip: class_mode_for)
        mode = int(os.environ.get("GBX_BSW_CLASSMODE", 2 if b.n < 32768 else 1 if b.n < 250000 else 0))
        if mode == 1:
            lo, hi = {"bsw_rows_2x16": (1, 32), "bsw_rows_4x16": (33, 64), "bsw_rows_4x24": (65, 96),
                      "bsw_rows_8x16": (97, 128), "bsw_rows_16x16": (129, 256)}.get(kernel, (lo, hi))
        elif mode == 2:
            lo, hi = {"bsw_rows_8x16": (1, 128), "bsw_rows_16x16": (129, 256)}.get(kernel, (lo, hi))
        sel = (b.len2 >= lo) & (b.len2 <= hi)
        units = float((b.len1[sel].astype(np.int64) * b.len2[sel]).sum())
        return int(b.len1[sel].astype(np.int64).sum() + b.len2[sel].astype(np.int64).sum() + 36 * sel.sum()), units

    def host_entry(self):
        """PCIe-inclusive rate of the same shard through gbx_bsw_extend_host (pageable host arrays in, results
        out): reported beside `value`, never as it."""
        from genomicsbench_amd import _native as N
        from genomicsbench_amd.bsw import extend_host
        N.check(N.lib().gbx_host_prepare())
        ms, out = [], np.full((self.batch.n, 6), -1, dtype=np.int32)      # touched pages, like a caller's SeqPair array
        # twelve calls: the oracle check just before leaves an OpenMP team winding down on the host cores, which costs the
        # first two calls 2-3 ms each (measured: 16.9, 17.2, then 14.0 ms), and one call in ten on the pool's boxes stalls for
        # 5-8 ms in its uploads (with every version of the pipeline: profiles/r05am_many.txt); best and median are reported
        for _ in range(12):
            t0 = time.perf_counter()
            extend_host(self.params, self.batch, out)
            ms.append((time.perf_counter() - t0) * 1e3)
        got = self.d.results()
        got = np.stack([got[f] for f in ("score", "tle", "gtle", "qle", "gscore", "max_off")], axis=1) if got.dtype.names else got
        return {"first_call_ms": ms[0], "best_ms": min(ms), "median_ms": float(np.median(ms[2:])), "calls": len(ms),
                "value": self.batch.nominal_cells / (min(ms) * 1e-3) / 1e9,
                "unit": "GCUPS", "what": "gbx_bsw_extend_host on the rank-0 shard: H2D + kernels + D2H from pageable memory",
                "same_as_device_entry": bool(np.array_equal(np.asarray(got), out))}

    def e2e(self):
        """SURVEY 8d leg (iii): the GPU driver (reference CLI, genomicsbench_amd/bin/bsw) on the same pairs as the reference's input
        file, `-t <host cores>`: conversion of the text + H2D + kernels + D2H, once with the parse and the device calls
        overlapped in four slices (--overlap 4) and once one after the other.  A process of its own (its own HIP context)."""
        import subprocess, tempfile
        from genomicsbench_amd.datagen import write_bsw_pairs_fast
        exe = os.path.join(ROOT, "genomicsbench_amd", "bin", "bsw")
        if not os.path.exists(exe):
            return {"error": "genomicsbench_amd/bin/bsw is not built"}
        cores = os.cpu_count() or 1
        # (ingest threads capped at the container's CPU quota were measured slower on the pool's boxes - 16-core quota, 256 hardware
        # threads: 66.6 ms end to end with -t 16 against 50.9 with -t 64, profiles/r06e_bench.json: a quota is CPU time per period,
        # and a short burst on many cores finishes inside it - so the cap stays at 64; `e2e_flow` says which flow won)
        d = tempfile.mkdtemp(prefix="gbx_bsw_e2e_")
        path = os.path.join(d, "pairs.txt")
        try:
            size = write_bsw_pairs_fast(path, self.batch)
            recs = {}
            for name, extra in (("overlapped", ["--overlap", "4"]), ("serial", [])):
                best = None
                for _ in range(2):
                    r = subprocess.run([exe, "-pairs", path, "-t", str(min(cores, 64)), "-b", str(self.batch.n)] + extra, capture_output=True, text=True, timeout=300)
                    if r.returncode != 0:
                        return {"error": (r.stderr or r.stdout)[-200:]}
                    j = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
                    if best is None or j["e2e_seconds"] < best["e2e_seconds"]:
                        best = j
                recs[name] = best
            # the same driver with its binary input cache (--cache: the converted arrays mapped instead of the text parsed again);
            # the first run writes it, the timed ones map it
            cpath = os.path.join(d, "pairs.gbxcache")
            cached = None
            for k in range(3):
                r = subprocess.run([exe, "-pairs", path, "-t", str(min(cores, 64)), "-b", str(self.batch.n), "--cache", cpath], capture_output=True, text=True, timeout=300)
                if r.returncode != 0:
                    cached = None
                    break
                j = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
                if k and "mapped from the cache" in r.stdout and (cached is None or j["e2e_seconds"] < cached["e2e_seconds"]):
                    cached = j
            if os.path.exists(cpath):
                os.remove(cpath)
            refdrv = self.refdriver(path)
        finally:
            try:
                os.remove(path)
                os.rmdir(d)
            except OSError:
                pass
        o, q = recs["overlapped"], recs["serial"]
        # (measured, profiles/r05k_bsw_e2e.txt: on the pool's boxes - 16 cores of CPU quota - the overlapped flow is the slower
        # one: every slice's call pays for fresh device buffers and shares the cores with the conversion; both are reported)
        return {"e2e_ms": min(o["e2e_seconds"], q["e2e_seconds"]) * 1e3, "e2e_flow": "overlapped" if o["e2e_seconds"] < q["e2e_seconds"] else "parse_then_call",
                "refdriver": refdrv, "e2e_ms_cached_input": cached["e2e_seconds"] * 1e3 if cached else None,
                "cached_ingest_ms": cached["ingest_seconds"] * 1e3 if cached else None, "e2e_ms_overlapped": o["e2e_seconds"] * 1e3,
                "e2e_ms_parse_then_call": q["e2e_seconds"] * 1e3, "ingest_ms": q["ingest_seconds"] * 1e3,
                "call_ms_in_driver": q["seconds"] * 1e3, "ingest_threads": o["ingest_threads"], "input_file_mb": round(size / 1e6, 1),
                "what": "genomicsbench_amd/bin/bsw -pairs <file> -t %d: text conversion + H2D + kernels + D2H, file already in memory"
                        % o["ingest_threads"]}

    def refdriver(self, path):
        """The reference's UNMODIFIED driver (main_banded.cpp on csrc/shims/bsw_class_shim.cpp: oracle/_ref/bsw_refdriver_gbx) on
        the same file, called as the reference's scripts call it - one getScores16 per 512 pairs per OpenMP thread
        (run-cpu.sh:61) -: its own timed region ("Overall SW cycles"), the concurrent calls combined by the host entry."""
        import re, subprocess
        exe = os.path.join(ROOT, "oracle", "_ref", "bsw_refdriver_gbx")
        if not os.path.exists(exe):
            return None
        out = {"what": "unmodified main_banded.cpp on the shim, the driver's own 'Overall SW cycles' region, seconds"}
        for key, t, b in (("t64_b512_s", 64, 512), ("t16_b512_s", 16, 512), ("t1_one_call_s", 1, self.batch.n)):
            try:
                r = subprocess.run([exe, "-pairs", path, "-t", str(t), "-b", str(b)], capture_output=True, text=True, timeout=300)
                m = re.search(r"Overall SW cycles = \d+, ([0-9.]+) s", r.stdout)
                out[key] = float(m.group(1)) if m else None
            except (OSError, subprocess.TimeoutExpired):
                out[key] = None
        return out

    def cpu_baseline(self, max_units):
        """The reference's own AVX2 getScores16 (oracle/_ref, kind 'reference') when its build travelled here,
        else the oracle restatement (kind 'port'); all host cores; bounded sample of the same workload."""
        from oracle import oracle_py as O
        cores = os.cpu_count() or 1
        n = min(self.batch.n, max_units or 2_000_000)
        sample = self.batch.slice(0, n)
        ref = O.ref_lib("bsw")
        if ref is not None and hasattr(ref, "ref_bsw_getscores16_mt"):
            out = np.zeros((n, 6), dtype=np.int32)
            secs, runs = C.c_double(0.0), []
            for _ in range(5):      # median of 5; the driver's own timed region (objects built outside it)
                ref.ref_bsw_getscores16_mt(*O._bsw_args(self.params, sample, out), C.c_int32(512), C.c_int32(cores),
                                           C.byref(secs))
                runs.append(secs.value)
            kind, dt = "reference", float(np.median(runs))
            what = "reference AVX2 getScores16 -b 512, one object per thread, median of 5"
            cols = [0, 1, 3, 5]     # score, tle, qle, max_off: the fields the AVX2 path defines like the scalar one
        else:
            runs = []
            for _ in range(5):
                t0 = time.perf_counter()
                out = O.bsw_oracle(self.params, sample, cores)
                runs.append(time.perf_counter() - t0)
            kind, dt, what = "port", float(np.median(runs)), "oracle/bsw_oracle.c scalar restatement, OpenMP, median of 5"
            cols = [0, 1, 2, 3, 4, 5]
        got = self.d.results()[:n]
        got = np.stack([got[f] for f in ("score", "tle", "gtle", "qle", "gscore", "max_off")], axis=1) if got.dtype.names else got
        same = bool(np.array_equal(np.asarray(got)[:, cols], np.asarray(out)[:, cols]))
        # in-band cells the algorithm actually visits (SURVEY 8d asks for them beside the nominal len1*len2): the
        # oracle counts them on a small sample
        m = min(n, 20000)
        _, inband = O.bsw_oracle(self.params, self.batch.slice(0, m), min(cores, 16), return_cells=True)
        inband_frac = inband / max(1.0, float(self.batch.slice(0, m).nominal_cells))
        return {"value": sample.nominal_cells / dt / 1e9, "unit": "GCUPS", "cores": cores, "kind": kind,
                "computed_over_nominal_cells": inband_frac,
                "sample": "first %d pairs of the rank-0 shard, %s, %.2f s" % (n, what, dt),
                "verified": "device results of these %d pairs %s the CPU run's (%d of 6 fields compared)"
                            % (n, "identical to" if same else "DIFFER from", len(cols))}


class ChainWork:
    metric, unit, dtype = "chain_large_gcups", "GCUPS", "int32+f64"

    large, seed = 10_000, 2001

    def __init__(self, args):
        self.n = args.size or self.large
        self.workload = ("chain large: %d synthetic minimap2 calls per GPU (seed 2001), resident in HBM; the step is bound by its longest "
                         "call (one serial recurrence: longest_job_ms)" % self.n)
        self.detail = {"cell": "evaluated predecessor pair (i, j) of chain_dp's inner loop"}

    def generate(self, first, n_units):
        from genomicsbench_amd.datagen import gen_chain
        return gen_chain(n_units, self.seed, first=first)

    def shards(self, full, parts):
        return S.chain_shards(*full, parts)

    to_arrays = staticmethod(lambda sh: S.chain_to_arrays(sh))
    n_units = staticmethod(lambda sh: len(sh[0]) - 1)

    def attach(self, tensors, dev, host_shard):
        from genomicsbench_amd.chain import DeviceChainBatch
        self.d = DeviceChainBatch.from_tensors(tensors, dev)
        self.case = host_shard
        self.units = None                                   # evaluated predecessor pairs: read from the device counter
        self.extra = {"calls_this_gpu": self.d.n_calls, "anchors_this_gpu": self.d.n_anchors}

    def output_tensor(self):
        import torch
        k = self.d.n_anchors
        return torch.stack([self.d.score[:k], self.d.parent[:k], self.d.target[:k], self.d.peak[:k]], dim=1)

    def check_gathered(self, full, ranges, parts, sample):
        from oracle import oracle_py as O
        off = full[0]
        bad = checked = 0
        for (lo, hi), got in zip(ranges, parts):
            m = min(hi - lo, sample)
            if m:
                sub = S.chain_shards(*full, 1, ranges=[(lo, lo + m)])[0]
                want = np.stack(O.chain_oracle(*sub, nthreads=min(os.cpu_count() or 1, 32)), axis=1)
                bad += int(not np.array_equal(got[:len(want)].cpu().numpy(), want))
                checked += len(want)
        return ("%d anchors (front of every shard) vs oracle, score/parent/target/peak: %s"
                % (checked, "identical" if not bad else "DIFFER"))

    def run(self, stream):
        self.d.run(stream)

    def finish(self, stream):
        self.units = float(self.d.evaluated_pairs(stream))
        self.extra["evaluated_pairs_per_gpu"] = int(self.units)
        jobs, longest = self.d.job_stats(stream)
        self.extra["jobs_this_gpu"], self.extra["longest_job_anchors"] = jobs, longest

    def longest_job(self, stream, steps=3):
        """The floor of the job: its longest call run ALONE (one wavefront: the recurrence over a call's anchors is serial).
        A chain job cannot end before this, on one GPU or on eight."""
        import torch
        from genomicsbench_amd.chain import DeviceChainBatch
        off, ax, ay, hdr = self.case
        c = int(np.argmax(np.diff(off)))
        a, b = int(off[c]), int(off[c + 1])
        d = DeviceChainBatch(np.array([0, b - a], dtype=np.int64), ax[a:b], ay[a:b], hdr[c:c + 1], self.d.off.device)
        d.run(stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            d.run(stream)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / steps
        return {"call": c, "anchors": b - a, "ms_alone": ms, "us_per_anchor": ms * 1e3 / max(b - a, 1)}

    def realistic(self, stream, steps=3):
        """The same call sizes with minimap2's structure inside a call (both strands, six reference ids in the upper x
        word, repeat copies, isolated hits: datagen.c gbx_gen_chain_fill_real) - the SURVEY 8d workload has one strand and
        one reference id, i.e. none of the points where a call falls apart into independent jobs.  Reported beside the
        'large' line, never as it; the first 40 calls are checked against the oracle."""
        import torch
        from genomicsbench_amd.chain import DeviceChainBatch
        from genomicsbench_amd.datagen import gen_chain
        from oracle import oracle_py as O
        case = gen_chain(self.d.n_calls, self.seed, realistic=True)
        d = DeviceChainBatch(*case, self.d.off.device)
        d.run(stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            d.run(stream)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / steps
        pairs = d.evaluated_pairs(stream)
        jobs, longest = d.job_stats(stream)
        results = d.results()
        # the same job with every call as one wavefront job (GBX_CHAIN_NOSPLIT, a test aid): what the cuts buy
        os.environ["GBX_CHAIN_NOSPLIT"] = "1"
        try:
            d.run(stream)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                d.run(stream)
            torch.cuda.synchronize()
            ms_whole = (time.perf_counter() - t0) * 1e3 / steps
            same_whole = all(np.array_equal(a, b) for a, b in zip(results, d.results()))
        finally:
            del os.environ["GBX_CHAIN_NOSPLIT"]
        off, ax, ay, hdr = case
        m = min(40, len(off) - 1)
        want = O.chain_oracle(off[:m + 1], ax[:off[m]], ay[:off[m]], hdr[:m], nthreads=min(os.cpu_count() or 1, 32))
        same = all(np.array_equal(g[:off[m]], w) for g, w in zip(results, want))
        return {"workload": "%d calls of the 'large' sizes, realistic structure (2 strands x 6 reference ids, 55 %% true locus / "
                            "25 %% repeat copies / 20 %% isolated hits)" % d.n_calls,
                "ms_per_step": ms, "value": pairs / (ms * 1e-3) / 1e9, "unit": self.unit, "anchors": int(d.n_anchors),
                "evaluated_pairs": int(pairs), "jobs": int(jobs), "longest_job_anchors": int(longest),
                "ms_per_step_one_job_per_call": ms_whole, "one_job_per_call_same_results": bool(same_whole),
                "verified": "first %d calls (%d anchors) vs oracle, score/parent/target/peak: %s"
                            % (m, int(off[m]), "identical" if same else "DIFFER")}

    def roofline_bytes(self, kernel):
        # 16 B anchor in + 4 x 4 B outputs (16 in + 8 out + 8 with targets / peaks exported)
        return 32 * self.d.n_anchors, self.units

    def host_entry(self):
        """PCIe-inclusive rate of the same shard through gbx_chain_host (pageable arrays in, the four result arrays out)."""
        from genomicsbench_amd import _native as N
        from genomicsbench_amd.chain import chain_host
        N.check(N.lib().gbx_host_prepare())
        n = int(self.case[0][-1])
        outs = tuple(np.full(n, -1, dtype=np.int32) for _ in range(4))          # exist and are touched, as a C driver's arrays are
        ms, got = _host_timed(lambda: chain_host(*self.case, out=outs))
        dev = self.d.results()
        return {"first_call_ms": ms[0], "best_ms": min(ms), "value": float(self.units or 0.0) / (min(ms) * 1e-3) / 1e9, "unit": self.unit,
                "what": "gbx_chain_host on the rank-0 shard: H2D + kernels + D2H, pageable memory on both sides (the four output arrays exist before the call)",
                "same_as_device_entry": bool(all(np.array_equal(a, b) for a, b in zip(got, dev)))}

    def cpu_baseline(self, max_units):
        from oracle import oracle_py as O
        cores = os.cpu_count() or 1
        off, ax, ay, hdr = self.case
        n = min(len(off) - 1, max_units or 2000)
        sub = (off[:n + 1], ax[:off[n]], ay[:off[n]], hdr[:n])
        ref = O.ref_lib("chain")
        t0 = time.perf_counter()
        if ref is not None:
            want = O.chain_ref(*sub, nthreads=cores)
            kind, what = "reference", "reference host_chain_kernel (chain_dp), OpenMP dynamic"
        else:
            want = O.chain_oracle(*sub, nthreads=cores)
            kind, what = "port", "oracle/chain_oracle.c, OpenMP dynamic"
        dt = time.perf_counter() - t0
        pairs = O.chain_oracle(*sub, nthreads=cores, return_pairs=True)[4]
        k = int(off[n])
        same = all(np.array_equal(g[:k], w[:k]) for g, w in zip(self.d.results(), want[:4]))
        return {"value": pairs / dt / 1e9, "unit": "GCUPS", "cores": cores, "kind": kind,
                "sample": "first %d calls of the rank-0 shard (%d evaluated pairs), %s, %.2f s" % (n, pairs, what, dt),
                "verified": "device score/parent/target/peak of these %d anchors %s the CPU run's"
                            % (k, "identical to" if same else "DIFFER from")}


class PhmmWork:
    metric, unit, dtype = "phmm_large_gcups", "GCUPS", "f32"

    large, seed = 20_000, 3001

    def __init__(self, args):
        self.n = args.size or self.large
        self.workload = "phmm large: %d synthetic GATK batches per GPU (seed 3001), inputs resident in HBM" % self.n
        self.detail = {"cell": "rslen*haplen per (read, haplotype) pair; fp32 with fp64 redo below 1e-28"}

    def generate(self, first, n_units):
        from genomicsbench_amd.datagen import gen_phmm
        return gen_phmm(n_units, self.seed, first=first)

    def shards(self, full, parts):
        return S.phmm_shards(full, parts)

    to_arrays = staticmethod(lambda sh: S.phmm_to_arrays(sh))
    n_units = staticmethod(lambda sh: len(sh.n_reads))

    def attach(self, tensors, dev, host_shard):
        from genomicsbench_amd.phmm import DevicePhmmBatchSet
        self.d = DevicePhmmBatchSet.from_tensors(tensors, dev)
        self.bs = host_shard
        d = self.d
        rl = d.read_len[d.pair_read[:d.n_pairs].long()].long()
        hl = d.hap_len[d.pair_hap[:d.n_pairs].long()].long()
        self.units = float((rl * hl).sum().item())
        self.extra = {"batches_this_gpu": int(tensors["n_reads"].numel()), "pairs_this_gpu": d.n_pairs,
                      "cells_this_gpu": int(self.units)}

    def output_tensor(self):
        return self.d.out[:self.d.n_pairs]

    def check_gathered(self, full, ranges, parts, sample):
        from oracle import oracle_py as O
        worst, checked, ok = 0.0, 0, True
        for (lo, hi), got in zip(ranges, parts):
            m = min(hi - lo, sample)
            if m:
                sub = full.take_batches(lo, lo + m)
                want = O.phmm_oracle(sub, min(os.cpu_count() or 1, 32))
                g = got[:sub.n_pairs].cpu().numpy()
                fin = np.isfinite(want)
                ok &= bool(np.array_equal(np.isfinite(g), fin))
                if fin.any():
                    worst = max(worst, float(np.max(np.abs(g[fin] - want[fin]) / np.maximum(np.abs(want[fin]), 1.0))))
                checked += sub.n_pairs
        return ("%d pairs (front of every shard) vs oracle: max rel err %.2e (bound 1e-5): %s"
                % (checked, worst, "within tolerance" if ok and worst <= 1e-5 else "DIFFER"))

    def run(self, stream):
        self.d.run(stream)

    def roofline_bytes(self, kernel):
        bs = self.bs
        lo, hi = PHMM_CLASS.get(kernel, (1, 1 << 30))
        rl, hl = bs.read_len[bs.pair_read].astype(np.int64), bs.hap_len[bs.pair_hap].astype(np.int64)
        sel = (rl >= lo) & (rl <= hi)
        return int((5 * rl[sel] + hl[sel] + 8).sum()), float((rl[sel] * hl[sel]).sum())

    def host_entry(self):
        """PCIe-inclusive rate of the same shard through gbx_phmm_forward_host."""
        from genomicsbench_amd import _native as N
        from genomicsbench_amd.phmm import forward_host
        N.check(N.lib().gbx_host_prepare())
        ms, got = _host_timed(lambda: forward_host(self.bs))
        dev = self.d.results()
        return {"first_call_ms": ms[0], "best_ms": min(ms), "value": float(self.units or 0.0) / (min(ms) * 1e-3) / 1e9, "unit": self.unit,
                "what": "gbx_phmm_forward_host on the rank-0 shard: H2D + kernels + D2H from pageable memory",
                "same_as_device_entry": bool(np.array_equal(got, dev))}

    def cpu_baseline(self, max_units):
        from oracle import oracle_py as O
        cores = os.cpu_count() or 1
        sub = self.bs.take_batches(0, min(len(self.bs.n_reads), max_units or 400))
        t0 = time.perf_counter()
        want = O.phmm_oracle(sub, cores)
        dt = time.perf_counter() - t0
        got = self.d.results()[:sub.n_pairs]                 # whole batches from the front: the same pairs, same order
        fin = np.isfinite(want)
        err = float(np.max(np.abs(got[fin] - want[fin]) / np.maximum(np.abs(want[fin]), 1.0))) if fin.any() else 0.0
        ok = err <= 1e-5 and np.array_equal(np.isfinite(got), fin)
        return {"value": sub.cells / dt / 1e9, "unit": "GCUPS", "cores": cores, "kind": "port",
                "sample": "first %d batches (%d pairs), oracle/phmm_oracle.c scalar fp32+fp64 redo, OpenMP, %.2f s "
                          "(GKL is an empty submodule: no reference build)" % (len(sub.n_reads), sub.n_pairs, dt),
                "verified": "device log10 likelihoods of these pairs %s the CPU run's (max rel err %.2e, bound 1e-5)"
                            % ("within tolerance of" if ok else "DIFFER from", err)}


class PoaWork:
    metric, unit, dtype = "poa_large_gcups", "GCUPS", "int16"

    large, seed = 6_000, 4001

    def __init__(self, args):
        from genomicsbench_amd.poa import make_params
        self.n = args.size or self.large
        self.params = make_params()
        self.workload = "poa large: %d synthetic 500-bp consensus windows per GPU (seed 4001), inputs resident in HBM" % self.n
        self.detail = {"cell": "graph node x sequence position per alignment"}

    def generate(self, first, n_units):
        from genomicsbench_amd.datagen import gen_poa
        return gen_poa(n_units, self.seed, first=first)

    def shards(self, full, parts):
        return S.poa_shards(full, parts)

    to_arrays = staticmethod(lambda sh: S.poa_to_arrays(sh))
    n_units = staticmethod(lambda sh: sh.n_windows)

    def attach(self, tensors, dev, host_shard):
        from genomicsbench_amd.poa import DevicePoaWindowSet
        self.d = DevicePoaWindowSet.from_tensors(tensors, dev)
        self.ws = host_shard
        self.units = None                                   # graph nodes x sequence length, device counter
        self.extra = {"windows_this_gpu": self.d.n_windows, "sequences_this_gpu": int(tensors["seq_len"].numel()),
                      "workspace_gb": round(self.d.work_bytes / 1e9, 2)}

    def output_tensor(self):
        """Fixed-stride records: int32 status, int32 length, consensus bytes."""
        import torch
        d = self.d
        n = d.n_windows
        head = torch.stack([d.status[:n], d.cons_len[:n]], dim=1).contiguous().view(torch.uint8)
        return torch.cat([head, d.cons[:n]], dim=1)

    @staticmethod
    def decode(rec):
        rec = rec.cpu().numpy()
        if not len(rec):
            return []
        head = np.ascontiguousarray(rec[:, :8]).view(np.int32)
        assert not head[:, 0].any(), "a window overflowed a device capacity"
        return [rec[w, 8:8 + head[w, 1]].tobytes().decode() for w in range(len(rec))]

    def check_gathered(self, full, ranges, parts, sample):
        from oracle import oracle_py as O
        bad = checked = 0
        for (lo, hi), got in zip(ranges, parts):
            m = min(hi - lo, sample)
            if m:
                want = O.poa_oracle(self.params, full.take(lo, lo + m), min(os.cpu_count() or 1, 32))
                bad += int(self.decode(got[:m]) != want)
                checked += m
        return "%d windows (front of every shard) vs oracle, consensus strings: %s" % (checked, "identical" if not bad else "DIFFER")

    def run(self, stream):
        self.d.run(self.params, stream)

    def finish(self, stream):
        self.units = float(self.d.cells(stream))
        self.extra["dp_cells_per_gpu"] = int(self.units)

    def roofline_bytes(self, kernel):
        # per cell: 4 B written (H int16 + one byte each of H-F and H-O: what the traceback and a far successor read) + 0.2 B
        # read (1.6 predecessor rows per row, 3 % of them further back than the six rows of the LDS ring); before the ring
        # (round 3) every predecessor row that was not the row before came from HBM: 8.8 B per cell
        return int(self.units * 4.2), self.units

    def host_entry(self):
        """PCIe-inclusive rate of the same shard through gbx_poa_consensus_host."""
        from genomicsbench_amd import _native as N
        from genomicsbench_amd.poa import consensus_host
        N.check(N.lib().gbx_host_prepare())
        ms, got = _host_timed(lambda: consensus_host(self.params, self.ws), reps=2)
        dev = self.d.results()
        return {"first_call_ms": ms[0], "best_ms": min(ms), "value": float(self.units or 0.0) / (min(ms) * 1e-3) / 1e9, "unit": self.unit,
                "what": "gbx_poa_consensus_host on the rank-0 shard: H2D + kernel + D2H from pageable memory (its own workspace: the "
                        "first call allocates it)",
                "same_as_device_entry": bool(list(got) == list(dev))}

    def cpu_baseline(self, max_units):
        from oracle import oracle_py as O
        cores = os.cpu_count() or 1
        sub = self.ws.take(0, min(self.ws.n_windows, max_units or 256))
        t0 = time.perf_counter()
        want, cells = O.poa_oracle(self.params, sub, cores, return_cells=True)
        dt = time.perf_counter() - t0
        same = self.d.results()[:sub.n_windows] == want
        return {"value": cells / dt / 1e9, "unit": "GCUPS", "cores": cores, "kind": "port",
                "sample": "first %d windows, oracle/poa_oracle.c scalar spoa restatement, OpenMP, %.2f s "
                          "(spoa is an empty submodule: no reference build)" % (sub.n_windows, dt),
                "verified": "device consensus strings of these %d windows %s the CPU run's"
                            % (sub.n_windows, "identical to" if same else "DIFFER from")}


class AbeaWork:
    """SURVEY 8f rank 4, widened in round 2: adaptive banded event alignment (R/benchmarks/abea/src/align.c:169-548)."""
    metric, unit, dtype = "abea_large_gcups", "GCUPS", "f32+f64"
    large, seed = 10000, 5001                           # the reference's large input is 10000 reads (R/scripts/run-gpu.sh:45)

    def __init__(self, args):
        self.n = args.size or self.large
        self.workload = "abea large: %d synthetic nanopore reads per GPU (seed 5001), inputs resident in HBM" % self.n
        self.detail = {"cell": "filled band cell (align.c `fills`)",
                       "generator": "synthetic 6-mer pore model, read lengths LogNormal(median 6000), 1.76 events per k-mer"}

    def generate(self, first, n_units):
        from genomicsbench_amd.datagen import gen_abea
        return gen_abea(n_units, self.seed, first=first)

    def shards(self, full, parts):
        return S.abea_shards(full, parts)

    to_arrays = staticmethod(lambda sh: S.abea_to_arrays(sh))
    n_units = staticmethod(lambda sh: sh.n_reads)

    def attach(self, tensors, dev, host_shard):
        from genomicsbench_amd.abea import DeviceAbeaReadSet
        self.d = DeviceAbeaReadSet.from_tensors(tensors, dev)
        self.rs = host_shard
        self.units = None                                   # filled band cells: device counter
        self.extra = {"reads_this_gpu": self.d.n_reads, "events_this_gpu": self.d.n_events_total, "bands_this_gpu": self.d.n_bands_total,
                      "workspace_gb": round(self.d.work_bytes / 1e9, 2)}

    def run(self, stream):
        self.d.run(stream)

    def finish(self, stream):
        self.units = float(self.d.cells(stream))
        self.extra["filled_cells_this_gpu"] = int(self.units)

    def output_tensor(self):
        """Fixed-stride records are not possible (pairs per read vary): the flat pair array followed by the counts."""
        import torch
        d = self.d
        return torch.cat([d.out.reshape(-1), d.n_pairs[:d.n_reads]])

    def check_gathered(self, full, ranges, parts, sample):
        from oracle import oracle_py as O
        bad = checked = 0
        for (lo, hi), got in zip(ranges, parts):
            m = min(hi - lo, sample)
            if m:
                sub_all = full.take(lo, hi)
                g = got.cpu().numpy()
                n_ev = int(sub_all.event_off[-1])
                pairs = g[:4 * max(n_ev, 1)].reshape(-1, 2)
                npairs = g[4 * max(n_ev, 1):]
                sub = sub_all.take(0, m)
                wo, wn = O.abea_oracle(sub, min(os.cpu_count() or 1, 32))
                ok = np.array_equal(npairs[:m], wn)
                for r in range(m):
                    a = 2 * int(sub.event_off[r])
                    ok = ok and np.array_equal(pairs[a:a + int(wn[r])], np.stack([wo["ref_pos"][a:a + int(wn[r])], wo["read_pos"][a:a + int(wn[r])]], axis=1))
                bad += int(not ok)
                checked += m
        return "%d reads (front of every shard) vs oracle, aligned pairs and QC verdicts: %s" % (checked, "identical" if not bad else "DIFFER")

    def roofline_bytes(self, kernel):
        # per band (100 cells): 64 B of back-pointers written and read once more by the traceback, + 1 B of move records
        # written and read; per k-mer the base and 16 B of scaled model parameters written and read; per event 4 B read
        # twice (range check, band loop); 8 B per aligned pair (~1 per event): ~1.6 B per cell
        d = self.d
        return int(d.n_bands_total * 130 + d.n_kmers_total * 33 + d.n_events_total * 16), self.units

    def host_entry(self):
        """PCIe-inclusive rate of the same shard through gbx_abea_align_host (pageable host arrays in - the events as the
        24-byte records a reference caller holds - aligned pairs out): reported beside `value`, never as it."""
        from genomicsbench_amd import _native as N
        from genomicsbench_amd.abea import PAIR_DTYPE
        rs = self.rs
        N.check(N.lib().gbx_host_prepare())
        ev = rs.events_struct()
        out = np.zeros(2 * max(int(rs.event_off[-1]), 1), dtype=PAIR_DTYPE)
        out["ref_pos"][:] = -1                                                # touched pages, like a caller's array
        n_pairs = np.zeros(max(rs.n_reads, 1), dtype=np.int32)
        ms = []
        for _ in range(4):
            t0 = time.perf_counter()
            N.check(N.lib().gbx_abea_align_host(rs.n_reads, N.ptr(rs.seq_off), N.ptr(rs.seq_len), N.ptr(rs.seq_arena), rs.seq_arena.size,
                                                N.ptr(rs.event_off), N.ptr(ev), N.ptr(rs.model), N.ptr(rs.scale), N.ptr(rs.shift),
                                                N.ptr(out), N.ptr(n_pairs)))
            ms.append((time.perf_counter() - t0) * 1e3)
        go, gn = self.d.results()
        same = np.array_equal(gn[:rs.n_reads], n_pairs[:rs.n_reads]) and all(
            np.array_equal(go[2 * int(rs.event_off[r]):2 * int(rs.event_off[r]) + int(gn[r])],
                           out[2 * int(rs.event_off[r]):2 * int(rs.event_off[r]) + int(gn[r])]) for r in range(0, rs.n_reads, max(rs.n_reads // 512, 1)))
        return {"first_call_ms": ms[0], "best_ms": min(ms), "value": float(self.units or 0.0) / (min(ms) * 1e-3) / 1e9, "unit": "GCUPS",
                "what": "gbx_abea_align_host on the rank-0 shard: H2D (%.2f GB of event records, means gathered on the way) + kernel "
                        "+ D2H (%.2f GB of pair slots) from pageable memory" % (ev.nbytes / 1e9, out.nbytes / 1e9),
                "same_as_device_entry": bool(same)}

    def cpu_baseline(self, max_units):
        from oracle import oracle_py as O
        cores = os.cpu_count() or 1
        sub = self.rs.take(0, min(self.rs.n_reads, max_units or 256))
        t0 = time.perf_counter()
        wo, wn, cells = O.abea_oracle(sub, cores, True)
        dt = time.perf_counter() - t0
        go, gn = self.d.results()
        same = np.array_equal(gn[:sub.n_reads], wn) and all(
            np.array_equal(go[2 * int(sub.event_off[r]):2 * int(sub.event_off[r]) + int(wn[r])],
                           wo[2 * int(sub.event_off[r]):2 * int(sub.event_off[r]) + int(wn[r])]) for r in range(sub.n_reads))
        return {"value": cells / dt / 1e9, "unit": "GCUPS", "cores": cores, "kind": "port",
                "sample": "first %d reads (%d filled cells), oracle/abea_oracle.c restating the CPU align(), OpenMP, %.2f s "
                          "(the reference translation unit needs htslib / HDF5 headers: no reference build)" % (sub.n_reads, cells, dt),
                "verified": "device pairs and QC verdicts of these %d reads %s the CPU run's" % (sub.n_reads, "identical to" if same else "DIFFER from")}


class FmiWork:
    """SURVEY 8f rank 4, second half: bwa-mem2 SMEM seeding on the FM-index (R/benchmarks/fmi/fmi.cpp:180-286).  The genome
    and its index are the same on every rank (built there on the GPU: the job of `bwa-mem2 index`, outside the timed
    region as in the reference, fmi.cpp:79-80); the reads are what is sharded."""
    metric, unit, dtype = "fmi_large_gext_per_s", "G backwardExt/s", "int64"
    large, seed = 10_000_000, 6001                      # the reference's large input is 10 M reads of 151 bases (R/scripts/run-cpu.sh:27)
    genome = 512 << 20                                  # 512 Mbp: the index covers both strands, 1 GB of checkpoints

    def __init__(self, args):
        self.n = args.size or self.large
        self.glen = int(os.environ.get("GBX_FMI_GENOME", self.genome))
        self.workload = "fmi large: %d synthetic 151-bp reads per GPU (seed 6001), %d-Mbp genome, index + reads resident in HBM" % (self.n, self.glen >> 20)
        self.detail = {"cell": "unit = backwardExt call (two checkpoint look-ups)",
                       "index": "FM-index of genome + reverse complement, %d MB of checkpoints, 4 %% repeat families, minSeedLen 19" % ((2 * self.glen + 1) >> 20)}
        self._g = None

    def genome_codes(self):
        if self._g is None:
            from genomicsbench_amd.datagen import gen_fmi_genome
            self._g = gen_fmi_genome(self.glen, self.seed)
        return self._g

    def generate(self, first, n_units):
        from genomicsbench_amd.datagen import gen_fmi_reads
        return gen_fmi_reads(self.genome_codes(), n_units, self.seed + 1, first=first)

    def shards(self, full, parts):
        return S.fmi_shards(full, parts)

    to_arrays = staticmethod(lambda sh: S.fmi_to_arrays(sh))
    n_units = staticmethod(lambda sh: sh.n_reads)

    def attach(self, tensors, dev, host_shard):
        import torch
        from genomicsbench_amd.fmi import DeviceFmi, build_index
        t0 = time.perf_counter()
        # ranks that share a GPU (GBX_BENCH_COMM=gloo test aid only) build their copies of the index one after the other: the
        # suffix sort's temporaries (~25 GB) times eight do not fit beside eight ranks' workspaces on one device
        share = _CTX.get("ranks_per_device", 1)
        for turn in range(share):
            if turn == _CTX.get("turn", 0):
                self.index = build_index(self.genome_codes(), device=dev)
                torch.cuda.synchronize()
                torch.cuda.empty_cache()
            if share > 1:
                _CTX["dist"].barrier()
        torch.cuda.synchronize()
        self.index_build_s = time.perf_counter() - t0
        self.d = DeviceFmi.from_tensors(self.index, (tensors["enc"], tensors["read_off"], tensors["read_len"]), dev)
        self.rs = host_shard
        self.units = None                                   # backwardExt calls: device counter
        self.extra = {"reads_this_gpu": self.d.n_reads, "index_mb": round(self.d.dindex.numel() / 1e6, 1),
                      "index_build_s_outside_timed_region": round(self.index_build_s, 2),
                      "workspace_gb": round(self.d.work_bytes / 1e9, 2)}

    def run(self, stream):
        self.d.run(stream)

    def finish(self, stream):
        self.units = float(self.d.extensions(stream))
        self.extra["extensions_this_gpu"] = int(self.units)
        self.extra["smems_this_gpu"] = int(self.d.n_out.item())
        # a truncated run must not be reported as a verified one: a read that outgrew its slot, or records beyond out_cap
        over, n_out = self.d.overflow(stream), int(self.d.n_out.item())
        self.truncated = bool(over) or n_out > self.d.out_cap
        if self.truncated:
            self.extra["TRUNCATED"] = "slot overflow %d, %d SMEMs for out_cap %d" % (over, n_out, self.d.out_cap)

    def output_tensor(self):
        """Variable-length result: [total | per-read offsets | records] as bytes."""
        import torch
        d = self.d
        n = min(int(d.n_out.item()), d.out_cap)
        return torch.cat([d.n_out.view(torch.uint8), d.smem_off.view(torch.uint8), d.out[:n * 40]])

    def check_gathered(self, full, ranges, parts, sample):
        from genomicsbench_amd.fmi import SMEM_DTYPE
        from oracle import oracle_py as O
        hidx = self.index.host()
        bad = checked = 0
        for (lo, hi), got in zip(ranges, parts):
            m = min(hi - lo, sample)
            if m:
                g = got.cpu().numpy()
                nrd = hi - lo
                off = g[8:8 + 8 * (nrd + 1)].view(np.int64)
                rec = g[8 + 8 * (nrd + 1):].view(SMEM_DTYPE)
                sub = full.take(lo, lo + m)
                wo, woff = O.fmi_oracle(hidx, sub, nthreads=min(os.cpu_count() or 1, 32))
                ok = np.array_equal(off[:m + 1], woff) and all(np.array_equal(rec[f][:len(wo)], wo[f]) for f in ("rid", "m", "n", "k", "l", "s"))
                bad += int(not ok)
                checked += m
        if getattr(self, "truncated", False):
            bad += 1
        return "%d reads (front of every shard) vs oracle, every field of every SMEM: %s" % (checked, "identical" if not bad else "DIFFER")

    def roofline_bytes(self, kernel):
        # a backwardExt reads two checkpoints of 64 bytes (rows k and k + s; often the same line late in a match, counted
        # twice here as the reference's GET_OCC pair does); a read's bases once; 40 bytes per SMEM written
        d = self.d
        if not kernel.startswith("fmi_smem"):
            return 0, 0.0
        return int((self.units or 0.0) * 128 + d.enc.numel() + int(d.n_out.item()) * 40), self.units

    def host_entry(self):
        """PCIe-inclusive rate through gbx_fmi_smem_host (pageable reads in, SMEM records out; the index is uploaded by
        the first call and kept on the device, as a caller that seeds batch after batch against one index needs it)."""
        from genomicsbench_amd import _native as N
        from genomicsbench_amd.fmi import smem_host
        N.check(N.lib().gbx_host_prepare())
        hidx = self.index.host()
        rs = self.rs.take(0, min(self.rs.n_reads, 1_000_000))
        ms = []
        for _ in range(3):
            t0 = time.perf_counter()
            got, goff = smem_host(hidx, rs, self.d.params)
            ms.append((time.perf_counter() - t0) * 1e3)
        dev, doff = self.d.results()
        k = int(doff[rs.n_reads])
        same = np.array_equal(goff, doff[:rs.n_reads + 1]) and all(np.array_equal(got[f], dev[f][:k]) for f in ("rid", "m", "n", "k", "l", "s"))
        frac = rs.n_reads / max(self.rs.n_reads, 1)
        return {"first_call_ms": ms[0], "best_ms": min(ms), "value": float(self.units or 0.0) * frac / (min(ms) * 1e-3) / 1e9, "unit": self.unit,
                "what": "gbx_fmi_smem_host on the first %d reads of the rank-0 shard: H2D of the reads + kernels + D2H of the records "
                        "from pageable memory; the first call also uploads the %d MB index" % (rs.n_reads, self.d.dindex.numel() >> 20),
                "same_as_device_entry": bool(same)}

    def cpu_baseline(self, max_units):
        from oracle import oracle_py as O
        cores = os.cpu_count() or 1
        sub = self.rs.take(0, min(self.rs.n_reads, max_units or 1_000_000))
        hidx = self.index.host()
        t0 = time.perf_counter()
        wo, woff, ext, rounds = O.fmi_oracle(hidx, sub, nthreads=cores, return_stats=True)
        dt = time.perf_counter() - t0
        dev, doff = self.d.results()
        k = int(doff[sub.n_reads])
        same = np.array_equal(woff, doff[:sub.n_reads + 1]) and all(np.array_equal(wo[f], dev[f][:k]) for f in ("rid", "m", "n", "k", "l", "s"))
        return {"value": ext / dt / 1e9, "unit": self.unit, "cores": cores, "kind": "port", "reads_per_s": sub.n_reads / dt,
                "sample": "first %d reads (%d backwardExt calls, %d / %d / %d SMEMs from the three rounds), oracle/fmi_oracle.c restating "
                          "bwa-mem2's FMI_search, OpenMP dynamic, %.2f s (tools/bwa-mem2 is an empty submodule: no reference build)"
                          % (sub.n_reads, ext, rounds[0], rounds[1], rounds[2], dt),
                "verified": "device SMEMs of these %d reads %s the CPU run's (every field)" % (sub.n_reads, "identical to" if same else "DIFFER from")}


WORKLOADS = {"bsw": BswWork, "chain": ChainWork, "phmm": PhmmWork, "poa": PoaWork, "abea": AbeaWork, "fmi": FmiWork}
_PROFILE_FILES = {"traffic": ("profiles/hbm_traffic.json", "bytes_per_launch"), "valu_busy": ("profiles/valu_busy.json", "valu_busy"),
                  "valu_insts": ("profiles/valu_insts.json", "valu_insts")}
_KIND_SOURCE = {"bsw": "bsw_kernels.hip", "chain": "chain_kernels.hip", "phmm": "phmm_kernels.hip", "poa": "poa_kernels.hip", "abea": "abea_kernels.hip",
                "fmi": "fmi_kernels.hip"}


def _hip_sha16(kind):
    """hash over every file of the kind's kernel translation unit (<kind>_kernels.hip + the headers it includes)"""
    from genomicsbench_amd.srchash import tu_sha16
    return tu_sha16(kind)


def _committed(kind, what, kernel_name, default_size):
    """HBM bytes per launch / VALU-busy fraction / VALU instructions of a kernel from the committed rocprofv3 --pmc passes (they
    cannot be collected inside the run that prints the line: PMC passes serialise the kernels); only for the default sizes,
    and only while the kernel source is the one the counters were collected on (scripts/make_profile_tables.py stamps the
    hash of the kind's .hip file into the table): after a kernel change the field is null until the passes are re-run."""
    rel, key = _PROFILE_FILES[what]
    path = os.path.join(ROOT, rel)
    if not default_size or not os.path.exists(path):
        return None, None
    table = json.load(open(path))
    if table.get("hip_sha16", {}).get(kind) != _hip_sha16(kind):
        return None, rel + " (stale: %s or a header it includes changed since the counters were collected)" % _KIND_SOURCE[kind]
    return table.get(key, {}).get(kernel_name), rel


def _valu_roof(kind, kernel_name):
    """Lane operations per second a chip issuing nothing but this kernel's VALU instruction mix would reach: the kernel's
    static mix (scripts/isa_mix.py) weighted with the measured per-form issue rates (scripts/valu_peak.hip ->
    profiles/valu_peak.json).  None when the mix table was made from another source."""
    path = os.path.join(ROOT, "profiles", "valu_mix.json")
    if not os.path.exists(path):
        return None, None
    t = json.load(open(path))
    if t.get("tu_sha16", {}).get(kind) != _hip_sha16(kind):
        return None, None
    # (the lane kernels' PACKED forms - fourth template argument true - run in the pipelined host entry only, not in the timed job)
    rows = [v for k, v in t["kernels"].items() if not re.match(r"bsw_lane_kernel<\w+, \w+, \w+, true>", k) and
            (v["stage"] == kernel_name or (kernel_name.startswith("bsw_lane_c") and v["stage"] == "bsw_lane_compact")
             or (kernel_name.startswith("bsw_lane_w") and v["stage"] == "bsw_lane_wide"))]
    if not rows:
        return None, None
    r = max(rows, key=lambda v: v["valu_static"])
    return r["roof_lane_ops_per_s"], r["mean_cycles_per_valu"]


def run_kernel(kind, args, ctx, steps, warmup, per_gpu_units=None, label=None):
    """One kernel's whole protocol on all ranks; returns the JSON object on rank 0 (None elsewhere)."""
    import torch
    from genomicsbench_amd import _native as N
    rank, world, dev, dist = ctx["rank"], ctx["world"], ctx["dev"], ctx["dist"]
    work = WORKLOADS[kind](args)
    n_per = per_gpu_units if per_gpu_units is not None else work.n
    stream = torch.cuda.current_stream().cuda_stream

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- inputs into HBM
    full = ranges = None
    t_gen = time.perf_counter()
    if args.mode == "local":
        mine_host = work.generate(rank * n_per, n_per)
        barrier()
        t0 = time.perf_counter()
        buf, meta = S.pack_arrays(work.to_arrays(mine_host))
        tensors = S.unpack_tensor(torch.from_numpy(buf).to(dev), meta)
        barrier()
        scatter_ms = (time.perf_counter() - t0) * 1e3
    else:
        per_rank = mine_host = None
        if rank == 0:
            full = work.generate(0, n_per * world)
            shards = work.shards(full, world)
            sizes = [work.n_units(sh) for sh in shards]
            ranges = [(sum(sizes[:r]), sum(sizes[:r + 1])) for r in range(world)]
            per_rank = [work.to_arrays(sh) for sh in shards]
            mine_host = shards[0]
        gen_s = time.perf_counter() - t_gen
        barrier()
        t0 = time.perf_counter()
        if world > 1:
            tensors, _ = S.scatter_arrays(per_rank, device=ctx["comm_dev"])
            tensors = {k: v.to(dev) for k, v in tensors.items()}         # no-op over RCCL (already in HBM)
        else:
            buf, meta = S.pack_arrays(per_rank[0])
            tensors = S.unpack_tensor(torch.from_numpy(buf).to(dev), meta)
        barrier()
        scatter_ms = (time.perf_counter() - t0) * 1e3
        per_rank = None
    work.attach(tensors, dev, mine_host)

    # ---- timed steps
    for _ in range(warmup):
        work.run(stream)
    barrier()
    N.profile_begin()
    t0 = time.perf_counter()
    for _ in range(steps):
        work.run(stream)
    barrier()
    dt = time.perf_counter() - t0
    stages = N.profile_end()
    if hasattr(work, "finish"):
        work.finish(stream)

    t = torch.tensor([dt], dtype=torch.float64, device=ctx["comm_dev"])
    units = torch.tensor([work.units], dtype=torch.float64, device=ctx["comm_dev"])
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(units, op=dist.ReduceOp.SUM)
    dt_max, total_units = float(t.item()), float(units.item())

    # ---- outputs back to rank 0
    out_t = work.output_tensor().contiguous()
    barrier()
    t0 = time.perf_counter()
    if world > 1 and args.mode != "local":
        parts = S.gather_array(out_t.to(ctx["comm_dev"]))
        if rank == 0:
            parts = [p.cpu() for p in parts]
    else:
        parts = [out_t.cpu()]
    barrier()
    gather_ms = (time.perf_counter() - t0) * 1e3
    if rank != 0:
        return None

    # ---- rank 0: the line
    # the dominant kernel = the longest-running launch among those that had a real share of the work (a launch whose class is empty
    # still waits for its LDS; phmm's smaller classes live as long as the dominant one - they run in what it leaves free - with a
    # fiftieth of its cells): at least a quarter of the largest launch's units
    umax = max((work.roofline_bytes(n_)[1] or 0) for n_ in stages) if stages else 0
    for name, (ms_sum, launches) in sorted(stages.items(), key=lambda kv: -kv[1][0]):
        alg_bytes, k_units = work.roofline_bytes(name)
        if k_units and k_units >= 0.25 * umax:
            break
    k_ms = ms_sum / max(launches, 1)
    # a job that runs as several launches of the same kernel per step (fmi: chunks of reads): bytes and units per launch
    lps = launches / max(steps, 1)
    if lps > 1.5:
        alg_bytes, k_units = int(alg_bytes / lps), k_units / lps
    achieved = alg_bytes / (k_ms * 1e-3) / 1e9
    default_size = not args.size and per_gpu_units is None and world == 1
    traffic, tsrc = _committed(kind, "traffic", name, default_size)
    valu_b, vsrc = _committed(kind, "valu_busy", name, default_size)
    valu = None
    vi, visrc = _committed(kind, "valu_insts", name, default_size)
    roof, mean_cyc = _valu_roof(kind, name)
    if vi and k_units:
        # VALU issue of the dominant kernel: committed SQ_INSTS_VALU per launch x 64 lanes over the units it processes and over
        # its live duration in this run, against the measured roof of its own instruction mix (profiles/valu_peak.json: a
        # wave64 v_add / v_and / v_mov issues in 2.25 SIMD cycles, v_max / v_max3 / compares / selects / DPP / SDWA / packed
        # forms in 4.17): bsw / phmm / abea are bound by this, not by HBM
        lane_ops = vi * 64.0
        valu = {"lane_ops_per_unit": lane_ops / k_units, "lane_ops_per_s": lane_ops / (k_ms * 1e-3), "insts_source": visrc,
                "roof_lane_ops_per_s": roof, "mean_cycles_per_valu_instruction": mean_cyc,
                "roof_source": "profiles/valu_mix.json x profiles/valu_peak.json" if roof else None}
        if roof:
            valu["frac"] = min(1.0, lane_ops / (k_ms * 1e-3) / roof)
        if kind == "bsw":
            allk = json.load(open(os.path.join(ROOT, "profiles/valu_insts.json")))["valu_insts"]
            tot = 64.0 * sum(v for kk, v in allk.items() if kk.startswith("bsw_"))
            # the class kernels overlap on four streams, so the job-level rate is the meaningful one: every bsw kernel's lane
            # operations over the step time
            valu["job_lane_ops_per_nominal_cell"] = tot / work.units
            valu["job_lane_ops_per_s"] = tot / (dt_max / steps)
            if roof:
                valu["job_frac"] = min(1.0, valu["job_lane_ops_per_s"] / roof)
    # (every string under 120 characters: the driver's record of the line cuts longer ones)
    cfg = {"workload": label or work.workload, "mode": args.mode if world > 1 else "single",
           "parallelism": "units sharded over %d rank(s), contiguous cost-balanced ranges, no data-path collective" % world,
           "inputs": ("rank 0 scatters packed shards / gathers outputs over %s p2p" % ctx["comm"]) if args.mode != "local"
                     else "every rank generates its own shard"}
    cfg.update(getattr(work, "detail", {}))
    cfg.update(work.extra)
    # job-level figures, flat (the driver's record keeps only flat keys of `roofline`): every stage of a step with work in it
    # (the class kernels of bsw overlap on four streams, so per-launch fractions of overlapping launches say little; these
    # are the stable ones) - algorithmic bytes of all stages over the step time, all VALU lane operations over the step time
    if kind == "bsw":
        b = work.batch                                       # every pair once, whichever launch took it: len1 + len2 + 36 bytes
        job_bytes = int(b.len1.astype(np.int64).sum() + b.len2.astype(np.int64).sum() + 36 * b.n)
    else:
        job_bytes = work.roofline_bytes(name)[0]            # the other kinds' figure is the whole job's already
    step_s = dt_max / steps
    flat = {"job_bytes_per_s": job_bytes / step_s, "job_hbm_frac": job_bytes / step_s / 1e9 / HBM_PEAK_GBS,
            "valu_frac": (valu or {}).get("frac"), "job_valu_frac": (valu or {}).get("job_frac"),
            "job_lane_ops_per_s": (valu or {}).get("job_lane_ops_per_s"), "valu_roof_lane_ops_per_s": roof}
    line = {
        "metric": work.metric, "value": total_units * steps / dt_max / 1e9, "unit": work.unit,
        "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": dt_max / steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": work.dtype, "data": "synthetic", "config": cfg,
        "roofline": {"bound": "hbm", "kernel": name, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": tsrc,
                     "valu_busy": valu_b, "valu_busy_source": vsrc, "valu": valu, "kernel_ms": k_ms,
                     "algorithmic_bytes_per_launch": alg_bytes,
                     "cells_per_s_dominant_kernel": (k_units or 0.0) / (k_ms * 1e-3), **flat},
        "kernels_ms": {k: v[0] / max(v[1], 1) for k, v in sorted(stages.items())},
        "scatter_ms": scatter_ms, "gather_ms": gather_ms, "rccl_ranks": world, "comm": ctx["comm"],
    }
    if kind == "fmi":
        # fmi's algorithmic bytes are line REQUESTS (two 64-byte lines per backwardExt), 60 % of which hit L2: the HBM fraction is
        # the counters' (FETCH_SIZE + WRITE_SIZE per launch, profiles/hbm_traffic.json), the request rate is kept beside it
        rf = line["roofline"]
        rf["l2_request_gbs"], rf["l2_request_frac"] = rf["achieved"], rf["frac"]
        if traffic:
            rf["achieved"] = traffic / (k_ms * 1e-3) / 1e9
            rf["frac"] = rf["achieved"] / HBM_PEAK_GBS
            rf["achieved_source"] = "HBM counters (traffic) over the kernel's live duration; l2_request_* = algorithmic line requests"
        else:
            rf["achieved_source"] = "line requests (no counter table for these sources): an upper bound on HBM bytes"
    if args.mode != "local":
        line["dataset_gen_s"] = gen_s
        line["gather_verified"] = work.check_gathered(full, ranges, parts, args.verify_units or
                                                      {"bsw": 20000, "chain": 40, "phmm": 20, "poa": 8, "abea": 8, "fmi": 20000}[kind])
        line["shard_units"] = [hi - lo for lo, hi in ranges]
    if not args.no_cpu and world == 1:                      # the CPU baseline is an N=1 figure (rank 0's host cores)
        # the host-buffer entry first, on quiet host cores (the CPU baseline runs OpenMP teams on all of them)
        if hasattr(work, "host_entry"):
            line["host_entry"] = work.host_entry()
            # SURVEY 8d's three timing legs, flat: (i) device-resident = `value`, (ii) H2D + kernels + D2H through the host entry,
            # (iii) end to end including the parse (bsw: the driver on the reference's input format)
            line["host_entry_ms"] = line["host_entry"]["best_ms"]
            line["host_entry_" + ("gcups" if work.unit == "GCUPS" else "value")] = line["host_entry"]["value"]
        if hasattr(work, "e2e"):
            line["e2e"] = work.e2e()
            line["e2e_ms"] = line["e2e"].get("e2e_ms")
        if hasattr(work, "realistic"):
            line["realistic"] = work.realistic(stream)
        if hasattr(work, "longest_job"):
            line["longest_job"] = work.longest_job(stream)
            line["longest_job_ms"] = line["longest_job"]["ms_alone"]
        line["cpu_baseline"] = work.cpu_baseline(args.cpu_units)
        line["cpu_baseline"]["cgroup_cpu_quota_cores"] = _cpu_quota()      # `cores` = threads used; this is what they could run on
    return line


def predict_shards(kind, args, ctx, parts, steps, warmup, whole_ms=None):
    """BASELINE config 4 on ONE GPU: ONE 'large' job cut into `parts` shards exactly as the N-GPU strong-scaling leg
    (`config4_strong`) cuts it, every shard run ALONE on this GPU through the same device entry (upload = the scatter's
    stand-in, timed steps, download, front of the shard checked against the oracle).  A rank of the N-GPU run holds one
    shard on a GPU of its own, so the slowest shard alone is what the N-GPU step would take (no data-path collective;
    scatter / gather are outside the timed region there as here): predicted_ms_per_step = max over shards,
    predicted_speedup = the whole job on this GPU / that."""
    import torch
    from genomicsbench_amd import _native as N
    dev = ctx["dev"]
    stream = torch.cuda.current_stream().cuda_stream
    sub = argparse.Namespace(**vars(args))
    sub.size = 0 if whole_ms is not None else args.size      # (--size shrinks the job of a stand-alone --predict-shards run: tests)
    probe = WORKLOADS[kind](sub)
    full = probe.generate(0, probe.n)
    verify = args.verify_units or {"bsw": 20000, "chain": 40, "phmm": 20, "poa": 8, "abea": 8, "fmi": 20000}[kind]

    def one(host_shard, lo, hi):
        w = WORKLOADS[kind](sub)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        buf, meta = S.pack_arrays(w.to_arrays(host_shard))
        tensors = S.unpack_tensor(torch.from_numpy(buf).to(dev), meta)
        torch.cuda.synchronize()
        up_ms = (time.perf_counter() - t0) * 1e3
        w.attach(tensors, dev, host_shard)
        for _ in range(warmup):
            w.run(stream)
        torch.cuda.synchronize()
        N.profile_begin()
        t0 = time.perf_counter()
        for _ in range(steps):
            w.run(stream)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / steps
        stages = N.profile_end()
        if hasattr(w, "finish"):
            w.finish(stream)
        t0 = time.perf_counter()
        out = w.output_tensor().contiguous().cpu()
        down_ms = (time.perf_counter() - t0) * 1e3
        rec = {"units": hi - lo, "ms_per_step": ms, "upload_ms": up_ms, "download_ms": down_ms, "work_units": w.units,
               "kernels_ms": {k: v[0] / max(v[1], 1) for k, v in sorted(stages.items(), key=lambda kv: -kv[1][0])[:3]}}
        rec.update({k: v for k, v in w.extra.items() if k in ("longest_job_anchors", "jobs_this_gpu")})
        if verify:
            rec["verified"] = w.check_gathered(full, [(lo, hi)], [out], verify)
        del w, tensors, out
        torch.cuda.empty_cache()
        return rec

    if whole_ms is None:
        n_all = probe.n_units(probe.shards(full, 1)[0])
        whole = one(probe.shards(full, 1)[0], 0, n_all)
        whole_ms = whole["ms_per_step"]
    shards = probe.shards(full, parts)
    sizes = [probe.n_units(sh) for sh in shards]
    recs, lo = [], 0
    for sh, n in zip(shards, sizes):
        recs.append(one(sh, lo, lo + n))
        lo += n
    worst = max(range(parts), key=lambda k: recs[k]["ms_per_step"])
    pred = recs[worst]["ms_per_step"]
    total_units = sum(r["work_units"] or 0.0 for r in recs)
    ok = all("DIFFER" not in r.get("verified", "") for r in recs)
    return {"what": "ONE %s 'large' job cut into %d shards as config4_strong cuts it; every shard alone on this GPU" % (kind, parts),
            "parts": parts, "steps": steps, "shard_units": sizes, "shard_ms": [round(r["ms_per_step"], 3) for r in recs],
            "whole_job_ms_1gpu": whole_ms, "predicted_ms_per_step": pred, "predicted_speedup": whole_ms / pred,
            "predicted_value": total_units / (pred * 1e-3) / 1e9, "unit": probe.unit, "slowest_shard": worst,
            "slowest_shard_kernels_ms": recs[worst]["kernels_ms"],
            "scatter_standin_ms_max": max(r["upload_ms"] for r in recs), "gather_standin_ms_max": max(r["download_ms"] for r in recs),
            "verified": ("front %d units of every shard vs oracle: " % verify) + ("identical" if ok else "DIFFER"),
            **({"longest_job_anchors": max(r.get("longest_job_anchors", 0) for r in recs)} if kind == "chain" else {})}


def launch_plan(args, argv):
    """argv + environment of every rank a plain `python bench.py --gpus N` starts (one process per GPU, rendezvous on
    127.0.0.1, the same variables torch.distributed.run sets).  Pure: no torch, no GPU."""
    import socket
    port = args.master_port
    if not port:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
    child_argv = [a for a in argv if a != "--launch-dry-run"]
    base = {"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "WORLD_SIZE": str(args.gpus),
            "LOCAL_WORLD_SIZE": str(args.gpus), "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")}
    # rank 0 generates and cuts the whole dataset (OpenMP generator) and runs the checks: it gets the host's cores
    cores = os.cpu_count() or 1
    return [{"argv": [sys.executable, os.path.abspath(__file__)] + child_argv,
             "env": dict(base, RANK=str(r), LOCAL_RANK=str(r), GROUP_RANK="0",
                         OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", str(cores if r == 0 else 4)))}
            for r in range(args.gpus)]


def self_launch(args, argv):
    """Parent of a self-launched run: starts the ranks as children, relays rank 0's stdout (the JSON line) and every
    rank's stderr, returns the worst child return code; when one rank fails the others are terminated (by PID)."""
    import subprocess
    plan = launch_plan(args, argv)
    if args.launch_dry_run:
        print(json.dumps({"launch": plan}), flush=True)
        return 0
    import threading
    procs, relays = [], []

    def relay(r, pipe):                                     # every rank's stderr, line by line, with its rank in front
        for ln in iter(pipe.readline, b""):
            sys.stderr.buffer.write(b"[rank %d] " % r + ln)
            sys.stderr.buffer.flush()
        pipe.close()

    for r, p in enumerate(plan):
        env = dict(os.environ)
        env.update(p["env"])
        pr = subprocess.Popen(p["argv"], env=env, stdout=None if r == 0 else subprocess.DEVNULL, stderr=subprocess.PIPE)
        procs.append(pr)
        t = threading.Thread(target=relay, args=(r, pr.stderr), daemon=True)
        t.start()
        relays.append(t)
    rc, live = 0, list(procs)
    while live:
        time.sleep(0.2)
        for pr in list(live):
            code = pr.poll()
            if code is None:
                continue
            live.remove(pr)
            if code != 0:
                rc = rc or (code if code > 0 else 128 - code)
                for other in live:                         # a dead rank leaves the others in a collective: stop them
                    other.terminate()
    for t in relays:
        t.join(timeout=5)
    return rc


def _pg_timeout():
    """A rank that never arrives must fail the run, not hang it: every collective and rendezvous of the process group gives up
    after GBX_BENCH_TIMEOUT_S (default 900 s: rank 0 generates and cuts N x 'large' before the first scatter)."""
    import datetime
    return datetime.timedelta(seconds=float(os.environ.get("GBX_BENCH_TIMEOUT_S", "900")))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--kernel", choices=sorted(WORKLOADS), default=None,
                    help="one kernel only (default: bsw headline + chain, phmm, poa, abea, fmi under 'kernels')")
    ap.add_argument("--mode", choices=["scatter", "local"], default="scatter")
    ap.add_argument("--size", type=int, default=0, help="units per GPU (pairs / calls / batches / windows / reads); 0 = 'large'")
    ap.add_argument("--pairs", type=int, default=0, help="alias of --size for bsw")
    ap.add_argument("--cpu-units", type=int, default=0, help="units in the CPU-baseline sample (0 = kernel default)")
    ap.add_argument("--verify-units", type=int, default=0, help="units per shard checked against the oracle after the gather")
    ap.add_argument("--other-steps", type=int, default=5, help="timed steps of chain / phmm / poa / abea in the all-kernel run")
    ap.add_argument("--full-kernels", action="store_true", help="the other kernels' records in full inside the line (default: short form in the line, "
                    "full records on stderr)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--predict-shards", type=int, default=0,
                    help="N=1 only, with --kernel K: cut ONE 'large' job into this many shards as the N-GPU strong-scaling leg does, run "
                         "every shard alone, print per-shard ms and the predicted N-GPU step (BASELINE config 4 is poa over 8)")
    ap.add_argument("--no-predict", action="store_true", help="skip poa's config4_predicted leg of the all-kernel run")
    ap.add_argument("--launch-dry-run", action="store_true",
                    help="print the argv / environment of the ranks `--gpus N` would start, start nothing")
    ap.add_argument("--launch-echo", action="store_true",
                    help="launcher self-test without GPUs: the ranks rendezvous over gloo, all-reduce their ranks, rank 0 "
                         "prints one JSON line (tests/test_bench_launch.py); --launch-echo-fail R makes rank R exit 7")
    ap.add_argument("--launch-echo-fail", type=int, default=-1)
    ap.add_argument("--master-port", type=int, default=0, help="rendezvous port of the self-launched ranks (0 = a free one)")
    args = ap.parse_args()
    if args.pairs:
        args.size = args.pairs

    # ---- self-launch: `python bench.py --gpus N` with no rank environment starts its own N ranks.  This happens BEFORE
    # torch.cuda / libgbx are imported and before any GPU call: the parent never touches the GPU, the ranks are child
    # processes (never an exec of a process that initialised HIP), rank 0's JSON line is relayed and the children's worst
    # return code is ours.
    if "RANK" not in os.environ and (args.gpus > 1 or args.launch_dry_run):
        sys.exit(self_launch(args, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: WORLD_SIZE=%d but --gpus %d (launch with --nproc-per-node %d, or run plain "
                         "`python bench.py --gpus %d`, which starts its own ranks)" % (world, args.gpus, args.gpus, args.gpus))

    if rank == 0 and world > 1 and os.environ.get("OMP_NUM_THREADS") == "1":
        # torch.distributed.run pins every rank to one OpenMP thread; rank 0 generates N x 'large' with the OpenMP
        # generator and runs the oracle checks, outside every timed region: give it the cores back (before libgomp loads)
        os.environ["OMP_NUM_THREADS"] = str(os.cpu_count() or 1)
    import torch
    import torch.distributed as dist
    global S
    from genomicsbench_amd import shard as S

    if args.launch_echo:
        if rank == args.launch_echo_fail:
            print("launch-echo: rank %d was told to fail" % rank, file=sys.stderr, flush=True)
            sys.exit(7)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL when every rank has a device of its own (the first contact of init_process_group("nccl", device_id=...) and a
        # grouped send / receive with this launcher), gloo otherwise (the launcher alone, no GPU needed)
        backend = "nccl" if torch.cuda.device_count() >= world and os.environ.get("GBX_BENCH_COMM", "nccl") != "gloo" else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local), timeout=_pg_timeout())
            edev = torch.device("cuda", local)
        else:
            dist.init_process_group("gloo", timeout=_pg_timeout())
            edev = torch.device("cpu")
        t = torch.tensor([float(rank)], dtype=torch.float64, device=edev)
        dist.all_reduce(t)
        # the scatter's shape in small: rank 0 sends every rank a buffer in several capped pieces, gets it back
        from genomicsbench_amd import shard as SH
        payload = [{"x": np.arange(50000 + 1000 * r, dtype=np.int32), "y": np.full(3000, r, dtype=np.uint8)} for r in range(world)]
        os.environ.setdefault("GBX_SHARD_MSG_BYTES", "65536")
        mine, _ = SH.scatter_arrays(payload if rank == 0 else None, device=edev)
        back = SH.gather_array(mine["x"])
        echo_ok = rank != 0 or all(np.array_equal(b.cpu().numpy(), payload[r]["x"]) for r, b in enumerate(back))
        if rank == 0:
            print(json.dumps({"launch_echo": True, "world": world, "rank_sum": float(t.item()), "backend": backend,
                              "scatter_gather_ok": bool(echo_ok), "omp_num_threads": os.environ.get("OMP_NUM_THREADS")}), flush=True)
        dist.barrier()
        dist.destroy_process_group()
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (libgbx has no CPU path)")
    # GBX_BENCH_COMM=gloo is a TEST AID for boxes with fewer GPUs than ranks (the builder's 1-GPU box): ranks share
    # the GPUs round-robin and the scatter / gather travel through host memory.  The judged path is RCCL.
    comm = os.environ.get("GBX_BENCH_COMM", "nccl")
    local = local % torch.cuda.device_count() if comm == "gloo" else local
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if comm == "gloo":
            dist.init_process_group("gloo", timeout=_pg_timeout())
        else:
            dist.init_process_group("nccl", device_id=dev, timeout=_pg_timeout())
    ctx = {"rank": rank, "world": world, "dev": dev, "dist": dist, "comm_dev": torch.device("cpu") if comm == "gloo" else dev,
           "comm": "rccl" if comm != "gloo" else "gloo (test aid)"}
    if comm == "gloo" and world > 1:
        ndev = torch.cuda.device_count()
        _CTX.update(dist=dist, ranks_per_device=-(-world // ndev), turn=rank // ndev)

    if args.kernel and args.predict_shards:
        if world != 1:
            raise SystemExit("bench.py: --predict-shards is a one-GPU measurement")
        line = {"metric": WORKLOADS[args.kernel].metric + "_predicted_%dgpu" % args.predict_shards, "n_gpus": 1, "scaling": "strong",
                "config4_predicted": predict_shards(args.kernel, args, ctx, args.predict_shards, args.steps, args.warmup)}
    elif args.kernel:
        line = run_kernel(args.kernel, args, ctx, args.steps, args.warmup)
    else:
        line = run_kernel("bsw", args, ctx, args.steps, args.warmup)
        others = {}
        for kind in ("chain", "phmm", "poa", "abea", "fmi"):
            torch.cuda.empty_cache()
            sub = argparse.Namespace(**vars(args))
            sub.size, sub.cpu_units = 0, 0                   # --size / --cpu-units speak about the headline kernel
            others[kind] = run_kernel(kind, sub, ctx, min(args.steps, args.other_steps), min(args.warmup, 2))
        # BASELINE config 4 as written: ONE 'large' poa job (6000 windows) sharded over the N GPUs (strong)
        if world > 1 and args.mode != "local":
            torch.cuda.empty_cache()
            sub = argparse.Namespace(**vars(args))
            sub.size, sub.cpu_units, sub.no_cpu = 0, 0, True
            st = run_kernel("poa", sub, ctx, min(args.steps, args.other_steps), min(args.warmup, 2),
                            per_gpu_units=-(-PoaWork.large // world),
                            label="poa large: ONE job of %d windows sharded over %d GPUs by scatter / gather (BASELINE config 4)"
                                  % (-(-PoaWork.large // world) * world, world))
            if st is not None:
                st["scaling"] = "strong"
                others["poa"]["config4_strong"] = st
        # ... and at N = 1 its prediction: the eight shards of that cut, each alone on this GPU (the slowest one is the 8-GPU step)
        if world == 1 and args.mode != "local" and not args.size and not args.no_predict:
            torch.cuda.empty_cache()
            others["poa"]["config4_predicted"] = predict_shards("poa", args, ctx, 8, 2, 1, whole_ms=others["poa"]["ms_per_step"])
        if line is not None:
            flat_config(line, others)
            # the headline line stays as measured; the five other kernels ride along in short form (what a reader of the
            # line's tail needs: value, time, roofline figures, what was checked) and in full on stderr
            if rank == 0 and not args.full_kernels:
                print("kernels in full: " + json.dumps(others), file=sys.stderr, flush=True)
            keep = ("shard_units", "scatter_ms", "gather_ms", "rccl_ranks", "comm") if world > 1 else ()      # the sharding's own figures
            line["kernels"] = others if args.full_kernels else {k: compact_entry(v, keep=keep) for k, v in others.items()}
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _verified(rec):
    """True / False from a record's check strings (gather check, CPU-baseline check, host entry), None when nothing was checked."""
    texts = [rec.get("gather_verified"), (rec.get("cpu_baseline") or {}).get("verified")]
    texts = [t for t in texts if t]
    he = (rec.get("host_entry") or {}).get("same_as_device_entry")
    if not texts and he is None:
        return None
    return all("DIFFER" not in t for t in texts) and he is not False


def flat_config(line, others):
    """The numbers that answer BASELINE's configs 2-5, as flat keys of `config` (the driver's record of the line keeps `config`,
    the flat keys of `roofline` and `cpu_baseline` - nothing nested): SURVEY 8d's timing legs of the headline, and per
    kernel its value, step time, roofline fractions, host-entry time, CPU figure and whether every check passed."""
    cfg = line["config"]
    cfg["timing_leg"] = "i: device-resident (value); ii: host_entry_* (H2D + kernels + D2H); iii: e2e_ms (text parse + ii)"
    he = line.get("host_entry") or {}
    cfg.update({"host_entry_ms": he.get("best_ms"), "host_entry_median_ms": he.get("median_ms"), "host_entry_gcups": he.get("value"),
                "e2e_ms": line.get("e2e_ms"), "e2e_ms_cached_input": (line.get("e2e") or {}).get("e2e_ms_cached_input"), "bsw_verified": _verified(line)})
    rd = (line.get("e2e") or {}).get("refdriver") or {}
    for k in ("t64_b512_s", "t16_b512_s", "t1_one_call_s"):
        if rd.get(k) is not None:
            cfg["refdriver_" + k] = rd[k]
    names = {"chain": "gpairs_per_s", "phmm": "gcups", "poa": "gcups", "abea": "gcups", "fmi": "gext_per_s"}
    for kind, rec in others.items():
        if not rec:
            continue
        rf, cb, h = rec.get("roofline") or {}, rec.get("cpu_baseline") or {}, rec.get("host_entry") or {}
        cfg.update({"%s_large_%s" % (kind, names[kind]): rec.get("value"), "%s_ms" % kind: rec.get("ms_per_step"),
                    "%s_hbm_frac" % kind: rf.get("frac"), "%s_valu_busy" % kind: rf.get("valu_busy"),
                    "%s_valu_frac" % kind: rf.get("valu_frac"), "%s_host_entry_ms" % kind: h.get("best_ms"),
                    "%s_cpu_value" % kind: cb.get("value"), "%s_verified" % kind: _verified(rec)})
        if kind == "chain":
            cfg["chain_realistic_ms"] = (rec.get("realistic") or {}).get("ms_per_step")
            cfg["chain_longest_job_ms"] = rec.get("longest_job_ms")
        if kind == "poa":
            cfg["poa_config4_predicted_speedup"] = (rec.get("config4_predicted") or {}).get("predicted_speedup")
        if kind == "fmi":
            cfg["fmi_l2_request_frac"] = rf.get("l2_request_frac")
    for k in [k for k, v in cfg.items() if v is None]:
        del cfg[k]


_COMPACT_DROP = {"n_gpus", "warmup", "higher_is_better", "scaling", "vs_baseline", "data", "rccl_ranks", "comm", "parallelism", "inputs", "mode",
                 "traffic_source", "valu_busy_source", "dataset_gen_s", "shard_units", "scatter_ms", "gather_ms", "first_call_ms", "what", "cell",
                 "algorithmic_bytes_per_launch", "cells_per_s_dominant_kernel", "job_lane_ops_per_s", "valu_roof_lane_ops_per_s", "valu"}


def compact_entry(v, key=None, depth=0, keep=()):
    """Short form of a per-kernel record for the all-kernels line: no nulls, no keys that repeat the headline's, strings cut
    at 100 characters, floats at five significant digits, the three longest stages of kernels_ms."""
    if isinstance(v, dict):
        if key == "kernels_ms":
            v = dict(sorted(v.items(), key=lambda kv: -kv[1])[:3])
        if "scaling" in v and v["scaling"] != "weak":
            v = dict(v, strong_scaling=True)
        return {k: compact_entry(x, k, depth + 1, keep) for k, x in v.items() if x is not None and (k not in _COMPACT_DROP or k in keep)}
    if isinstance(v, float):
        return float("%.5g" % v)
    if isinstance(v, str):
        return v if len(v) <= 100 else v[:97] + "..."
    if isinstance(v, (list, tuple)):
        return [compact_entry(x, None, depth + 1, keep) for x in v[:8]]
    return v


S = None
_CTX = {}

if __name__ == "__main__":
    main()
