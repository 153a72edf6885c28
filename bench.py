#!/usr/bin/env python3
"""bench.py — headline benchmark: bsw 'large' GCUPS on N MI355X (BASELINE.json), one JSON line on rank 0.

    python bench.py [--gpus N --steps K --warmup W]            # bsw large (the headline metric)
    python bench.py --kernel chain|phmm|poa                    # the other three DP kernels, same JSON shape

A step = one pass of the kernel's hot path (every launch of the *_device entry point) over this
rank's shard of the synthetic dataset, inputs resident in HBM.  Work units are independent, so ranks
shard them with no data-path collective ("weak": per-GPU work fixed); value = units of ALL ranks per
second of the slowest rank.  Per-kernel durations come from HIP events recorded around every launch on
the launch stream inside the timed region (gbx_profile_begin/end).
"""
import argparse
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
# read-length range handled by each phmm kernel (csrc/phmm_kernels.hip: 31 row lanes x K rows per lane on the stream
# path, one pair per wavefront beyond 248 rows)
PHMM_CLASS = {"phmm_stream_rpl%d" % k: (31 * (k - 1) + 1, 31 * k) for k in range(1, 9)}
PHMM_CLASS.update({"phmm_f32_rpl4": (249, 256), "phmm_f32_rpl6": (257, 384), "phmm_f32_rpl8": (385, 1 << 30)})


# ------------------------------------------------------------------------------------------ workloads
class BswWork:
    metric, unit, dtype = "bsw_large_gcups", "GCUPS", "int32"

    def __init__(self, args, rank, dev):
        from genomicsbench_amd.bsw import DeviceBswBatch, make_params
        from genomicsbench_amd.datagen import gen_bsw
        self.n = args.size or 2_000_000
        self.params = make_params()
        self.batch = gen_bsw(self.n, 1002, first=rank * self.n)
        self.d = DeviceBswBatch(self.batch, dev)
        self.units = float(self.batch.nominal_cells)
        self.workload = ("bsw large: %d synthetic 151-bp seed-extension pairs per GPU (seed 1002), "
                         "nominal cells = sum len1*len2" % self.n)
        self.extra = {"pairs_per_gpu": self.n, "nominal_cells_per_gpu": self.batch.nominal_cells}

    def run(self, stream):
        self.d.run(self.params, stream)

    def roofline_bytes(self, kernel):
        b = self.batch
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
        for _ in range(3):
            t0 = time.perf_counter()
            extend_host(self.params, self.batch, out)
            ms.append((time.perf_counter() - t0) * 1e3)
        got = self.d.results()
        got = np.stack([got[f] for f in ("score", "tle", "gtle", "qle", "gscore", "max_off")], axis=1) if got.dtype.names else got
        return {"first_call_ms": ms[0], "best_ms": min(ms), "value": self.batch.nominal_cells / (min(ms) * 1e-3) / 1e9,
                "unit": "GCUPS", "what": "gbx_bsw_extend_host on the rank-0 shard: H2D + kernels + D2H from pageable memory",
                "same_as_device_entry": bool(np.array_equal(np.asarray(got), out))}

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
            secs, best = C.c_double(0.0), None
            for _ in range(3):      # best of 3; the driver's own timed region (objects are built outside it)
                ref.ref_bsw_getscores16_mt(*O._bsw_args(self.params, sample, out), C.c_int32(512), C.c_int32(cores),
                                           C.byref(secs))
                best = secs.value if best is None else min(best, secs.value)
            kind, dt, what = "reference", best, "reference AVX2 getScores16 -b 512, one object per thread"
            cols = [0, 1, 3, 5]     # score, tle, qle, max_off: the fields the AVX2 path defines like the scalar one
        else:
            t0 = time.perf_counter()
            out = O.bsw_oracle(self.params, sample, cores)
            kind, dt, what = "port", time.perf_counter() - t0, "oracle/bsw_oracle.c scalar restatement, OpenMP"
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

    def __init__(self, args, rank, dev):
        from genomicsbench_amd.chain import DeviceChainBatch
        from genomicsbench_amd.datagen import gen_chain
        self.n = args.size or 10_000
        self.case = gen_chain(self.n, 2001, first=rank * self.n)
        self.d = DeviceChainBatch(*self.case, dev)
        self.units = None                                   # evaluated predecessor pairs: read from the device counter
        self.workload = ("chain large: %d synthetic minimap2 chaining calls per GPU (seed 2001), "
                         "cell = evaluated predecessor pair" % self.n)
        self.extra = {"calls_per_gpu": self.n, "anchors_per_gpu": int(self.case[0][-1])}

    def run(self, stream):
        self.d.run(stream)

    def finish(self, stream):
        self.units = float(self.d.evaluated_pairs(stream))
        self.extra["evaluated_pairs_per_gpu"] = int(self.units)

    def roofline_bytes(self, kernel):
        return 40 * int(self.case[0][-1]), self.units        # 16 B anchor + 4x4 B outputs + 8 B re-read of score/parent

    def cpu_baseline(self, max_units):
        from oracle import oracle_py as O
        cores = os.cpu_count() or 1
        n = min(self.n, max_units or 2000)
        off, ax, ay, hdr = self.case
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

    def __init__(self, args, rank, dev):
        from genomicsbench_amd.datagen import gen_phmm
        from genomicsbench_amd.phmm import DevicePhmmBatchSet
        self.n = args.size or 20_000
        self.bs = gen_phmm(self.n, 3001, first=rank * self.n)
        self.d = DevicePhmmBatchSet(self.bs, dev)
        self.units = float(self.bs.cells)
        self.workload = ("phmm large: %d synthetic GATK batches per GPU (seed 3001), %d read x haplotype pairs, "
                         "cell = rslen*haplen; fp32 with fp64 redo below 1e-28" % (self.n, self.bs.n_pairs))
        self.extra = {"batches_per_gpu": self.n, "pairs_per_gpu": self.bs.n_pairs, "cells_per_gpu": self.bs.cells}

    def run(self, stream):
        self.d.run(stream)

    def roofline_bytes(self, kernel):
        bs = self.bs
        lo, hi = PHMM_CLASS.get(kernel, (1, 1 << 30))
        rl, hl = bs.read_len[bs.pair_read].astype(np.int64), bs.hap_len[bs.pair_hap].astype(np.int64)
        sel = (rl >= lo) & (rl <= hi)
        return int((5 * rl[sel] + hl[sel] + 8).sum()), float((rl[sel] * hl[sel]).sum())

    def cpu_baseline(self, max_units):
        from oracle import oracle_py as O
        cores = os.cpu_count() or 1
        sub = self.bs.take_batches(0, min(self.n, max_units or 400))
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

    def __init__(self, args, rank, dev):
        from genomicsbench_amd.datagen import gen_poa
        from genomicsbench_amd.poa import DevicePoaWindowSet, make_params
        self.n = args.size or 6_000
        self.params = make_params()
        self.ws = gen_poa(self.n, 4001, first=rank * self.n)
        self.d = DevicePoaWindowSet(self.ws, dev)
        self.units = None                                   # graph nodes x sequence length, device counter
        self.workload = ("poa large: %d synthetic 500-bp consensus windows per GPU (seed 4001), "
                         "cell = graph node x sequence position per alignment" % self.n)
        self.extra = {"windows_per_gpu": self.n, "sequences_per_gpu": self.ws.n_seqs,
                      "workspace_gb": round(self.d.work_bytes / 1e9, 2)}

    def run(self, stream):
        self.d.run(self.params, stream)

    def finish(self, stream):
        self.units = float(self.d.cells(stream))
        self.extra["dp_cells_per_gpu"] = int(self.units)

    def roofline_bytes(self, kernel):
        return int(self.units * 13), self.units             # 3 x 2 B written (H, F, O) + 3 x 2 B x ~1.2 predecessor rows read per cell

    def cpu_baseline(self, max_units):
        from oracle import oracle_py as O
        cores = os.cpu_count() or 1
        sub = self.ws.take(0, min(self.n, max_units or 256))
        t0 = time.perf_counter()
        want, cells = O.poa_oracle(self.params, sub, cores, return_cells=True)
        dt = time.perf_counter() - t0
        same = self.d.results()[:sub.n_windows] == want
        return {"value": cells / dt / 1e9, "unit": "GCUPS", "cores": cores, "kind": "port",
                "sample": "first %d windows, oracle/poa_oracle.c scalar spoa restatement, OpenMP, %.2f s "
                          "(spoa is an empty submodule: no reference build)" % (sub.n_windows, dt),
                "verified": "device consensus strings of these %d windows %s the CPU run's"
                            % (sub.n_windows, "identical to" if same else "DIFFER from")}


WORKLOADS = {"bsw": BswWork, "chain": ChainWork, "phmm": PhmmWork, "poa": PoaWork}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--kernel", choices=sorted(WORKLOADS), default="bsw")
    ap.add_argument("--size", type=int, default=0, help="units per GPU (pairs / calls / batches / windows); 0 = 'large'")
    ap.add_argument("--pairs", type=int, default=0, help="alias of --size for bsw")
    ap.add_argument("--cpu-units", type=int, default=0, help="units in the CPU-baseline sample (0 = kernel default)")
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()
    if args.pairs:
        args.size = args.pairs

    import torch
    import torch.distributed as dist
    from genomicsbench_amd import _native as N

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (libgbx has no CPU path)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    work = WORKLOADS[args.kernel](args, rank, dev)
    stream = torch.cuda.current_stream().cuda_stream

    for _ in range(args.warmup):
        work.run(stream)
    barrier()
    N.profile_begin()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        work.run(stream)
    barrier()
    dt = time.perf_counter() - t0
    stages = N.profile_end()
    if hasattr(work, "finish"):
        work.finish(stream)

    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    units = torch.tensor([work.units], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(units, op=dist.ReduceOp.SUM)
    dt_max, total_units = float(t.item()), float(units.item())

    if rank == 0:
        # roofline of the dominant kernel: algorithmic bytes of the units it processes / its mean duration
        name, (ms_sum, launches) = max(stages.items(), key=lambda kv: kv[1][0])
        k_ms = ms_sum / max(launches, 1)
        alg_bytes, k_units = work.roofline_bytes(name)
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9
        traffic = None                      # HBM bytes per launch from the committed rocprofv3 PMC passes (default sizes only)
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if not args.size and os.path.exists(tpath):
            traffic = json.load(open(tpath))["bytes_per_launch"].get(name)
        valu = None                         # VALU issue fraction of the dominant kernel, same kind of committed PMC pass
        vpath = os.path.join(ROOT, "profiles", "valu_busy.json")
        if not args.size and os.path.exists(vpath):
            valu = json.load(open(vpath))["valu_busy"].get(name)
        cfg = {"workload": work.workload,
               "parallelism": "units sharded over %d rank(s), no data-path collective" % world}
        cfg.update(work.extra)
        line = {
            "metric": work.metric, "value": total_units * args.steps / dt_max / 1e9, "unit": work.unit,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt_max / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": work.dtype, "data": "synthetic", "config": cfg,
            "roofline": {"bound": "hbm", "kernel": name, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "valu_busy": valu, "kernel_ms": k_ms,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "cells_per_s_dominant_kernel": (k_units or 0.0) / (k_ms * 1e-3)},
            "kernels_ms": {k: v[0] / max(v[1], 1) for k, v in sorted(stages.items())},
        }
        if not args.no_cpu:
            line["cpu_baseline"] = work.cpu_baseline(args.cpu_units)
            if hasattr(work, "host_entry"):
                line["host_entry"] = work.host_entry()
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
