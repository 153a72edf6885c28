#!/usr/bin/env python3
"""Rehearsal of `bench.py --gpus N` on a box with fewer GPUs than ranks (the builder's one-GPU box): the N ranks share
the GPU(s) round-robin and the scatter / gather travel through host memory (GBX_BENCH_COMM=gloo, a test aid - the judged
path is RCCL).  What it proves: the default six-kernel protocol fits the GPU's memory with N ranks' buffers alive at
once, fits the driver's time limit, and the front of every shard verifies against the oracle.  Writes rank 0's JSON
line and a summary (wall time, peak VRAM in use, peak host memory of all ranks) under gpurun_out/.

    python3 scripts/rehearse_nrank.py <tag> <N> [bench.py arguments]

This process never touches the GPU: bench.py is a child (which starts the ranks as its own children)."""
import json
import os
import subprocess
import sys
import threading
import time

import psutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def vram_used():
    try:
        out = subprocess.run(["rocm-smi", "--showmeminfo", "vram", "--csv"], capture_output=True, text=True, timeout=20).stdout
        rows = [ln.split(",") for ln in out.strip().splitlines()[1:] if ln.strip()]
        return sum(int(r[2]) for r in rows if len(r) >= 3 and r[2].strip().isdigit())
    except Exception:
        return None


def main():
    tag, n = sys.argv[1], int(sys.argv[2])
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    base = os.path.join(out, "%s_%drank" % (tag, n))
    env = dict(os.environ, GBX_BENCH_COMM="gloo")
    # (no caching allocator in the ranks: eight of them on one device otherwise strand ~6 GB each in reserved-but-unused
    # segments, which is the difference between fitting 288 GB and not: 8 x (12.5 GB of fmi buffers + 18.4 GB of workspace))
    env.setdefault("PYTORCH_NO_CUDA_MEMORY_CACHING", "1")
    env.setdefault("PYTORCH_NO_HIP_MEMORY_CACHING", "1")
    peak = {"vram": 0, "rss": 0, "vram_idle": vram_used()}
    t0 = time.time()
    with open(base + "_bench.json", "w") as fo, open(base + "_stderr.log", "w") as fe:
        child = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n)] + sys.argv[3:], env=env, stdout=fo, stderr=fe, cwd=ROOT)
        stop = threading.Event()

        def poll():
            while not stop.is_set():
                v = vram_used()
                if v:
                    peak["vram"] = max(peak["vram"], v)
                try:
                    procs = [psutil.Process(child.pid)] + psutil.Process(child.pid).children(recursive=True)
                    peak["rss"] = max(peak["rss"], sum(p.memory_info().rss for p in procs))
                except psutil.Error:
                    pass
                stop.wait(2.0)

        th = threading.Thread(target=poll, daemon=True)
        th.start()
        try:
            rc = child.wait(timeout=float(os.environ.get("GBX_REHEARSE_TIMEOUT", 1750)))
        except subprocess.TimeoutExpired:
            for p in psutil.Process(child.pid).children(recursive=True):
                p.terminate()
            child.terminate()
            rc = -9
        stop.set()
        th.join()
    line = None
    for ln in open(base + "_bench.json"):
        if ln.startswith("{"):
            line = json.loads(ln)
    summ = {"ranks": n, "rc": rc, "wall_s": round(time.time() - t0, 1), "driver_limit_s": 1800,
            "vram_peak_bytes_in_use": peak["vram"], "vram_in_use_before_bytes": peak["vram_idle"], "host_rss_peak_bytes_all_ranks": peak["rss"],
            "argv": sys.argv[3:]}
    if line:
        ks = dict(line.get("kernels", {}), bsw=line)
        summ["kernels"] = {k: {"value": v["value"], "unit": v["unit"], "ms_per_step": v["ms_per_step"], "gather_verified": v.get("gather_verified"),
                               "shard_units": v.get("shard_units"), "dataset_gen_s": v.get("dataset_gen_s"), "scatter_ms": v.get("scatter_ms"),
                               "gather_ms": v.get("gather_ms")} for k, v in ks.items() if v}
        c4 = (line.get("kernels", {}).get("poa") or {}).get("config4_strong")
        if c4:
            summ["poa_config4_strong"] = {"value": c4["value"], "ms_per_step": c4["ms_per_step"], "gather_verified": c4.get("gather_verified"),
                                          "shard_units": c4.get("shard_units")}
    json.dump(summ, open(base + "_summary.json", "w"), indent=1)
    print(json.dumps(summ))
    print(open(base + "_stderr.log").read()[-800:], file=sys.stderr)
    sys.exit(0 if rc == 0 and line else 1)


if __name__ == "__main__":
    main()
