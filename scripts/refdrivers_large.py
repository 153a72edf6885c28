#!/usr/bin/env python3
"""The reference's UNMODIFIED drivers (oracle/_ref/*_refdriver_gbx: main_banded.cpp, PairHMMUnitTest.cpp, msa_spoa_omp.cpp
on our shims) on the 'large' inputs, called the way the reference's scripts call them - one small call per OpenMP thread
(run-cpu.sh:61 `-t <n> -b 512`) - with and without the host entries' call combiner (GBX_COMBINE=0), beside one call for
the whole job.  Prints each driver's own timed region.  Needs a GPU.

usage: refdrivers_large.py [bsw] [phmm] [poa] [--pairs N] [--batches N] [--windows N] [--out FILE]"""
import os
import re
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from genomicsbench_amd import io as gio  # noqa: E402
from genomicsbench_amd.datagen import gen_bsw, gen_phmm, gen_poa, write_bsw_pairs_fast  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref")


def arg(name, default):
    return int(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else default


def run(cmd, env=None, timeout=1500):
    e = dict(os.environ)
    e.update(env or {})
    t0 = time.time()
    with tempfile.TemporaryFile("w+") as so:            # (the phmm driver prints a line per pair: only its tail is read)
        r = subprocess.run(cmd, stdout=so, stderr=subprocess.PIPE, text=True, timeout=timeout, env=e)
        wall = time.time() - t0
        so.seek(0, os.SEEK_END)
        so.seek(max(0, so.tell() - 4000))
        r.stdout = so.read()
    return r, wall


def main():
    kernels = [k for k in ("bsw", "phmm", "poa") if k in sys.argv] or ["bsw", "phmm", "poa"]
    out = open(sys.argv[sys.argv.index("--out") + 1], "w") if "--out" in sys.argv else None

    def say(s):
        print(s, flush=True)
        if out:
            out.write(s + "\n")
            out.flush()

    tmp = tempfile.mkdtemp(prefix="gbx_refdrv_")
    quota = "?"
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        quota = "%.1f" % (int(q) / int(p)) if q != "max" else "none"
    except OSError:
        pass
    say("== unmodified reference drivers on the shims; host threads %d, cpu quota (cores) %s" % (os.cpu_count(), quota))
    if "bsw" in kernels:
        n = arg("--pairs", 2_000_000)
        path = os.path.join(tmp, "pairs.txt")
        write_bsw_pairs_fast(path, gen_bsw(n, 1002))
        exe = os.path.join(REF, "bsw_refdriver_gbx")
        for t, b in ((1, n), (64, 512), (16, 512), (8, 512), (1, 512), (64, 4096)):
            for comb in ("1", "0"):
                if comb == "0" and (b == n or (t, b) in ((8, 512), (64, 4096))):
                    continue
                r, wall = run([exe, "-pairs", path, "-t", str(t), "-b", str(b)], {"GBX_COMBINE": comb})
                m = re.search(r"Overall SW cycles = \d+, ([0-9.]+) s", r.stdout)
                say("bsw  %8d pairs  -t %-3d -b %-8d combine=%s   SW region %s s   (process %.1f s)" % (n, t, b, comb, m.group(1) if m else "?" + r.stderr[-200:], wall))
    if "phmm" in kernels:
        nb = arg("--batches", 20_000)
        path = os.path.join(tmp, "phmm.in")
        gio.write_phmm_batches(path, gen_phmm(nb, 3001))
        exe = os.path.join(REF, "phmm_refdriver_gbx")
        for t in (64, 16, 1):
            for comb in ("1", "0"):
                if t == 1 and comb == "0":
                    continue
                r, wall = run([exe, "-f", path, "-t", str(t)], {"GBX_COMBINE": comb})
                m = re.search(r"Kernel runtime: ([0-9.]+) sec", r.stdout[-400:])
                say("phmm %8d batches -t %-3d combine=%s   kernel region %s s   (process %.1f s)" % (nb, t, comb, m.group(1) if m else "?" + r.stderr[-200:], wall))
    if "poa" in kernels:
        nw = arg("--windows", 6_000)
        path = os.path.join(tmp, "poa.fasta")
        gio.write_poa_windows(path, gen_poa(nw, 4001))
        exe = os.path.join(REF, "poa_refdriver_gbx")
        for t in (64, 256, 16):
            for comb in ("1", "0"):
                if t == 256 and comb == "0":
                    continue
                r, wall = run([exe, "-s", path, "-t", str(t)], {"GBX_COMBINE": comb})
                m = re.search(r"Runtime: ([0-9.]+)", r.stderr)
                say("poa  %8d windows -t %-3d combine=%s   %s   (process %.1f s)" % (nw, t, comb, ("runtime " + m.group(1) + " s") if m else r.stderr[-200:].strip(), wall))
    for f in os.listdir(tmp):
        os.remove(os.path.join(tmp, f))
    os.rmdir(tmp)


if __name__ == "__main__":
    main()
