#!/usr/bin/env python3
"""Full-size parity THROUGH the call combiner: the reference's unmodified drivers on the shims at -t 64 (one small call per OpenMP
thread, combined by the host entries) against one call of the same host entry for the whole 'large' job.
  bsw   : every getScores16 call's results (GBX_SHIM_DUMP: batch-local id + six fields per pair) as a multiset against the
          one-call results with the same batch-local ids (the dump's order depends on the threads)
  phmm  : the driver's printed log10 likelihoods (PRINT_OUTPUT, six decimals) against gbx_phmm_forward_host in one call
  poa   : the driver's printed consensus sequences against gbx_poa_consensus_host in one call
usage: check_refdrivers_large.py [bsw] [phmm] [poa] [--threads N]"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from genomicsbench_amd import io as gio  # noqa: E402
from genomicsbench_amd.datagen import gen_bsw, gen_phmm, gen_poa, write_bsw_pairs_fast  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref")
T = sys.argv[sys.argv.index("--threads") + 1] if "--threads" in sys.argv else "64"
kinds = [k for k in ("bsw", "phmm", "poa") if k in sys.argv] or ["bsw", "phmm", "poa"]
tmp = tempfile.mkdtemp(prefix="gbx_chk_")
ok_all = True
if "bsw" in kinds:
    from genomicsbench_amd.bsw import extend_host, make_params
    b = gen_bsw(2_000_000, 1002)
    path, dump = os.path.join(tmp, "pairs.txt"), os.path.join(tmp, "dump.txt")
    write_bsw_pairs_fast(path, b)
    subprocess.run([os.path.join(REF, "bsw_refdriver_gbx"), "-pairs", path, "-t", T, "-b", "512"], env=dict(os.environ, GBX_SHIM_DUMP=dump),
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    got = np.loadtxt(dump, dtype=np.int64)
    want = np.concatenate([(np.arange(b.n, dtype=np.int64) % 512)[:, None], extend_host(make_params(), b).astype(np.int64)], axis=1)
    srt = lambda a: a[np.lexsort(a.T[::-1])]
    ok = got.shape == want.shape and np.array_equal(srt(got), srt(want))
    print("bsw  -t %s -b 512: %d result rows from %d combined getScores16 calls %s the one-call results (as multisets of (batch-local id, six fields))"
          % (T, got.shape[0], (b.n + 511) // 512, "==" if ok else "DIFFER FROM"), flush=True)
    ok_all &= ok
    os.remove(path); os.remove(dump)
if "phmm" in kinds:
    from genomicsbench_amd.phmm import forward_host
    bs = gen_phmm(20_000, 3001)
    path, out = os.path.join(tmp, "phmm.in"), os.path.join(tmp, "phmm.out")
    gio.write_phmm_batches(path, bs)
    with open(out, "w") as fh:
        subprocess.run([os.path.join(REF, "phmm_refdriver_gbx"), "-f", path, "-t", T], stdout=fh, stderr=subprocess.DEVNULL)
    vals = []
    with open(out) as fh:
        for ln in fh:
            try:
                vals.append(float(ln))
            except ValueError:
                pass
    got = np.array(vals[-bs.n_pairs:])
    want = forward_host(bs)
    ok = len(got) == bs.n_pairs and bool(np.all(np.abs(got - want) <= 1e-6 + 1e-9 * np.abs(want)))      # printed with six decimals
    print("phmm -t %s: %d printed likelihoods %s one call's (to the six printed decimals; max |diff| %.2e)"
          % (T, len(got), "==" if ok else "DIFFER FROM", float(np.max(np.abs(got - want))) if len(got) == bs.n_pairs else -1), flush=True)
    ok_all &= ok
    os.remove(path); os.remove(out)
if "poa" in kinds:
    from genomicsbench_amd.poa import consensus_host, make_params as poa_params
    ws = gen_poa(6_000, 4001)
    path = os.path.join(tmp, "poa.fasta")
    gio.write_poa_windows(path, ws)
    r = subprocess.run([os.path.join(REF, "poa_refdriver_gbx"), "-s", path, "-t", T], capture_output=True, text=True)
    lines = r.stdout.splitlines()
    got = [lines[k + 1] for k in range(len(lines) - 1) if lines[k] == ">Consensus_sequence"]
    want = consensus_host(poa_params(), ws)
    ok = got == want
    print("poa  -t %s: %d printed consensus sequences %s one call's" % (T, len(got), "==" if ok else "DIFFER FROM"), flush=True)
    ok_all &= ok
    os.remove(path)
os.rmdir(tmp)
sys.exit(0 if ok_all else 1)
