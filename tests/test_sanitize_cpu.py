"""Host-code hygiene on the CPU box: an ASan + UBSan build (tests/sanitize/Makefile, `make SAN=1` for the drivers)
of everything on the host side that runs without a GPU — the oracle's C restatements, the product's serial poa graph
code (host build of csrc/poa_graph.h), the generators, and the four drivers' threaded ingest (--parse-only) — must
run clean and produce the same checksums as the normal build.  GPU code is never sanitized (no GPU ASAN / XNACK on
this pool).  Analogue of the reference's VTune/ITT hooks in spirit only: SURVEY §5."""
import json
import os
import subprocess

import numpy as np
import pytest

from genomicsbench_amd import io as gio
from genomicsbench_amd.datagen import gen_bsw, gen_chain, gen_phmm, gen_poa

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = os.path.join(ROOT, "tests", "sanitize")
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1",
           OMP_NUM_THREADS="4")


@pytest.fixture(scope="module")
def san_build():
    r = subprocess.run(["make", "-C", SAN], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return os.path.join(SAN, "_build", "san_main")


def _clean(r):
    bad = [w for w in ("AddressSanitizer", "runtime error:", "LeakSanitizer") if w in r.stderr or w in r.stdout]
    assert r.returncode == 0 and not bad, (r.stdout[-1500:] + r.stderr[-3000:])


def test_oracle_and_product_graph_code_under_asan_ubsan(san_build):
    r = subprocess.run([san_build], capture_output=True, text=True, timeout=600, env=ENV)
    _clean(r)
    assert "san_main: ok" in r.stdout


def test_driver_ingest_under_asan_ubsan(san_build, tmp_path):
    """--parse-only of the sanitized drivers == the normal drivers' checksums, for 1 and 3 ingest threads."""
    gio.write_bsw_pairs(str(tmp_path / "pairs.txt"), gen_bsw(1500, 5))
    gio.write_chain_calls(str(tmp_path / "chain.in"), *gen_chain(10, 6))
    gio.write_phmm_batches(str(tmp_path / "phmm.in"), gen_phmm(8, 7))
    gio.write_poa_windows(str(tmp_path / "poa.fa"), gen_poa(6, 8))
    cmds = {"bsw": ["-pairs", str(tmp_path / "pairs.txt"), "--parse-only", "1"],
            "chain": ["-i", str(tmp_path / "chain.in"), "-o", str(tmp_path / "unused"), "--parse-only"],
            "phmm": ["-f", str(tmp_path / "phmm.in"), "--parse-only"],
            "poa": ["-s", str(tmp_path / "poa.fa"), "--parse-only"]}
    # the drivers link libgbx.so (and through it the HIP runtime, which keeps process-lifetime allocations): leak
    # checking is for san_main above; here ASan/UBSan watch the ingest code itself
    env = dict(ENV, ASAN_OPTIONS="detect_leaks=0")
    for k, args in cmds.items():
        sums = set()
        for exe_dir, t in (("bin-san", "1"), ("bin-san", "3"), ("bin", "3")):
            exe = os.path.join(ROOT, "genomicsbench_amd", exe_dir, k)
            a = list(args)
            a[2:2] = ["-t", t]
            r = subprocess.run([exe] + a, capture_output=True, text=True, timeout=300, env=env)
            _clean(r)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
            sums.add(json.loads(line)["checksum"])
        assert len(sums) == 1, (k, sums)
    # fmi: positional CLI (ref_file query_set batch_size minSeedLen n_threads), FASTQ and wrapped FASTA
    from genomicsbench_amd import fmi as FM
    from genomicsbench_amd.datagen import gen_fmi_genome, gen_fmi_reads
    g = gen_fmi_genome(20000, 9)
    rs = gen_fmi_reads(g, 300, 10)
    rs = FM.FmiReadSet(rs.enc, rs.read_off, np.maximum(1, rs.read_len - (np.arange(300) % 5).astype(np.int32) * 11))
    FM.write_reads(str(tmp_path / "r.fastq"), rs, fastq=True)
    FM.write_reads(str(tmp_path / "r.fasta"), rs, fastq=False, wrap=70)
    sums = set()
    for exe_dir, t, name in (("bin-san", "1", "r.fastq"), ("bin-san", "3", "r.fasta"), ("bin", "3", "r.fastq")):
        exe = os.path.join(ROOT, "genomicsbench_amd", exe_dir, "fmi")
        r = subprocess.run([exe, "unused", str(tmp_path / name), "512", "19", t, "--parse-only"], capture_output=True, text=True,
                           timeout=300, env=env)
        _clean(r)
        sums.add(json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])["fnv1a"])
    assert len(sums) == 1, ("fmi", sums)


def test_host_pipeline_under_thread_sanitizer(san_build):
    """csrc/host_pipeline.h (lanes, upload workers, downloader, pinned slabs, device-block cache, events) compiled
    unmodified against a mock HIP runtime (tests/sanitize/mock_hip: streams are in-order worker threads) and driven by
    four caller threads at once through every call shape the host entries use - staged, unstaged / packed, three
    overlapped chunks with packed bases, strided-field upload + scattered download, a call abandoned after start(), a
    transfer failing mid-call: no data race, no hang, right results; and the call combiner (csrc/host_combine.h, unmodified):
    twelve callers submitting small requests of two classes, two leaders in flight, failing requests redone one by one, every
    caller its own results and status.  The harness is checked to see a race when there is one."""
    exe = os.path.join(SAN, "_build", "pipe_tsan")
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=0:second_deadlock_stack=1")
    r = subprocess.run([exe, "--selftest-race"], capture_output=True, text=True, timeout=120, env=env)
    assert "ThreadSanitizer: data race" in r.stderr, "the TSan build does not report a deliberate race"
    r = subprocess.run([exe, "4", "1"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0 and "ThreadSanitizer" not in r.stderr and "pipe_tsan: ok" in r.stdout and "pipe_tsan: combiner" in r.stdout, \
        r.stdout[-1500:] + r.stderr[-4000:]
