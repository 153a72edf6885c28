"""The C++ drivers keep the reference CLIs and file formats: plumbing on CPU, end-to-end parity on the GPU."""
import json
import os
import subprocess

import numpy as np
import pytest

from conftest import has_gpu
from genomicsbench_amd import io as gio
from genomicsbench_amd.bsw import make_params
from genomicsbench_amd.datagen import gen_bsw, gen_chain, gen_phmm, gen_poa
from genomicsbench_amd.poa import make_params as poa_params
from oracle import oracle_py as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "genomicsbench_amd", "bin")


@pytest.fixture(scope="module")
def data(tmp_path_factory):
    d = tmp_path_factory.mktemp("inputs")
    b = gen_bsw(3000, 11)
    gio.write_bsw_pairs(str(d / "pairs.txt"), b)
    c = gen_chain(12, 12)
    gio.write_chain_calls(str(d / "chain.in"), *c)
    ph = gen_phmm(5, 13)
    gio.write_phmm_batches(str(d / "phmm.in"), ph)
    po = gen_poa(4, 14)
    gio.write_poa_windows(str(d / "poa.fasta"), po)
    return d, b, c, ph, po


def run(args):
    return subprocess.run(args, capture_output=True, text=True, timeout=600)


def test_drivers_exist_and_print_usage():
    for name in ("bsw", "chain", "phmm", "poa", "fmi"):
        assert os.path.exists(os.path.join(BIN, name)), "driver %s not built" % name
    assert "Need five arguments : ref_file query_set batch_size minSeedLen n_threads" in run([os.path.join(BIN, "fmi")]).stderr
    assert "usage: bsw -pairs" in run([os.path.join(BIN, "bsw")]).stderr
    assert run([os.path.join(BIN, "chain")]).returncode != 0


def test_gkl_dropin_library_exports_the_symbols_the_reference_driver_binds():
    """libgkl_pairhmm_c.so (csrc/shims/gkl_pairhmm_shim.cpp): the C++-mangled names PairHMMUnitTest.cpp:84-86 declares
    and the ConvertChar table of pairhmm_common.h:27, so that the unmodified driver links."""
    lib = os.path.join(ROOT, "genomicsbench_amd", "libgkl_pairhmm_c.so")
    assert os.path.exists(lib), "make -C genomicsbench_amd/csrc builds it next to libgbx.so"
    syms = run(["nm", "-D", "--defined-only", lib]).stdout
    for s in ("_Z11initPairHMMv", "_Z22computelikelihoodsbothP8testcasePdi", "_Z23computelikelihoodsfloatP8testcasePf",
              "_ZN11ConvertChar15conversionTableE"):
        assert s in syms, s


def test_bsw_class_shim_defines_every_public_entry_point_of_the_class():
    """csrc/shims/bsw_class_shim.cpp against the reference's bandedSWA.h:116-315: a harness that calls scalarBandedSWA,
    scalarBandedSWAWrapper, getScores8 / 16 and both batch wrappers links against the shim alone (oracle/build_ref.sh;
    only where /root/reference was present at build time), i.e. no member of the class is left undefined."""
    exe = os.path.join(ROOT, "oracle", "_ref", "bsw_members_gbx")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/bsw_members_gbx not built (no /root/reference at build time)")
    undefined = [l for l in run(["nm", "-u", "-C", exe]).stdout.splitlines() if "BandedPairWiseSW" in l]
    assert undefined == [], undefined
    defined = run(["nm", "-C", "--defined-only", exe]).stdout
    for m in ("scalarBandedSWA(", "scalarBandedSWAWrapper(", "getScores8(", "getScores16(", "smithWatermanBatchWrapper8(",
              "smithWatermanBatchWrapper16(", "getTicks("):
        assert "BandedPairWiseSW::" + m in defined, m


def test_bsw_binary_input_cache(tmp_path):
    """`bsw --cache FILE` writes the converted arrays after the first conversion and maps them on
    later runs of the same input - same arrays (checksum), whatever the thread count; a cache of an input that has changed
    since (size / modification time) is ignored and rewritten."""
    from genomicsbench_amd.datagen import gen_bsw, write_bsw_pairs_fast
    pairs, cache = str(tmp_path / "pairs.txt"), str(tmp_path / "pairs.gbxcache")
    write_bsw_pairs_fast(pairs, gen_bsw(3000, 5))
    exe = os.path.join(BIN, "bsw")
    first = run([exe, "-pairs", pairs, "-t", "3", "--parse-only", "1", "--cache", cache])
    assert first.returncode == 0 and os.path.exists(cache) and "mapped from the cache" not in first.stdout, first.stderr
    want = json.loads(first.stdout.strip().splitlines()[-1])
    for t in ("1", "5"):
        r = run([exe, "-pairs", pairs, "-t", t, "--parse-only", "1", "--cache", cache])
        assert r.returncode == 0 and "mapped from the cache" in r.stdout, r.stdout + r.stderr
        got = json.loads(r.stdout.strip().splitlines()[-1])
        assert got["pairs"] == 3000 and got["checksum"] == want["checksum"]
    write_bsw_pairs_fast(pairs, gen_bsw(2000, 6))                 # another input under the same name
    r = run([exe, "-pairs", pairs, "-t", "2", "--parse-only", "1", "--cache", cache])
    got = json.loads(r.stdout.strip().splitlines()[-1])
    assert "mapped from the cache" not in r.stdout and got["pairs"] == 2000 and got["checksum"] != want["checksum"]
    r = run([exe, "-pairs", pairs, "-t", "2", "--parse-only", "1", "--cache", cache])
    assert "mapped from the cache" in r.stdout and json.loads(r.stdout.strip().splitlines()[-1])["checksum"] == got["checksum"]


def _fnv1a(chunks):
    h = 1469598103934665603
    for c in chunks:
        for b in bytes(c):
            h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def test_bsw_parallel_ingest_matches_an_independent_reader(data):
    """the driver's multithreaded ingest (line split, digit conversion, packing) yields the same
    arrays whatever the thread count, and the same as the Python reader (checksum over h0, lengths and bases)."""
    d, b = data[0], data[1]
    outs = []
    for t in ("1", "3", "8"):
        r = run([os.path.join(BIN, "bsw"), "-pairs", str(d / "pairs.txt"), "-t", t, "--parse-only", "1"])
        assert r.returncode == 0, r.stderr
        outs.append(json.loads(r.stdout.strip().splitlines()[-1]))
    assert all(o["pairs"] == b.n for o in outs) and len({o["checksum"] for o in outs}) == 1
    chunks = [b.h0.astype(np.int32).tobytes(), b.len1.astype(np.int32).tobytes(), b.len2.astype(np.int32).tobytes()]
    for k in range(b.n):
        chunks.append(b.ref[b.idr[k]:b.idr[k] + b.len1[k]].tobytes())
        chunks.append(b.qer[b.idq[k]:b.idq[k] + b.len2[k]].tobytes())
    assert outs[0]["checksum"] == "%016x" % _fnv1a(chunks)


def test_chain_parallel_ingest_matches_an_independent_reader(data):
    d, c = data[0], data[2]
    off, ax, ay, hdr = c
    outs = []
    for t in ("1", "5"):
        r = run([os.path.join(BIN, "chain"), "-i", str(d / "chain.in"), "-o", str(d / "chain.unused"), "-t", t, "--parse-only"])
        assert r.returncode == 0, r.stderr
        outs.append(json.loads(r.stdout.strip().splitlines()[-1]))
    assert outs[0]["calls"] == len(off) - 1 and outs[0]["anchors"] == int(off[-1]) and outs[0]["checksum"] == outs[1]["checksum"]
    chunks = [np.asarray(off, dtype=np.int64).tobytes()]
    for k in range(len(off) - 1):
        chunks.append(np.float32(hdr[k]["avg_qspan"]).tobytes())
        chunks.append(np.array([hdr[k]["max_dist_x"], hdr[k]["max_dist_y"], hdr[k]["bw"], hdr[k]["n_segs"]], dtype=np.int32).tobytes())
    chunks += [np.asarray(ax[:off[-1]], dtype=np.uint64).tobytes(), np.asarray(ay[:off[-1]], dtype=np.uint64).tobytes()]
    assert outs[0]["checksum"] == "%016x" % _fnv1a(chunks)


def test_phmm_parallel_ingest_matches_an_independent_reader(data):
    d = data[0]
    bs = gio.read_phmm_batches(str(d / "phmm.in"))
    outs = []
    for t in ("1", "6"):
        r = run([os.path.join(BIN, "phmm"), "-f", str(d / "phmm.in"), "-t", t, "--parse-only"])
        assert r.returncode == 0, r.stderr
        outs.append(json.loads(r.stdout.strip().splitlines()[-1]))
    assert outs[0]["pairs"] == bs.n_pairs and outs[0]["checksum"] == outs[1]["checksum"]
    nb = int(bs.read_len.astype(np.int64).sum()), int(bs.hap_len.astype(np.int64).sum())
    chunks = [bs.read_len.astype(np.int32).tobytes(), bs.hap_len.astype(np.int32).tobytes(),
              bs.pair_read.astype(np.int32).tobytes(), bs.pair_hap.astype(np.int32).tobytes()]
    chunks += [a[:nb[0]].tobytes() for a in (bs.rs, bs.q, bs.qi, bs.qd, bs.qc)] + [bs.hap[:nb[1]].tobytes()]
    assert outs[0]["checksum"] == "%016x" % _fnv1a(chunks)


def test_poa_parallel_ingest_matches_an_independent_reader(data):
    d = data[0]
    ws = gio.read_poa_windows(str(d / "poa.fasta"))
    outs = []
    for t in ("1", "4"):
        r = run([os.path.join(BIN, "poa"), "-s", str(d / "poa.fasta"), "-t", t, "--parse-only"])
        assert r.returncode == 0, r.stderr
        outs.append(json.loads(r.stdout.strip().splitlines()[-1]))
    assert outs[0]["windows"] == ws.n_windows and outs[0]["sequences"] == ws.n_seqs and outs[0]["checksum"] == outs[1]["checksum"]
    seqs = b"".join(ws.arena[ws.seq_off[k]:ws.seq_off[k] + ws.seq_len[k]].tobytes() for k in range(ws.n_seqs))
    chunks = [np.asarray(ws.win_first_seq, dtype=np.int64).tobytes(), np.asarray(ws.seq_len, dtype=np.int32).tobytes(), seqs]
    assert outs[0]["checksum"] == "%016x" % _fnv1a(chunks)


def test_file_formats_roundtrip(data):
    d, b, c, ph, po = data
    assert gio.read_poa_windows(str(d / "poa.fasta")).window(1) == po.window(1)
    assert np.array_equal(gio.read_phmm_batches(str(d / "phmm.in")).read_len, ph.read_len)
    assert gio.read_bsw_pairs(str(d / "pairs.txt")).n == b.n


@pytest.mark.skipif(has_gpu(), reason="checks the no-GPU failure mode")
def test_drivers_fail_loudly_without_gpu(data):
    d = data[0]
    r = run([os.path.join(BIN, "bsw"), "-pairs", str(d / "pairs.txt"), "-t", "1", "-b", "512"])
    assert r.returncode != 0 and "no HIP device" in r.stderr


@pytest.mark.gpu
def test_bsw_driver_end_to_end(data):
    d, b = data[0], data[1]
    r = run([os.path.join(BIN, "bsw"), "-pairs", str(d / "pairs.txt"), "-t", "1", "-b", "512", "--repeat", "2",
             "--dump", str(d / "bsw.out")])
    assert r.returncode == 0, r.stderr
    assert "Number of input pairs: %d" % b.n in r.stdout
    got = np.loadtxt(str(d / "bsw.out"), dtype=np.int32)
    assert np.array_equal(got, O.bsw_oracle(make_params(), gio.read_bsw_pairs(str(d / "pairs.txt")), 4))
    assert json.loads(r.stdout.strip().split("\n")[-1])["pairs"] == b.n


@pytest.mark.gpu
def test_bsw_driver_overlapped_ingest(data):
    """--overlap S: the pairs converted in S slices, slice k through the host entry (re-based arena stretch, a caller thread of
    its own) while slice k+1 is converted: the same results as one call, and the e2e record."""
    d, b = data[0], data[1]
    want = O.bsw_oracle(make_params(), gio.read_bsw_pairs(str(d / "pairs.txt")), 4)
    for slices, threads in ((3, 1), (4, 4), (7, 2)):
        out = d / ("bsw_ov%d.out" % slices)
        r = run([os.path.join(BIN, "bsw"), "-pairs", str(d / "pairs.txt"), "-t", str(threads), "-b", "512", "--overlap", str(slices), "--dump", str(out)])
        assert r.returncode == 0, r.stderr
        assert np.array_equal(np.loadtxt(str(out), dtype=np.int32), want)
        rec = json.loads(r.stdout.strip().split("\n")[-1])
        assert rec["pairs"] == b.n and rec["overlap_slices"] == slices and rec["e2e_seconds"] >= rec["ingest_seconds"] > 0


@pytest.mark.gpu
def test_chain_driver_end_to_end(data):
    d, c = data[0], data[2]
    r = run([os.path.join(BIN, "chain"), "-i", str(d / "chain.in"), "-o", str(d / "chain.out"), "--print"])
    assert r.returncode == 0 and "Time in kernel" in r.stderr
    case = gio.read_chain_calls(str(d / "chain.in"))
    s, p, _, _ = O.chain_oracle(*case)
    import io
    buf = io.StringIO()
    gio.write_chain_returns(buf, case[0], s, p)
    assert open(str(d / "chain.out")).read() == buf.getvalue()


@pytest.mark.gpu
def test_drivers_with_binary_input_cache(data, tmp_path):
    """`--cache FILE` (optional): the first run converts the text and writes the arrays, the second maps them -
    both give the output of a run without a cache; bsw through --dump, chain through its output file."""
    d = data[0]
    ref_out, c1, c2 = str(tmp_path / "plain.out"), str(tmp_path / "first.out"), str(tmp_path / "second.out")
    cache = str(tmp_path / "chain.gbxcache")
    assert run([os.path.join(BIN, "chain"), "-i", str(d / "chain.in"), "-o", ref_out, "--print"]).returncode == 0
    r1 = run([os.path.join(BIN, "chain"), "-i", str(d / "chain.in"), "-o", c1, "--print", "-t", "3", "--cache", cache])
    r2 = run([os.path.join(BIN, "chain"), "-i", str(d / "chain.in"), "-o", c2, "--print", "--cache", cache])
    assert r1.returncode == 0 and r2.returncode == 0 and "mapped from the cache" not in r1.stderr and "mapped from the cache" in r2.stderr, r1.stderr + r2.stderr
    assert open(c1).read() == open(ref_out).read() == open(c2).read()
    bcache = str(tmp_path / "bsw.gbxcache")
    dumps = []
    for k in range(3):
        dump = str(tmp_path / ("bsw%d.txt" % k))
        args = [os.path.join(BIN, "bsw"), "-pairs", str(d / "pairs.txt"), "-t", "2", "-b", "512", "--dump", dump] + (["--cache", bcache] if k else [])
        r = run(args)
        assert r.returncode == 0 and ("mapped from the cache" in r.stdout) == (k == 2), r.stdout + r.stderr
        dumps.append(open(dump).read())
    assert dumps[0] == dumps[1] == dumps[2] and len(dumps[0]) > 100


@pytest.mark.gpu
def test_phmm_driver_end_to_end(data):
    d, ph = data[0], data[3]
    r = run([os.path.join(BIN, "phmm"), "-f", str(d / "phmm.in"), "-t", "1", "--print"])
    assert r.returncode == 0 and "Kernel runtime" in r.stdout
    vals = []
    for ln in r.stdout.split("\n"):
        try:
            vals.append(float(ln))
        except ValueError:
            pass
    want = O.phmm_oracle(gio.read_phmm_batches(str(d / "phmm.in")), 4)
    got = np.array(vals[:len(want)])
    assert len(vals) >= len(want) and np.all(np.abs(got - want) <= 1e-5 * np.maximum(1, np.abs(want)) + 5e-7)


@pytest.mark.gpu
def test_poa_driver_end_to_end(data):
    d, po = data[0], data[4]
    r = run([os.path.join(BIN, "poa"), "-s", str(d / "poa.fasta"), "-t", "1", "--print", "--warmup"])
    assert r.returncode == 0, r.stderr
    lines = [ln for ln in r.stdout.split("\n") if ln]
    assert lines[0::2] == [">Consensus_sequence"] * po.n_windows
    assert lines[1::2] == O.poa_oracle(poa_params(), po, 4)


def test_config0_reference_cpu_driver_plumbing(tmp_path):
    """BASELINE config 0: the reference's own bsw driver (unmodified main_banded.cpp + bandedSWA.cpp, built by
    oracle/build_ref.sh) with -t 1 -b 512 as R/scripts/run-cpu.sh:61 runs it, on a generated input file in the
    reference's 3-lines-per-pair format.  CPU only.  scripts/run-cpu-bsw-small.sh runs the 100 000-pair set."""
    import subprocess
    from genomicsbench_amd import io as gio
    from genomicsbench_amd.datagen import gen_bsw
    exe = os.path.join(ROOT, "oracle", "_ref", "bsw_refdriver_cpu")
    if not os.path.exists(exe):
        pytest.skip("reference CPU driver not built (no /root/reference at build time)")
    pairs = str(tmp_path / "pairs.txt")
    gio.write_bsw_pairs(pairs, gen_bsw(2000, 1001))
    r = subprocess.run([exe, "-pairs", pairs, "-t", "1", "-b", "512"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 1                                  # by design, main_banded.cpp:352
    assert "Number of input pairs: 2000" in r.stdout and "Total Pairs processed: 2000" in r.stdout
    assert "Overall SW cycles" in r.stdout


# ---- fmi: the reference CLI  fmi <ref_file> <query_set> <batch_size> <minSeedLen> <n_threads>  (fmi.cpp:54-58)
@pytest.fixture(scope="module")
def fmi_data(tmp_path_factory):
    from genomicsbench_amd import fmi as FM
    from genomicsbench_amd.datagen import gen_fmi_genome, gen_fmi_reads
    d = tmp_path_factory.mktemp("fmi")
    g = gen_fmi_genome(60000, 6001)
    idx = FM.build_index(g)
    FM.save_index(idx, str(d / "genome.gbxfmi"))
    rs = gen_fmi_reads(g, 500, 6002)
    ragged = FM.FmiReadSet(rs.enc, rs.read_off, np.maximum(1, rs.read_len - (np.arange(500) % 7).astype(np.int32) * 9))
    FM.write_reads(str(d / "reads.fastq"), ragged, fastq=True)
    FM.write_reads(str(d / "reads.fasta"), ragged, fastq=False, wrap=60)
    return d, idx, ragged


def test_fmi_index_file_roundtrip_and_parallel_ingest(fmi_data):
    """save_index / load_index keep every table; FASTQ and wrapped FASTA give the same encoded reads for 1 and 4 ingest
    threads, equal to the bytes the Python side holds (reads padded to the longest with 4s, fmi.cpp:97-127)."""
    from genomicsbench_amd import fmi as FM
    d, idx, rs = fmi_data
    back = FM.load_index(str(d / "genome.gbxfmi"))
    assert back.ref_seq_len == idx.ref_seq_len and back.count == idx.count and back.sentinel_index == idx.sentinel_index
    assert np.array_equal(back.cp_occ.view(np.uint8), idx.cp_occ.view(np.uint8))
    L = rs.max_len
    enc = np.full((rs.n_reads, L), 4, dtype=np.uint8)
    for r in range(rs.n_reads):
        a = int(rs.read_off[r])
        enc[r, :rs.read_len[r]] = rs.enc[a:a + rs.read_len[r]]
    want = "%016x" % _fnv1a([rs.read_len.astype("<i4").tobytes(), enc.tobytes()])
    for name in ("reads.fastq", "reads.fasta"):
        for t in ("1", "4"):
            r = run([os.path.join(BIN, "fmi"), str(d / "genome.gbxfmi"), str(d / name), "512", "19", t, "--parse-only"])
            assert r.returncode == 0, r.stderr
            got = json.loads(r.stdout.strip().splitlines()[-1])
            assert got["reads"] == rs.n_reads and got["max_readlength"] == L and got["fnv1a"] == want, (name, t)


def test_fmi_driver_derives_the_sentinel_of_a_trailerless_index(fmi_data, tmp_path):
    """A .bwt.2bit.64 with one suffix-array sample per row and no trailing sentinel_index (what a build without SA compression
    that derives the sentinel on load would write): the driver finds the row whose sample is 0."""
    from genomicsbench_amd import fmi as FM
    d, idx, _ = fmi_data
    n = idx.ref_seq_len
    sa = np.arange(1, n + 1, dtype=np.int64)                        # (only which sample is 0 matters to the reader)
    sa[idx.sentinel_index] = 0
    path = FM.save_bwa_mem2_index(idx, str(tmp_path / "plain.fa"), sa=sa, sa_compx=0)
    with open(path, "r+b") as f:
        f.truncate(os.path.getsize(path) - 8)
    r = run([os.path.join(BIN, "fmi"), "--index-info", path])
    assert r.returncode == 0, r.stderr
    got = json.loads(r.stdout)
    own = json.loads(run([os.path.join(BIN, "fmi"), "--index-info", str(d / "genome.gbxfmi")]).stdout)
    assert got == own and "derived" in r.stderr


def test_fmi_driver_reads_a_bwa_mem2_index_by_prefix(fmi_data, tmp_path):
    """The reference opens the index by prefix (fmi.cpp:79-80: FMI_search(argv[1]), load_index() reads
    <prefix>.bwt.2bit.64).  save_bwa_mem2_index writes that layout as published (both sizes of the suffix-array sample
    section), the Python reader and the driver's reader give back the tables save_index / load_index hold."""
    from genomicsbench_amd import fmi as FM
    d, idx, _ = fmi_data
    own = json.loads(run([os.path.join(BIN, "fmi"), "--index-info", str(d / "genome.gbxfmi")]).stdout)
    for compx in (3, 0):
        prefix = str(tmp_path / ("genome_%d.fa" % compx))
        path = FM.save_bwa_mem2_index(idx, prefix, sa_compx=compx)
        n = idx.ref_seq_len
        assert os.path.getsize(path) == 48 + ((n >> 6) + 1) * 64 + 5 * (((n >> 3) + 1) if compx else n) + 8
        back = FM.load_bwa_mem2_index(prefix)
        assert back.ref_seq_len == idx.ref_seq_len and list(back.count) == list(idx.count) and back.sentinel_index == idx.sentinel_index
        assert np.array_equal(back.cp_occ.view(np.uint8), idx.cp_occ.view(np.uint8))
        for arg in (prefix, path):                                   # by prefix, and the file itself
            r = run([os.path.join(BIN, "fmi"), "--index-info", arg])
            assert r.returncode == 0, r.stderr
            assert json.loads(r.stdout) == own
    bad = str(tmp_path / "bad.bwt.2bit.64")
    open(bad, "wb").write(open(path, "rb").read()[:-3])
    assert run([os.path.join(BIN, "fmi"), "--index-info", bad]).returncode != 0
    with pytest.raises(ValueError):
        FM.load_bwa_mem2_index(bad)


@pytest.mark.gpu
def test_fmi_driver_on_a_bwa_mem2_index(fmi_data, tmp_path):
    from genomicsbench_amd import fmi as FM
    d, idx, rs = fmi_data
    prefix = str(tmp_path / "genome.fa")
    FM.save_bwa_mem2_index(idx, prefix)
    r = run([os.path.join(BIN, "fmi"), prefix, str(d / "reads.fastq"), "512", "19", "2", "--print"])
    assert r.returncode == 0, r.stderr
    lines = r.stdout.splitlines()
    want, _ = O.fmi_oracle(idx, rs, FM.default_params(19))
    k = next(i for i, ln in enumerate(lines) if ln.startswith("totalSmems"))
    assert lines[k] == "totalSmems = %d" % len(want) and lines[k + 1:] == FM.smems_text(want)


@pytest.mark.gpu
def test_fmi_driver_prints_the_oracles_smems(fmi_data):
    from genomicsbench_amd import fmi as FM
    d, idx, rs = fmi_data
    r = run([os.path.join(BIN, "fmi"), str(d / "genome.gbxfmi"), str(d / "reads.fastq"), "512", "19", "2", "--print"])
    assert r.returncode == 0, r.stderr
    lines = r.stdout.splitlines()
    want, _ = O.fmi_oracle(idx, rs, FM.default_params(19))
    k = next(i for i, ln in enumerate(lines) if ln.startswith("totalSmems"))
    assert lines[k] == "totalSmems = %d" % len(want)
    assert lines[k + 1:] == FM.smems_text(want)


# ---- --gpus N: the drivers hand the device count to the host entries (gbx_host_set_devices); on a one-GPU box the logical
# devices are mapped onto GPU 0 (GBX_DEVICE_MAP) and small jobs are cut all the same (GBX_SHARD_MIN_UNITS)
def test_gpus_flag_is_taken_out_of_the_reference_options(data):
    """--gpus sits anywhere on the command line without disturbing the reference-style option loops (CPU: --parse-only)."""
    d, b = data[0], data[1]
    for args in (["--gpus", "4", "-pairs", str(d / "pairs.txt"), "-t", "2", "--parse-only", "1"],
                 ["-pairs", str(d / "pairs.txt"), "--gpus", "2", "-t", "2", "--parse-only", "1"],
                 ["-pairs", str(d / "pairs.txt"), "-t", "2", "--parse-only", "1", "--gpus", "8"]):
        r = run([os.path.join(BIN, "bsw")] + args)
        assert r.returncode == 0, r.stderr
        assert json.loads(r.stdout.strip().splitlines()[-1])["pairs"] == b.n
    assert run([os.path.join(BIN, "bsw"), "-pairs", str(d / "pairs.txt"), "--gpus", "0"]).returncode != 0
    r = run([os.path.join(BIN, "chain"), "--gpus", "3", "-i", str(d / "chain.in"), "-o", str(d / "unused"), "-t", "2", "--parse-only"])
    assert r.returncode == 0 and json.loads(r.stdout.strip().splitlines()[-1])["calls"] == len(data[2][0]) - 1


@pytest.mark.gpu
def test_drivers_with_gpus_flag_end_to_end(data, fmi_data):
    d, b, c, ph, po = data
    env = dict(os.environ, GBX_DEVICE_MAP="0,0,0", GBX_SHARD_MIN_UNITS="1")

    def run_env(args):
        return subprocess.run(args, capture_output=True, text=True, timeout=600, env=env)

    r = run_env([os.path.join(BIN, "bsw"), "-pairs", str(d / "pairs.txt"), "-t", "2", "--gpus", "3", "--dump", str(d / "bsw3.out")])
    assert r.returncode == 0 and " x 3" in r.stderr, r.stderr
    assert np.array_equal(np.loadtxt(str(d / "bsw3.out"), dtype=np.int32), O.bsw_oracle(make_params(), b, 4))
    r = run_env([os.path.join(BIN, "chain"), "-i", str(d / "chain.in"), "-o", str(d / "chain2.out"), "--print", "--gpus", "2"])
    assert r.returncode == 0, r.stderr
    case = gio.read_chain_calls(str(d / "chain.in"))
    s, p, _, _ = O.chain_oracle(*case)
    import io
    buf = io.StringIO()
    gio.write_chain_returns(buf, case[0], s, p)
    assert open(str(d / "chain2.out")).read() == buf.getvalue()
    r = run_env([os.path.join(BIN, "poa"), "-s", str(d / "poa.fasta"), "-t", "1", "--print", "--gpus", "2"])
    assert r.returncode == 0, r.stderr
    lines = [ln for ln in r.stdout.split("\n") if ln]
    assert lines[1::2] == O.poa_oracle(poa_params(), po, 4)
    r = run_env([os.path.join(BIN, "phmm"), "-f", str(d / "phmm.in"), "-t", "1", "--print", "--gpus", "2"])
    assert r.returncode == 0, r.stderr
    vals = []
    for ln in r.stdout.split("\n"):
        try:
            vals.append(float(ln))
        except ValueError:
            pass
    want = O.phmm_oracle(gio.read_phmm_batches(str(d / "phmm.in")), 4)
    assert len(vals) >= len(want) and np.all(np.abs(np.array(vals[:len(want)]) - want) <= 1e-5 * np.maximum(1, np.abs(want)) + 5e-7)
    from genomicsbench_amd import fmi as FM
    fd, idx, rs = fmi_data
    r = run_env([os.path.join(BIN, "fmi"), str(fd / "genome.gbxfmi"), str(fd / "reads.fastq"), "512", "19", "2", "--print", "--gpus", "3"])
    assert r.returncode == 0, r.stderr
    lines = r.stdout.splitlines()
    wantf, _ = O.fmi_oracle(idx, rs, FM.default_params(19))
    k = next(i for i, ln in enumerate(lines) if ln.startswith("totalSmems"))
    assert lines[k] == "totalSmems = %d" % len(wantf) and lines[k + 1:] == FM.smems_text(wantf)
