"""GPU parity tests for poa: HIP kernel (through the C-ABI) vs the oracle, exact consensus strings."""
import numpy as np
import pytest

from genomicsbench_amd import _native as N
from genomicsbench_amd.datagen import gen_poa
from genomicsbench_amd.poa import DevicePoaWindowSet, PoaWindowSet, consensus_host, make_params
from oracle import oracle_py as O

pytestmark = pytest.mark.gpu


def diff(got, want):
    bad = [w for w in range(len(want)) if got[w] != want[w]]
    assert not bad, "%d/%d windows differ; first %d: got %s... want %s..." % (
        len(bad), len(want), bad[0], got[bad[0]][:60], want[bad[0]][:60])


def test_known_answers():
    ws = PoaWindowSet.from_lists([["ACGTACGTAC"] * 3, ["ACGTACGTAC", "ACGTACGTAC", "ACGAACGTAC"],
                                  ["ACGTACGTAC", "ACGTCGTAC", "ACGTACGTAC"], ["AAAA", "AATAA", "AATAA"], ["ACGT"],
                                  ["A", "C", "A"], ["GATTACA", "GATTACA", "GATTTACA", "GATTACA"]])
    p = make_params()
    got = consensus_host(p, ws)
    assert got == ["ACGTACGTAC", "ACGTACGTAC", "ACGTACGTAC", "AATAA", "ACGT", "A", "GATTACA"]
    diff(got, O.poa_oracle(p, ws))


def random_windows(seed, n, max_len, max_reads, alphabet="ACGTN"):
    rng = np.random.default_rng(seed)
    wins = []
    for _ in range(n):
        base = "".join(rng.choice(list(alphabet), int(rng.integers(5, max_len))))
        reads = []
        for _ in range(int(rng.integers(1, max_reads + 1))):
            r = [c for c in base if rng.random() > 0.08]
            r = [c if rng.random() > 0.08 else "ACGT"[int(rng.integers(4))] for c in r]
            for _ in range(int(rng.integers(0, 4))):
                r.insert(int(rng.integers(0, len(r) + 1)), "ACGT"[int(rng.integers(4))])
            reads.append("".join(r) or "A")
        wins.append(reads)
    return PoaWindowSet.from_lists(wins)


@pytest.mark.parametrize("max_len", [40, 200, 300, 600, 900])
def test_random_small_windows_every_column_class(max_len):
    """Sequence lengths across every columns-per-lane class (4/8/12/16) incl. ragged and tiny reads."""
    ws = random_windows(max_len, 24, max_len, 7)
    p = make_params()
    diff(consensus_host(p, ws), O.poa_oracle(p, ws, 4))


def test_generated_windows():
    ws = gen_poa(48, 4001)
    p = make_params()
    diff(consensus_host(p, ws), O.poa_oracle(p, ws, 8))


def test_affine_scoring_and_shard_equivalence():
    ws = gen_poa(16, 9)
    pa = make_params(o2=4, e2=2)
    full = consensus_host(pa, ws)
    diff(full, O.poa_oracle(pa, ws, 8))
    assert consensus_host(pa, ws.take(0, 5)) + consensus_host(pa, ws.take(5, 16)) == full


@pytest.mark.parametrize("kw", [dict(o1=0, e1=2), dict(o1=0, e1=4, o2=2, e2=1), dict(m=1, x=1, o1=0, e1=1)])
def test_linear_gap_subtype(kw, monkeypatch):
    """g >= e is spoa's linear subtype (the driver's -o 0,... : msa_spoa_omp.cpp:170-196): the device runs it as the affine DP
    with e = q = c = g and a backtrack of single cells (PoaScore::linear); consensus == the oracle's one-matrix restatement,
    through the window kernel, the team kernel (small jobs) and the long-window launch."""
    p = make_params(**kw)
    assert p.g >= p.e
    ws = gen_poa(24, 31)
    diff(consensus_host(p, ws), O.poa_oracle(p, ws, 8))
    monkeypatch.setenv("GBX_POA_TEAM", "0")
    diff(consensus_host(p, ws), O.poa_oracle(p, ws, 8))
    monkeypatch.delenv("GBX_POA_TEAM")
    long_ws = PoaWindowSet.from_lists([[s * 3 for s in ws.window(w)[:6]] for w in range(3)])      # sequences over 512 bases
    diff(consensus_host(p, long_ws), O.poa_oracle(p, long_ws, 8))


def _long_read_window(length, n_reads, seed):
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 4, length)
    reads = []
    for _ in range(n_reads):
        r = base[rng.random(length) > 0.03]                      # deletions
        sub = rng.random(r.size) < 0.05
        r = np.where(sub, rng.integers(0, 4, r.size), r)
        ins = np.sort(rng.integers(0, r.size, max(1, length // 40)))
        r = np.insert(r, ins, rng.integers(0, 4, ins.size))
        reads.append("".join("ACGT"[c] for c in r))
    return reads


def test_wide_path_on_ordinary_windows(monkeypatch):
    """The int32 path (poa_wide_kernel: column-block DP and five-plane traceback on 32-bit cells, global-memory sort) forced
    onto windows the int16 paths usually take: same consensus as the oracle, with fewer slots than windows (the cursor hands a
    slot its next window) and with the linear subtype."""
    monkeypatch.setenv("GBX_POA_FORCE_WIDE", "1")
    ws = gen_poa(20, 57)
    p = make_params()
    want = O.poa_oracle(p, ws, 8)
    diff(consensus_host(p, ws), want)
    monkeypatch.setenv("GBX_POA_WIDE_SLOTS", "3")
    diff(consensus_host(p, ws), want)
    pl = make_params(o1=0, e1=2)
    diff(consensus_host(pl, ws), O.poa_oracle(pl, ws, 8))


def test_long_read_window_takes_the_wide_path_beside_ordinary_windows():
    """Reads of 6 kb: the plan's worst-case score leaves the int16 range (poa_scores_fit_int16), where the entry used to
    return GBX_ERR_UNSUPPORTED and spoa switches to 32-bit lanes; now that window runs on int32 cells while the ordinary
    windows of the same call keep their kernels.  Consensus of every window == the oracle's."""
    ws0 = gen_poa(6, 58)
    wins = [ws0.window(w) for w in range(3)] + [_long_read_window(6000, 4, 1)] + [ws0.window(w) for w in range(3, 6)]
    ws = PoaWindowSet.from_lists(wins)
    p = make_params()
    got = consensus_host(p, ws)
    diff(got, O.poa_oracle(p, ws, 8))
    assert len(got[3]) > 5000


def test_window_of_12_kb_reads():
    """a window of 12-kb reads (three of them: the oracle's five int planes are 3.4 GB) against the oracle."""
    ws = PoaWindowSet.from_lists([_long_read_window(12000, 3, 2)])
    p = make_params()
    got = consensus_host(p, ws)
    diff(got, O.poa_oracle(p, ws, 2))
    assert len(got[0]) > 11000


def test_host_entry_staged_transfers(monkeypatch):
    """Staged host path (pinned slabs, upload workers, downloader) against the oracle; see test_chain_gpu."""
    ws = gen_poa(24, 616)
    p = make_params()
    want = O.poa_oracle(p, ws, 8)
    monkeypatch.setenv("GBX_HOST_STAGE_MIN", "0")
    diff(consensus_host(p, ws), want)
    monkeypatch.setenv("GBX_HOST_PAGEABLE", "1")
    diff(consensus_host(p, ws), want)


def test_long_insertion_chain_deeper_than_the_lds_stack():
    """A read with a 900-base insertion adds a chain of 900 fresh nodes; the topological sort walks it back node by
    node, deeper than its LDS stack (256 entries), and has to fall back to the global-memory sort.  Same consensus
    as the oracle, also for the reads added afterwards."""
    rng = np.random.default_rng(11)
    base = "".join(rng.choice(list("ACGT"), 300))
    ins = "".join(rng.choice(list("ACGT"), 900))
    long_read = base[:150] + ins + base[150:]
    ws = PoaWindowSet.from_lists([[base, long_read, base, long_read, base[:290]],
                                  [long_read, base, base]])
    p = make_params()
    diff(consensus_host(p, ws), O.poa_oracle(p, ws))


def test_deep_unrelated_window_is_redone_with_a_larger_graph():
    """The plan's node capacity is 6 x the longest read + 256 (typical windows need about 3x).  A window of 120
    unrelated 200-base reads grows far past it; the host entry re-runs exactly the overflowing windows with room
    for their worst case (spoa itself has no limit, msa_spoa_omp.cpp:237-252) and the other windows keep their
    first-pass results."""
    rng = np.random.default_rng(5)
    deep = ["".join(rng.choice(list("ACGT"), 200)) for _ in range(120)]
    base = "".join(rng.choice(list("ACGT"), 180))
    ws = PoaWindowSet.from_lists([[base] * 4, deep, [base[:170], base, base[5:]]])
    p = make_params()
    want = O.poa_oracle(p, ws, 4)
    assert len(want[1]) > 0
    diff(consensus_host(p, ws), want)


def test_windows_with_long_sequences_take_the_second_launch():
    """A few sequences of more than 512 bases must not size every slot of the job: their windows go to a second launch with
    five-plane slots of their own (column-block DP), the others keep two-plane slots.  Results are the oracle's for both."""
    import ctypes as C
    from genomicsbench_amd.poa import PoaPlan
    rng = np.random.default_rng(12)
    def window(L, n):
        base = "".join(rng.choice(list("ACGT"), L))
        out = []
        for _ in range(n):
            s = list(base)
            for k in rng.choice(L, max(L // 20, 1), replace=False):
                s[k] = "ACGT"[rng.integers(4)]
            cut = int(rng.integers(0, 12))
            out.append("".join(s)[cut:])
        return out
    lens = [300, 520, 180, 700, 400, 513, 512, 90, 1100, 250, 333, 512]
    windows = [window(L, 6 + i % 5) for i, L in enumerate(lens)]
    windows[1][0] = windows[1][0][:500]                       # a long window also holds short sequences
    ws = PoaWindowSet.from_lists(windows)
    plan = PoaPlan()
    N.check(N.lib().gbx_poa_plan_host(ws.n_windows, N.ptr(ws.win_first_seq), N.ptr(ws.seq_len), C.byref(plan)))
    n_long = sum(any(len(s) > 512 for s in w) for w in windows)
    assert plan.n_long_windows == n_long >= 3 and plan.long_slots == n_long and plan.n_slots == ws.n_windows - n_long and plan.n_windows == ws.n_windows
    p = make_params()
    want = O.poa_oracle(p, ws, 4)
    assert consensus_host(p, ws) == want
    import torch
    d = DevicePoaWindowSet(ws, torch.device("cuda:0"))
    d.run(p, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert d.results() == want
    # every window long / no window long
    for sub in ([windows[3], windows[8]], [windows[0], windows[2], windows[7]]):
        w2 = PoaWindowSet.from_lists(sub)
        assert consensus_host(p, w2) == O.poa_oracle(p, w2, 2)


# ---- the team kernel (round 5, poa_kernels.hip: a window per workgroup of four wavefronts, the DP's rows as a dataflow over a
# shared LDS ring).  Jobs of up to four windows per CU take it by themselves, i.e. every small job of this file; here the three
# ways a job can be split over the kernels are forced and compared: team everywhere (default), no team at all
# (GBX_POA_TEAM=0: one wavefront per window, the kernel of 'large'), and team for the long windows only (GBX_POA_TEAM_MAX=0).
# GBX_POA_SERIAL_FORM / GBX_POA_TEAM_FORM = 2: the same kernels with traceback, add_alignment, the sort and the consensus in one
# out-of-line function (poa_serial_call) - measured no faster (profiles/r05m_poa_serial_out_of_line_ab.txt), kept selectable.
@pytest.mark.parametrize("env", [{}, {"GBX_POA_TEAM": "0"}, {"GBX_POA_TEAM_MAX": "0"}, {"GBX_POA_TEAM_MAX": "5"},
                                 {"GBX_POA_TEAM_FORM": "2", "GBX_POA_SERIAL_FORM": "2"}, {"GBX_POA_TEAM": "0", "GBX_POA_SERIAL_FORM": "2"},
                                 {"GBX_POA_TEAM_MAX": "5", "GBX_POA_TEAM_FORM": "2", "GBX_POA_SERIAL_FORM": "2"}])
def test_team_kernel_equals_the_window_kernel(monkeypatch, env):
    p = make_params()
    rng = np.random.default_rng(5)
    def window(L, n):
        base = "".join(rng.choice(list("ACGT"), L))
        return ["".join(c if rng.random() > 0.07 else "ACGT"[rng.integers(4)] for c in base)[int(rng.integers(0, 9)):] for _ in range(n)]
    sets = [gen_poa(40, 4001), random_windows(17, 24, 300, 7), random_windows(18, 30, 520, 9),
            PoaWindowSet.from_lists([window(L, 5 + i % 4) for i, L in enumerate([300, 530, 512, 760, 90, 513, 1040, 400])]),
            PoaWindowSet.from_lists([["ACGTACGTAC"] * 3, ["A", "C", "A"], ["ACGT"], [], ["GATTACA", "GATTACA", "GATTTACA", "GATTACA"]])]
    want = [O.poa_oracle(p, ws, 8) for ws in sets]
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    for ws, w in zip(sets, want):
        diff(consensus_host(p, ws), w)


@pytest.mark.parametrize("env", [{"GBX_POA_LDS_NCAP": "640"}, {"GBX_POA_LDS_NCAP": "640", "GBX_POA_TEAM": "0"},
                                 {"GBX_POA_LDS_NCAP": "1200", "GBX_POA_TEAM": "0", "GBX_POA_SERIAL_FORM": "2"},
                                 {"GBX_POA_LDS_NCAP": "640", "GBX_POA_LOCKSTEP": "1"},
                                 {"GBX_POA_OCC": "4", "GBX_POA_MAX_WAVES": "16", "GBX_POA_TEAM": "0"}])
def test_windows_that_outgrow_the_sorts_lds_arrays(monkeypatch, env):
    """The topological sort's per-node LDS arrays hold what fits a window's share of the CU (PoaArgs::lds_ncap), not the graph's
    capacity: a window that may outgrow them switches to the global-memory sort for the rest of its life, the others keep the LDS
    sort.  GBX_POA_LDS_NCAP (test aid) makes that happen early and in the middle of windows; the last case is the sixteen-windows-
    per-CU instance, whose arrays hold 2 352 nodes."""
    p = make_params()
    sets = [gen_poa(24, 4001), random_windows(27, 16, 520, 9),
            PoaWindowSet.from_lists([["ACGTACGTAC"] * 3, ["A", "C", "A"], ["GATTACA", "GATTACA", "GATTTACA", "GATTACA"]])]
    want = [O.poa_oracle(p, ws, 8) for ws in sets]
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    for ws, w in zip(sets, want):
        diff(consensus_host(p, ws), w)


# ---- the lock-step form (poa_kernels.hip: a slot per window, one launch per phase and sequence index).  Jobs with more
# windows than the chip holds wavefronts take it by themselves ('large': tests/test_fullsize_gpu.py); GBX_POA_LOCKSTEP=1
# forces it on the small jobs here.
@pytest.mark.parametrize("tb_serial,occ", [("0", "5"), ("0", "6"), ("1", "5"), ("1", "6")])
def test_lockstep_form_equals_the_window_kernel(monkeypatch, tb_serial, occ):
    p = make_params()
    sets = [gen_poa(48, 4001), random_windows(7, 24, 300, 7), random_windows(8, 30, 500, 9),
            PoaWindowSet.from_lists([["ACGTACGTAC"] * 3, ["A", "C", "A"], ["ACGT"], ["GATTACA", "GATTACA", "GATTTACA", "GATTACA"]])]
    want = [O.poa_oracle(p, ws, 8) for ws in sets]
    mono = [consensus_host(p, ws) for ws in sets]
    monkeypatch.setenv("GBX_POA_LOCKSTEP", "1")
    monkeypatch.setenv("GBX_POA_TB_SERIAL", tb_serial)
    monkeypatch.setenv("GBX_POA_DP_OCC", occ)
    for ws, w, m in zip(sets, want, mono):
        got = consensus_host(p, ws)
        diff(got, w)
        assert got == m


def test_lockstep_form_edge_paths(monkeypatch):
    """Under the lock-step form: the sort's fallback to global memory (state must survive the launches without root flags),
    windows with long sequences beside it (second launch keeps the window kernel), the node-capacity redo of the host entry,
    windows of one sequence and ragged sequence counts (a window leaves the launches when it runs out of sequences)."""
    monkeypatch.setenv("GBX_POA_LOCKSTEP", "1")
    p = make_params()
    rng = np.random.default_rng(11)
    base = "".join(rng.choice(list("ACGT"), 300))
    ins = "".join(rng.choice(list("ACGT"), 900))
    long_read = base[:150] + ins + base[150:]
    ws = PoaWindowSet.from_lists([[base, long_read, base, long_read, base[:290]], [long_read, base, base], [base], [base[:40]] * 9])
    diff(consensus_host(p, ws), O.poa_oracle(p, ws))
    rng = np.random.default_rng(5)
    deep = ["".join(rng.choice(list("ACGT"), 200)) for _ in range(120)]
    b2 = "".join(rng.choice(list("ACGT"), 180))
    ws = PoaWindowSet.from_lists([[b2] * 4, deep, [b2[:170], b2, b2[5:]]])
    diff(consensus_host(p, ws), O.poa_oracle(p, ws, 4))
    for tb in ("0", "1"):
        monkeypatch.setenv("GBX_POA_TB_SERIAL", tb)
        ws = random_windows(21, 40, 700, 6)                                     # some windows over 512 columns: second launch
        diff(consensus_host(p, ws), O.poa_oracle(p, ws, 4))
    # the device entry re-run on the same workspace (slots hold the previous run's headers and state bytes)
    import torch
    ws = gen_poa(32, 77)
    d = DevicePoaWindowSet(ws, torch.device("cuda:0"))
    want = O.poa_oracle(p, ws, 8)
    for _ in range(2):
        d.run(p, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert d.results() == want
