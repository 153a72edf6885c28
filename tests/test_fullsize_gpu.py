"""Full-size parity: every kernel's BASELINE.json 'large' configuration, the whole job on the device entry, compared
with the CPU side unit for unit (bsw: oracle, all six fields of all 2 M pairs; chain: the compiled reference's
chain_dp on all 10 000 calls, oracle if the reference build did not travel; phmm: oracle on all ~10.9 M pairs of
the 20 000 batches; poa: oracle on all 6 000 windows; fmi: all 10 M reads on the 1-GB index, ten strata of 100 k reads
against the oracle).  The CPU side runs on the GPU box's host cores (OpenMP):
about a minute and a half in total there."""
import os

import numpy as np
import pytest

from conftest import has_gpu

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")]
CORES = os.cpu_count() or 1


def _stream():
    import torch
    return torch.cuda.current_stream().cuda_stream


def test_bsw_large_all_fields():
    import torch
    from genomicsbench_amd.bsw import DeviceBswBatch, make_params
    from genomicsbench_amd.datagen import gen_bsw
    from oracle import oracle_py as O
    b, p = gen_bsw(2_000_000, 1002), make_params()
    d = DeviceBswBatch(b, torch.device("cuda:0"))
    d.run(p, _stream())
    torch.cuda.synchronize()
    got, want = d.results(), O.bsw_oracle(p, b, CORES)
    bad = np.nonzero((got != want).any(axis=1))[0]
    assert not len(bad), "%d of %d pairs differ, first %d: got %s want %s" % (len(bad), b.n, bad[0], got[bad[0]], want[bad[0]])
    # a checksum of checksums per field, for the log
    print("bsw large: field sums", got.astype(np.int64).sum(axis=0).tolist())


@pytest.mark.parametrize("realistic", [False, True])
def test_chain_large_all_calls(realistic):
    """All 10 000 calls / 41 M anchors of chain 'large', and of the same call sizes with minimap2's structure inside a
    call (both strands, six reference ids: the calls are cut into independent jobs on the device)."""
    import torch
    from genomicsbench_amd.chain import DeviceChainBatch
    from genomicsbench_amd.datagen import gen_chain
    from oracle import oracle_py as O
    case = gen_chain(10_000, 2001, realistic=realistic)
    d = DeviceChainBatch(*case, torch.device("cuda:0"))
    d.run(_stream())
    torch.cuda.synchronize()
    want = O.chain_ref(*case, nthreads=CORES) if O.ref_lib("chain") is not None else O.chain_oracle(*case, nthreads=CORES)
    for name, g, w in zip(("score", "parent", "target", "peak"), d.results(), want):
        bad = np.nonzero(g != w)[0]
        assert not len(bad), "chain %s: %d of %d anchors differ, first at %d" % (name, len(bad), len(w), bad[0])


def test_phmm_large_all_pairs():
    """All ~10.9 M pairs of the 20 000 batches against the oracle (every unit class of the device path - the stream units
    of 1..8 rows per lane, the tiles, the fp64 redo - wherever in the job it occurs): about 100 s of oracle on the box."""
    import torch
    from genomicsbench_amd.datagen import gen_phmm
    from genomicsbench_amd.phmm import DevicePhmmBatchSet
    from oracle import oracle_py as O
    bs = gen_phmm(20_000, 3001)
    d = DevicePhmmBatchSet(bs, torch.device("cuda:0"))
    d.run(_stream())
    torch.cuda.synchronize()
    got = d.results()
    assert len(got) == bs.n_pairs
    worst = 0.0
    step = 2500                                                              # batches per oracle call: bounded host memory
    for b0 in range(0, 20_000, step):
        sub = bs.take_batches(b0, min(20_000, b0 + step))
        lo = int(bs.batch_pair_off[b0])
        g, want = got[lo:lo + sub.n_pairs], O.phmm_oracle(sub, CORES)
        fin = np.isfinite(want)
        assert np.array_equal(np.isfinite(g), fin), "batches %d..: finiteness differs" % b0
        err = np.abs(g[fin] - want[fin]) / np.maximum(np.abs(want[fin]), 1.0)    # 1e-5 relative, |want| floored at 1
        assert err.max() <= 1e-5, "max rel err %.3g at pair %d" % (err.max(), lo + int(np.argmax(err)))
        worst = max(worst, float(err.max()))
    print("phmm large: %d pairs, max relative error %.3g" % (bs.n_pairs, worst))


def test_poa_large_all_windows():
    import torch
    from genomicsbench_amd.datagen import gen_poa
    from genomicsbench_amd.poa import DevicePoaWindowSet, make_params
    from oracle import oracle_py as O
    ws, p = gen_poa(6_000, 4001), make_params()
    d = DevicePoaWindowSet(ws, torch.device("cuda:0"))
    d.run(p, _stream())
    torch.cuda.synchronize()
    got, want = d.results(), O.poa_oracle(p, ws, CORES)
    bad = [w for w in range(ws.n_windows) if got[w] != want[w]]
    assert not bad, "%d of %d windows differ, first %d" % (len(bad), ws.n_windows, bad[0])


def test_abea_large_all_reads():
    import torch
    from genomicsbench_amd.abea import DeviceAbeaReadSet
    from genomicsbench_amd.datagen import gen_abea
    from oracle import oracle_py as O
    rs = gen_abea(10_000, 5001)                      # the whole 'large' job of bench.py, longest reads included
    d = DeviceAbeaReadSet(rs, torch.device("cuda:0"))
    d.run(_stream())
    torch.cuda.synchronize()
    (go, gn), (wo, wn, cells) = d.results(), O.abea_oracle(rs, CORES, True)
    assert np.array_equal(gn, wn), "QC verdicts / pair counts differ for %d reads" % int((gn != wn).sum())
    assert d.cells(_stream()) == cells
    for r in range(rs.n_reads):
        a = 2 * int(rs.event_off[r])
        assert np.array_equal(go[a:a + int(wn[r])], wo[a:a + int(wn[r])]), "read %d" % r


def test_fmi_large_real_index_stratified():
    """fmi 'large' as bench.py runs it: all 10 M reads of 151 bases in one device call against the index of the 512-Mbp
    genome (1 GB of checkpoints: beyond L2 and Infinity Cache, where the kernel actually lives).  Checked against the
    oracle, every field of every SMEM: ten strata of 100 000 reads spread over the job (reads k M .. k M + 100 k), and the
    extension count of those strata run on their own."""
    import torch
    from genomicsbench_amd.datagen import gen_fmi_genome, gen_fmi_reads
    from genomicsbench_amd.fmi import DeviceFmi, build_index
    from oracle import oracle_py as O
    dev = torch.device("cuda:0")
    g = gen_fmi_genome(512 << 20, 6001)
    idx = build_index(g, device=dev)
    rs = gen_fmi_reads(g, 10_000_000, 6002)
    d = DeviceFmi(idx, rs, dev)
    d.run(_stream())
    torch.cuda.synchronize()
    got, goff = d.results()
    hidx = idx.host()
    assert goff[0] == 0 and goff[-1] == len(got) and np.all(np.diff(goff) >= 0)
    checked = 0
    for k in range(10):
        lo, hi = k * 1_000_000, k * 1_000_000 + 100_000
        wo, woff = O.fmi_oracle(hidx, rs.take(lo, hi), nthreads=CORES)
        a, b = int(goff[lo]), int(goff[hi])
        assert np.array_equal(goff[lo:hi + 1] - a, woff), "stratum %d: per-read SMEM counts differ" % k
        sub = got[a:b]
        assert np.array_equal(sub["rid"], wo["rid"] + lo), "stratum %d: rid" % k
        for f in ("m", "n", "k", "l", "s"):
            bad = np.nonzero(sub[f] != wo[f])[0]
            assert not len(bad), "fmi %s, stratum %d: %d of %d records differ, first at %d" % (f, k, len(bad), len(wo), bad[0])
        checked += len(wo)
    print("fmi large: %d SMEMs of 1 M reads in ten strata identical to the oracle (job: %d SMEMs, %d extensions)" % (checked, len(got), d.extensions()))
