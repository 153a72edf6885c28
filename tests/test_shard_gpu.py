"""Shard equivalence on the GPU: split -> (pack, H2D, unpack) -> compute per shard -> concatenate == the unsharded
device run == the oracle, for every kernel, through the same shard builders and packed-message layout that
bench.py's multi-GPU mode scatters over RCCL (the ranks are played one after another by the one GPU)."""
import numpy as np
import pytest

from conftest import has_gpu

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")]


def _to_device(arrays, dev):
    import torch
    from genomicsbench_amd import shard as S
    buf, meta = S.pack_arrays(arrays)
    return S.unpack_tensor(torch.from_numpy(buf).to(dev), meta)


def _sync_stream():
    import torch
    return torch.cuda.current_stream().cuda_stream


@pytest.mark.parametrize("parts", [2, 5])
def test_bsw_split_compute_concat(parts):
    import torch
    from genomicsbench_amd import shard as S
    from genomicsbench_amd.bsw import DeviceBswBatch, make_params
    from genomicsbench_amd.datagen import gen_bsw
    from oracle import oracle_py as O
    dev, p = torch.device("cuda:0"), make_params()
    full = gen_bsw(60_000, 77)
    got = []
    for sh in S.bsw_shards(full, parts):
        d = DeviceBswBatch.from_tensors(_to_device(S.bsw_to_arrays(sh), dev), dev)
        d.run(p, _sync_stream())
        torch.cuda.synchronize()
        got.append(d.results())
    assert np.array_equal(np.concatenate(got), O.bsw_oracle(p, full, 8))


@pytest.mark.parametrize("parts", [2, 3])
def test_chain_split_compute_concat(parts):
    import torch
    from genomicsbench_amd import shard as S
    from genomicsbench_amd.chain import DeviceChainBatch
    from genomicsbench_amd.datagen import gen_chain
    from oracle import oracle_py as O
    dev = torch.device("cuda:0")
    full = gen_chain(120, 2001, first=500)
    want = O.chain_oracle(*full, nthreads=8)
    whole = DeviceChainBatch(*full, dev)
    whole.run(_sync_stream())
    torch.cuda.synchronize()
    got = [[] for _ in range(4)]
    for sh in S.chain_shards(*full, parts):
        d = DeviceChainBatch.from_tensors(_to_device(S.chain_to_arrays(sh), dev), dev)
        d.run(_sync_stream())
        torch.cuda.synchronize()
        for f, a in zip(got, d.results()):
            f.append(a)
    for f in range(4):
        cat = np.concatenate(got[f])
        assert np.array_equal(cat, want[f]) and np.array_equal(cat, whole.results()[f])


@pytest.mark.parametrize("parts", [2, 4])
def test_phmm_split_compute_concat(parts):
    import torch
    from genomicsbench_amd import shard as S
    from genomicsbench_amd.datagen import gen_phmm
    from genomicsbench_amd.phmm import DevicePhmmBatchSet
    from oracle import oracle_py as O
    dev = torch.device("cuda:0")
    full = gen_phmm(160, 3001, first=40)
    want = O.phmm_oracle(full, 8)
    got = []
    for sh in S.phmm_shards(full, parts):
        d = DevicePhmmBatchSet.from_tensors(_to_device(S.phmm_to_arrays(sh), dev), dev)
        assert d.n_pairs == sh.n_pairs
        # the pair list rebuilt on the device equals the host's
        assert np.array_equal(d.pair_read[:d.n_pairs].cpu().numpy(), sh.pair_read)
        assert np.array_equal(d.pair_hap[:d.n_pairs].cpu().numpy(), sh.pair_hap)
        d.run(_sync_stream())
        torch.cuda.synchronize()
        got.append(d.results())
    got = np.concatenate(got)
    fin = np.isfinite(want)
    assert np.array_equal(np.isfinite(got), fin)
    # tolerance: 1e-5 relative on the log10 likelihood, floored at |want| = 1
    assert np.all(np.abs(got[fin] - want[fin]) <= 1e-5 * np.maximum(1.0, np.abs(want[fin])))


@pytest.mark.parametrize("parts", [2, 3])
def test_poa_split_compute_concat(parts):
    import torch
    from genomicsbench_amd import shard as S
    from genomicsbench_amd.datagen import gen_poa
    from genomicsbench_amd.poa import DevicePoaWindowSet, make_params
    from oracle import oracle_py as O
    dev, p = torch.device("cuda:0"), make_params()
    full = gen_poa(40, 4001, first=100)
    want = O.poa_oracle(p, full, 8)
    got = []
    for sh in S.poa_shards(full, parts):
        d = DevicePoaWindowSet.from_tensors(_to_device(S.poa_to_arrays(sh), dev), dev)
        d.run(p, _sync_stream())
        torch.cuda.synchronize()
        got += d.results()
    assert got == want


@pytest.mark.parametrize("parts", [2, 3])
def test_abea_split_compute_concat(parts):
    import torch
    from genomicsbench_amd import shard as S
    from genomicsbench_amd.abea import DeviceAbeaReadSet
    from genomicsbench_amd.datagen import gen_abea
    from oracle import oracle_py as O
    dev = torch.device("cuda:0")
    full = gen_abea(36, 5001, first=50)
    want = full.split_pairs(*O.abea_oracle(full, 8))
    got = []
    for sh in S.abea_shards(full, parts):
        d = DeviceAbeaReadSet.from_tensors(_to_device(S.abea_to_arrays(sh), dev), dev)
        d.run(_sync_stream())
        torch.cuda.synchronize()
        got += sh.split_pairs(*d.results())
    assert len(got) == len(want) and all(np.array_equal(g, w) for g, w in zip(got, want))


@pytest.mark.parametrize("parts", [2, 3])
def test_fmi_split_compute_concat(parts):
    import torch
    from genomicsbench_amd import shard as S
    from genomicsbench_amd.datagen import gen_fmi_genome, gen_fmi_reads
    from genomicsbench_amd.fmi import DeviceFmi, build_index
    from oracle import oracle_py as O
    dev = torch.device("cuda:0")
    g = gen_fmi_genome(100_000, 6001)
    idx = build_index(g)
    full = gen_fmi_reads(g, 2000, 6002, first=500)
    wo, woff = O.fmi_oracle(idx, full, nthreads=8)
    didx = idx.to(dev)
    recs, offs, first = [], [np.zeros(1, np.int64)], 0
    for sh in S.fmi_shards(full, parts):
        t = _to_device(S.fmi_to_arrays(sh), dev)
        d = DeviceFmi.from_tensors(didx, (t["enc"], t["read_off"], t["read_len"]), dev)
        d.run(_sync_stream())
        torch.cuda.synchronize()
        r, o = d.results()
        r["rid"] += first                                    # shard-local rids, like the batch-local ones of fmi.cpp:270-273
        recs.append(r)
        offs.append(o[1:] + offs[-1][-1])
        first += sh.n_reads
    got, goff = np.concatenate(recs), np.concatenate(offs)
    assert np.array_equal(goff, woff)
    for f in ("rid", "m", "n", "k", "l", "s"):
        assert np.array_equal(got[f], wo[f]), f


@pytest.mark.parametrize("kind,size", [("poa", 96), ("chain", 60)])
def test_bench_predict_shards_on_one_gpu(kind, size):
    """`bench.py --kernel K --predict-shards P` (BASELINE config 4 measured on one GPU: the P shards of the N-GPU cut, each run alone
    and checked against the oracle; the slowest one is the predicted N-GPU step) on a small job."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--kernel", kind, "--predict-shards", "3", "--size", str(size), "--steps", "1", "--warmup", "1",
                        "--verify-units", "4"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    p = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])["config4_predicted"]
    assert p["parts"] == 3 and sum(p["shard_units"]) == size and len(p["shard_ms"]) == 3
    assert abs(p["predicted_ms_per_step"] - max(p["shard_ms"])) < 1e-2 and p["predicted_speedup"] > 0 and "identical" in p["verified"] and "DIFFER" not in p["verified"]
