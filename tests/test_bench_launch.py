"""bench.py --gpus N starts its own ranks: the launcher is proven here without GPUs.

The parent must not import torch.cuda / libgbx, must start N children with the rank environment, relay rank 0's JSON line
and return a non-zero code when any rank fails."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    return env


def test_dry_run_prints_one_child_per_gpu():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--steps", "3", "--warmup", "1", "--launch-dry-run"],
                       capture_output=True, text=True, env=_env(), timeout=120)
    assert r.returncode == 0, r.stderr
    plan = json.loads(r.stdout.strip().splitlines()[-1])["launch"]
    assert len(plan) == 8
    ports = {p["env"]["MASTER_PORT"] for p in plan}
    assert len(ports) == 1
    for rank, p in enumerate(plan):
        assert p["argv"][1] == BENCH and p["argv"][2:] == ["--gpus", "8", "--steps", "3", "--warmup", "1"]
        e = p["env"]
        assert (e["RANK"], e["LOCAL_RANK"], e["WORLD_SIZE"], e["MASTER_ADDR"]) == (str(rank), str(rank), "8", "127.0.0.1")
        assert e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_parent_never_loads_torch_or_libgbx():
    code = ("import sys, runpy\n"
            "sys.argv = ['bench.py', '--gpus', '2', '--launch-dry-run']\n"
            "try:\n    runpy.run_path(%r, run_name='__main__')\nexcept SystemExit as e:\n    assert not e.code, e.code\n"
            "assert 'torch' not in sys.modules and 'genomicsbench_amd._native' not in sys.modules, 'parent touched torch/libgbx'\n"
            % BENCH)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=_env(), timeout=120)
    assert r.returncode == 0, r.stderr + r.stdout


def test_self_launch_two_ranks_relays_rank0_line():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--launch-echo"], capture_output=True, text=True, env=_env(),
                       timeout=300)
    assert r.returncode == 0, r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                     # rank 1's stdout is not relayed
    line = json.loads(lines[0])
    assert line["launch_echo"] and line["world"] == 2 and line["rank_sum"] == 1.0
    # the echo also runs the scatter / gather of genomicsbench_amd/shard.py in capped pieces, over RCCL when both ranks have a
    # device of their own (backend "nccl": the first contact of the judged path), over gloo here
    assert line["scatter_gather_ok"] is True and line["backend"] in ("gloo", "nccl")


def test_rank_stderr_reaches_the_parent_with_its_rank():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--launch-echo", "--launch-echo-fail", "1"],
                       capture_output=True, text=True, env=dict(_env(), GBX_ECHO_NOISE="1"), timeout=300)
    assert "[rank 1] launch-echo: rank 1 was told to fail" in r.stderr, r.stderr[-1500:]


def test_failing_rank_fails_the_launch():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--launch-echo", "--launch-echo-fail", "1"],
                       capture_output=True, text=True, env=_env(), timeout=300)
    assert r.returncode == 7, (r.returncode, r.stderr)


def test_world_size_mismatch_is_an_error_not_an_assert():
    env = dict(_env(), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2"], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=1 but --gpus 2" in r.stderr


def test_torchrun_launch_still_works():
    """The driver's own launch shape for N > 1 (python -m torch.distributed.run ... bench.py --gpus N)."""
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29731", BENCH, "--gpus", "2", "--launch-echo"],
                       capture_output=True, text=True, env=_env(), timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["world"] == 2 and line["omp_num_threads"] == str(os.cpu_count())


def test_short_form_of_the_other_kernels_records():
    """The all-kernels line carries chain / phmm / poa / abea / fmi in short form (the full records go to stderr): no nulls, no
    keys that repeat the headline's, long strings cut, the three longest stages; the sharding's own figures stay for N > 1."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    rec = {"metric": "x", "value": 1.23456789, "unit": "GCUPS", "n_gpus": 2, "scaling": "weak", "vs_baseline": None, "ms_per_step": 2.0,
           "config": {"workload": "w" * 300, "parallelism": "p", "calls": 7},
           "roofline": {"bound": "hbm", "traffic": None, "traffic_source": "s", "frac": 0.123456789, "valu": {"a": 1}},
           "kernels_ms": {"a": 1.0, "b": 5.0, "c": 3.0, "d": 0.1}, "shard_units": [5, 5], "scatter_ms": 1.5,
           "config4_strong": {"scaling": "strong", "value": 2.0, "shard_units": [1, 2]}}
    c = m.compact_entry(rec)
    assert "n_gpus" not in c and "vs_baseline" not in c and "shard_units" not in c and "scatter_ms" not in c
    assert c["value"] == 1.2346 and len(c["config"]["workload"]) == 100 and c["config"]["calls"] == 7 and "parallelism" not in c["config"]
    assert c["roofline"] == {"bound": "hbm", "frac": 0.12346} and list(c["kernels_ms"]) == ["b", "c", "a"]
    k = m.compact_entry(rec, keep=("shard_units", "scatter_ms"))
    assert k["shard_units"] == [5, 5] and k["scatter_ms"] == 1.5 and k["config4_strong"]["shard_units"] == [1, 2] and k["config4_strong"]["strong_scaling"]
    assert len(json.dumps(c)) < len(json.dumps(rec))


def test_config4_predicted_survives_the_short_form_and_predict_shards_is_a_one_gpu_flag():
    """`kernels.poa.config4_predicted` (the eight shards of BASELINE config 4's cut, each alone on one GPU) keeps its per-shard times and
    the prediction in the short form of the line; `--predict-shards` with more than one rank is refused before anything touches a GPU."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod2", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    rec = {"metric": "poa_large_gcups", "value": 450.0, "ms_per_step": 200.0,
           "config4_predicted": {"what": "x" * 200, "parts": 8, "shard_units": [750] * 8, "shard_ms": [64.0 + k / 10 for k in range(8)],
                                 "whole_job_ms_1gpu": 200.0, "predicted_ms_per_step": 64.7, "predicted_speedup": 3.0912345,
                                 "verified": "front 8 units of every shard vs oracle: identical"}}
    c = m.compact_entry(rec)["config4_predicted"]
    assert len(c["shard_ms"]) == 8 and c["predicted_speedup"] == 3.0912 and c["predicted_ms_per_step"] == 64.7 and "what" not in c
    assert "identical" in c["verified"]
    ap_src = open(os.path.join(ROOT, "bench.py")).read()
    assert "--predict-shards is a one-GPU measurement" in ap_src
