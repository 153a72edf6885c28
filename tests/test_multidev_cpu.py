"""The multi-device layer under the *_host entries (csrc/host_multi.h), as far as it runs without a GPU: the cut rule
against genomicsbench_amd/shard.py, the device-set API, and that an entry asked for several devices still reports
argument errors first and GBX_ERR_NO_DEVICE otherwise (no CPU fallback)."""
import ctypes as C
import os

import numpy as np
import pytest

from genomicsbench_amd import _native as N
from genomicsbench_amd import shard as S
from genomicsbench_amd.bsw import extend_host, make_params
from genomicsbench_amd.datagen import gen_bsw
from conftest import has_gpu


def c_split(costs, parts):
    costs = np.ascontiguousarray(costs, dtype=np.float64)
    cuts = np.zeros(parts + 1, dtype=np.int64)
    N.check(N.lib().gbx_split_by_cost(len(costs), N.ptr(costs), parts, N.ptr(cuts)))
    return [(int(cuts[k]), int(cuts[k + 1])) for k in range(parts)]


@pytest.mark.parametrize("n", [0, 1, 5, 1000, 65536, 200001, 3000000])
@pytest.mark.parametrize("parts", [1, 2, 3, 8])
def test_cut_rule_is_shard_py(n, parts):
    """Integer costs (cells, anchors, bases): the C++ drivers and bench.py cut a job at exactly the same units."""
    rng = np.random.default_rng(n * 31 + parts)
    costs = rng.integers(0, 40000, size=n).astype(np.float64)
    if n > 10:
        costs[rng.integers(0, n, size=n // 10)] = 0            # runs of zero-cost units
        costs[n // 3] = 5e7                                     # one unit heavier than a whole share
    assert c_split(costs, parts) == S.split_by_cost(costs, parts)


def test_cut_rule_edge_cases():
    assert c_split(np.zeros(7), 3) == S.split_by_cost(np.zeros(7), 3)             # nothing to balance
    assert c_split([5.0], 4) == S.split_by_cost([5.0], 4)                          # more shards than units
    assert c_split([-3.0, 2.0, 2.0], 2) == [(0, 2), (2, 3)]                        # negative costs count as 0
    b = gen_bsw(50000, 11)
    assert c_split(S.bsw_cost(b), 8) == S.split_by_cost(S.bsw_cost(b), 8)
    cuts = np.zeros(3, dtype=np.int64)
    assert N.lib().gbx_split_by_cost(4, None, 2, N.ptr(cuts)) == -1
    assert N.lib().gbx_split_by_cost(4, N.ptr(np.ones(4)), 0, N.ptr(cuts)) == -1


def test_device_set_api_bounds():
    L = N.lib()
    assert L.gbx_host_set_devices(-1) == -1
    assert L.gbx_host_set_devices(17) == -1
    assert L.gbx_host_set_devices(0) == 0
    if not has_gpu():
        assert L.gbx_host_devices() == 0
        assert L.gbx_host_set_devices(2) == -1                 # more devices than present
        assert b"2 devices asked for, 0 present" in L.gbx_last_error()


@pytest.mark.skipif(has_gpu(), reason="checks the no-device behaviour")
def test_multi_device_request_without_devices(monkeypatch):
    """GBX_GPUS=2 on a box without GPUs: bad arguments are still GBX_ERR_ARG (with the one-device texts), a valid call is
    GBX_ERR_NO_DEVICE - never a CPU result."""
    monkeypatch.setenv("GBX_GPUS", "2")
    monkeypatch.setenv("GBX_SHARD_MIN_UNITS", "1")
    b = gen_bsw(300, 5)
    with pytest.raises(N.GbxError) as e:
        extend_host(make_params(), b)
    assert e.value.code == -2
    bad = gen_bsw(300, 5)
    bad.idr[7] = -5
    with pytest.raises(N.GbxError) as e:
        extend_host(make_params(), bad)
    assert e.value.code == -1 and "pair 7" in str(e.value)
