"""Host threads calling different kernels' C-ABI entries at the same time (they share the device's side streams,
each takes its own lane): same results as one after the other."""
import threading

import numpy as np
import pytest

from genomicsbench_amd.bsw import extend_host, make_params as bsw_params
from genomicsbench_amd.chain import chain_host
from genomicsbench_amd.abea import align_host
from genomicsbench_amd.datagen import gen_abea, gen_bsw, gen_chain, gen_phmm, gen_poa
from genomicsbench_amd.phmm import forward_host
from genomicsbench_amd.poa import consensus_host, make_params as poa_params

pytestmark = pytest.mark.gpu


def test_mixed_kernels_from_concurrent_threads():
    pb, pp = bsw_params(), poa_params()
    jobs = [
        ("bsw-a", lambda b=gen_bsw(40000, 71): extend_host(pb, b)),
        ("bsw-b", lambda b=gen_bsw(3000, 72): extend_host(pb, b)),
        ("chain", lambda c=gen_chain(200, 73): chain_host(*c)),
        ("phmm", lambda s=gen_phmm(12, 74): forward_host(s)),
        ("poa", lambda w=gen_poa(40, 75): consensus_host(pp, w)),
        # 3.8 M events: the staged path (means gathered by the upload workers, threaded copy-out); the slots behind a
        # read's n_pairs are not part of the result
        ("abea", lambda r=gen_abea(300, 76): (lambda o, n: (r.split_pairs(o, n), n))(*align_host(r))),
    ]
    want = [f() for _, f in jobs]

    def same(a, b):
        if isinstance(a, (list, tuple)):
            return len(a) == len(b) and all(same(x, y) for x, y in zip(a, b))
        if isinstance(a, np.ndarray):
            return np.array_equal(a, b)
        return a == b

    for _ in range(4):
        got = [None] * len(jobs)

        def work(k):
            got[k] = jobs[k][1]()

        th = [threading.Thread(target=work, args=(k,)) for k in range(len(jobs))]
        for t in th:
            t.start()
        for t in th:
            t.join()
        for k, (name, _) in enumerate(jobs):
            assert same(got[k], want[k]), "%s differs when run concurrently" % name


def test_host_cache_release_between_calls():
    """gbx_host_release() drops the device blocks the host entries keep between calls; the next call allocates
    afresh and gives the same results."""
    from genomicsbench_amd import _native as N
    pb = bsw_params()
    b = gen_bsw(20000, 81)
    first = extend_host(pb, b)
    N.check(N.lib().gbx_host_release())
    N.check(N.lib().gbx_host_prepare())
    assert np.array_equal(extend_host(pb, b), first)
    N.check(N.lib().gbx_host_release())
