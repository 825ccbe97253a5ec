"""Site log-likelihoods sent to the host before the caller asks (engine_abi.cpp sitePrefetchAfterRoot / sitePrefetchTake).

BeagleTreeLikelihood reads the site values back after EVERY evaluation (BeagleTreeLikelihood.java:1050).  After two such reads in a
row the engine copies the next root sum's site values into a pinned host buffer right behind the root's kernel; the getter then
finds them there.  What the getter returns has to be, bit for bit, what the stream-ordered download returns
(BEAGLE_MI355_NO_SITE_PREFETCH=1) — whatever comes between the root sum and the read, and never the values of an earlier sum."""
import os

import numpy as np
import pytest

import beast_mcmc_amd as bm
import helpers
from beast_mcmc_amd.inputs import substmodel
from beast_mcmc_amd.treelikelihood import BeagleTreeLikelihood, RESCALE_ALWAYS, RESCALE_DYNAMIC, RESCALE_NONE

pytestmark = pytest.mark.gpu
REL_TOL = 1e-10


def make(wl, prefetch, scheme, **kw):
    old = os.environ.get("BEAGLE_MI355_NO_SITE_PREFETCH")
    os.environ["BEAGLE_MI355_NO_SITE_PREFETCH"] = "0" if prefetch else "1"       # read when the instance is created
    try:
        tl = BeagleTreeLikelihood(wl, rescaling=scheme, delay_rescaling=False, **kw)
    finally:
        if old is None:
            del os.environ["BEAGLE_MI355_NO_SITE_PREFETCH"]
        else:
            os.environ["BEAGLE_MI355_NO_SITE_PREFETCH"] = old
    return tl, bm.beagle.Beagle.attach(tl)


def models(wl, n, seed):
    rng = np.random.default_rng(seed)
    if wl.state_count == 4:
        return [substmodel.gtr(rng.gamma(2.0, 1.0, size=6) + 0.1, wl.freqs) for _ in range(n)]
    return [None] * n


def btl_chain(tl, wl, steps, seed, read):
    """BeagleTreeLikelihood's protocol: a model move or a branch move, the sum, then (read(i): the caller's choice) the site values.
    -> [(lnL, site values or None)]"""
    rng = np.random.default_rng(seed)
    eig = models(wl, 3, seed + 1)
    t_, n_ = wl.tree.tip_count, wl.tree.node_count
    height = np.array(wl.tree.height, dtype=float)
    out = []
    for i in range(steps):
        tl.storeState()
        if i % 3 != 1 or n_ - t_ < 2:
            if eig[0] is not None:
                tl.set_substitution_model(eig[i % 3], wl.freqs)
            tl.makeDirty()
        else:
            node = int(rng.integers(t_, n_))
            while wl.tree.parent[node] < 0:
                node = int(rng.integers(t_, n_))
            lo = max(height[int(wl.tree.left[node])], height[int(wl.tree.right[node])])
            hi = height[wl.tree.parent[node]]
            height[node] = lo + (hi - lo) * float(rng.uniform(0.1, 0.9))
            tl.set_node_height(node, float(height[node]))
        v = tl.getLogLikelihood()
        out.append((v, tl.getSiteLogLikelihoods().copy() if read(i) else None))
    return out


@pytest.mark.parametrize("S,C,T,P", [(4, 4, 40, 1000), (4, 1, 9, 127), (4, 3, 25, 4097), (4, 8, 6, 1), (20, 4, 12, 300), (61, 2, 8, 70), (4, 4, 30, 20000)])
@pytest.mark.parametrize("scheme", [RESCALE_NONE, RESCALE_DYNAMIC, RESCALE_ALWAYS])
def test_prefetched_site_values_equal_the_download_bit_for_bit(S, C, T, P, scheme, oracle_lib):
    every = lambda i: True
    runs = []
    for prefetch in (True, False):
        wl = helpers.random_workload(T, P, S, C, seed=900 + S + C + T)
        tl, raw = make(wl, prefetch, scheme)
        runs.append((btl_chain(tl, wl, 9, 11, every), raw.sitePrefetchCount()))
        tl.close()
    (a, na), (b, nb) = runs
    # the first two reads switch it on: the third sum is the first whose values travel ahead of the read (a sum that underflows and is
    # repeated with new scale factors — DYNAMIC — is a sum nobody read the values of: two more reads switch it on again)
    assert nb == 0 and (na == len(a) - 2 if scheme == RESCALE_NONE else 1 <= na <= len(a) - 2), (na, nb)
    for (va, sa), (vb, sb) in zip(a, b):
        assert va == vb
        assert np.array_equal(sa, sb)
    wl = helpers.random_workload(T, P, S, C, seed=900 + S + C + T)
    o = BeagleTreeLikelihood(wl, library=oracle_lib, rescaling=scheme, delay_rescaling=False)
    for (va, sa), (vo, so) in zip(a, btl_chain(o, wl, 9, 11, every)):
        assert abs(va - vo) <= REL_TOL * abs(vo)
        np.testing.assert_allclose(sa, so, rtol=REL_TOL, atol=0)
    o.close()


def test_a_sum_nobody_reads_switches_it_off_and_no_read_sees_an_earlier_sum():
    pattern = [True, True, True, True, False, True, True, False, False, True, True, True, True]
    runs = []
    for prefetch in (True, False):
        wl = helpers.random_workload(30, 3000, 4, 4, seed=77)
        tl, raw = make(wl, prefetch, RESCALE_NONE)                     # (no rescaling retries: the count below is exact)
        out = btl_chain(tl, wl, len(pattern), 5, lambda i: pattern[i])
        counts = raw.sitePrefetchCount()
        # a second and a third read of the same sum
        s1 = tl.getSiteLogLikelihoods().copy()
        s2 = tl.getSiteLogLikelihoods().copy()
        assert np.array_equal(s1, s2) and np.array_equal(s1, out[-1][1])
        runs.append((out, counts))
        tl.close()
    (a, na), (b, nb) = runs
    # reads 0, 1 switch it on; sums 2, 3 are served; sum 4 is prefetched and never read (the read of sum 5 must not see it: streak 0 again
    # after sum 5 arrives unread-before); 5, 6 switch it on again, 7 and 8 unread, 9, 10 on, 11, 12 served
    assert nb == 0 and na == 4, (na, nb)
    for (va, sa), (vb, sb) in zip(a, b):
        assert va == vb
        assert (sa is None) == (sb is None)
        if sa is not None:
            assert np.array_equal(sa, sb)
    # and the sums differ from one another, so a stale vector would have shown
    reads = [s for _, s in a if s is not None]
    assert all(not np.array_equal(reads[k], reads[k + 1]) for k in range(len(reads) - 1))


def test_pinned_getter_and_the_sharded_handle(oracle_lib):
    import ctypes as C_
    wl = helpers.random_workload(20, 2500, 4, 4, seed=3)
    tl, raw = make(wl, True, RESCALE_NONE)
    out = btl_chain(tl, wl, 5, 2, lambda i: True)
    fn = raw._ext("beagleMi355GetSiteLogLikelihoodsPinned", [C_.c_int, C_.POINTER(C_.POINTER(C_.c_double)), C_.POINTER(C_.c_long)])
    ptr, n = C_.POINTER(C_.c_double)(), C_.c_long(0)
    before = raw.sitePrefetchCount()
    assert fn(raw.instance, C_.byref(ptr), C_.byref(n)) == 0 and n.value == wl.pattern_count
    assert raw.sitePrefetchCount() == before + 1                       # (the last sum's values were prefetched: handed over in place)
    assert np.array_equal(np.ctypeslib.as_array(ptr, shape=(n.value,)), out[-1][1])
    tl.close()
    # every GPU of the box as ONE instance (resource G + 1; three shards on this one): each shard serves its own block of the patterns
    g = len(bm.beagle.engine().resource_list()) - 2                    # [CPU placeholder, GPU 1..G, all-GPUs]
    old = os.environ.get("BEAGLE_MI355_SHARDS")
    os.environ["BEAGLE_MI355_SHARDS"] = "3"
    try:
        wl = helpers.random_workload(20, 2500, 4, 4, seed=3)
        sh, rs = make(wl, True, RESCALE_NONE, resource_list=(g + 1,))
        o2 = btl_chain(sh, wl, 5, 2, lambda i: True)
        assert rs.sitePrefetchCount() == 3                             # (shard 0's count: sums 2, 3, 4)
        for (va, sa), (vb, sb) in zip(out, o2):
            assert abs(va - vb) <= 1e-12 * abs(va)
            assert np.array_equal(sa, sb)
        sh.close()
    finally:
        if old is None:
            os.environ.pop("BEAGLE_MI355_SHARDS", None)
        else:
            os.environ["BEAGLE_MI355_SHARDS"] = old
