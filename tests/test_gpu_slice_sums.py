"""accumulateScaleFactors from the per-slice products of factors a write-mode walk leaves behind (round 6).

BEAST asks for the sum of the logarithms of ALL internal nodes' factors right behind an updatePartials that has just written them
(BeagleTreeLikelihood.java:1013-1026: ALWAYS; DYNAMIC's every-100th and underflow evaluations).  Reading a thousand per-node buffers
back for that cost config A 250 us per evaluation.  The 4-state walk now keeps, per slice and pattern, the product of the factors the
slice writes — mantissa and binary exponent, renormalised at every node so that it cannot leave the range — and a call over exactly
the buffers the last walk wrote adds those (kernels.hip k_accumulateSlices).  Any other call takes the general kernel.

What must hold: the cumulative buffer and the log-likelihood agree with the general kernel to rounding and with the oracle to 1e-10;
both walk kernels (assembly loop, C++) leave products (of their own slices: to rounding); a call the shortcut does not cover — a branch move's operation list writes
the factors of its path only, BEAST still lists every node — is answered by the general kernel, correctly; the counters say which ran."""
import os

import numpy as np
import pytest

import beast_mcmc_amd as bm
import helpers
from beast_mcmc_amd.treelikelihood import BeagleTreeLikelihood, RESCALE_ALWAYS, RESCALE_DYNAMIC

pytestmark = pytest.mark.gpu


def chain(wl, scheme, env, moves=8):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        tl = BeagleTreeLikelihood(wl, rescaling=scheme, delay_rescaling=False)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    raw = bm.beagle.Beagle.attach(tl)
    vals = [tl.getLogLikelihood()]
    cum = [raw.getLogScaleFactors(tl.cumulative_scale_index()).copy()]
    info0 = raw.walkLaunchInfo()
    tl.makeDirty()
    vals.append(tl.getLogLikelihood())
    r = np.random.default_rng(31)
    for _ in range(moves):
        node = int(r.integers(wl.tree.tip_count, wl.tree.node_count - 1))
        tl.storeState()
        tl.set_node_height(node, helpers.proposed_height(wl.tree, node, r))
        vals.append(tl.getLogLikelihood())
        if r.random() < 0.4:
            tl.restoreState()
            vals.append(tl.getLogLikelihood())
    tl.makeDirty()
    vals.append(tl.getLogLikelihood())
    cum.append(raw.getLogScaleFactors(tl.cumulative_scale_index()).copy())
    site = tl.getSiteLogLikelihoods().copy()
    info = raw.walkLaunchInfo()
    tl.close()
    return vals, cum, site, info0, info


@pytest.mark.parametrize("T,P,C,kind", [(300, 5000, 4, "coalescent"), (64, 129, 1, "yule"), (90, 700, 8, "coalescent"), (40, 1, 4, "coalescent"), (150, 640, 4, "caterpillar")])
def test_always_rescaling_adds_slice_products(T, P, C, kind, oracle_lib):
    wl = helpers.random_workload(T, P, 4, C, seed=9100 + T, tree_kind=kind)
    sv, sc, ss, sinfo0, sinfo = chain(wl, RESCALE_ALWAYS, {"BEAGLE_MI355_NO_SLICE_SUMS": "0"})
    gv, gc, gs, _, ginfo = chain(wl, RESCALE_ALWAYS, {"BEAGLE_MI355_NO_SLICE_SUMS": "1"})
    cv, cc, cs, _, cinfo = chain(wl, RESCALE_ALWAYS, {"BEAGLE_MI355_NO_FAST_WALK": "1"})
    assert sinfo0["slice_accumulations"] == 1                      # the first evaluation already
    assert sinfo["slice_accumulations"] >= 3                        # every full evaluation; the branch moves' calls are not covered
    assert ginfo["slice_accumulations"] == 0 and cinfo["slice_accumulations"] >= 3
    # the C++ kernel leaves products too — of OTHER slices (it runs one launch per wave of shorter slices): to rounding, like the general kernel
    for a, b in zip(sc, cc):
        assert np.max(np.abs(a - b)) <= 1e-12 * max(1.0, float(np.max(np.abs(b))))
    for a, b in zip(sv, cv):
        assert helpers.rel_err(a, b) <= 1e-13
    # against the general kernel: to rounding (a sum of ~T logarithms of order 1-10 each, formed in another order)
    for a, b in zip(sc, gc):
        assert np.max(np.abs(a - b)) <= 1e-12 * max(1.0, float(np.max(np.abs(b))))
    for a, b in zip(sv, gv):
        assert helpers.rel_err(a, b) <= 1e-13
    o = BeagleTreeLikelihood(wl, library=oracle_lib, rescaling=RESCALE_ALWAYS, delay_rescaling=False)
    ref = o.getLogLikelihood()
    so = o.getSiteLogLikelihoods()
    oc = bm.beagle.Beagle.attach(o).getLogScaleFactors(o.cumulative_scale_index())
    o.close()
    assert helpers.rel_err(sv[0], ref) <= 1e-10
    assert np.max(np.abs(sc[0] - oc)) <= 1e-9


def test_dynamic_rescaling_evaluations_use_them_and_partial_lists_do_not(oracle_lib):
    wl = helpers.random_workload(200, 2000, 4, 4, seed=9191, tree_kind="coalescent")
    sv, _, ss, sinfo0, sinfo = chain(wl, RESCALE_DYNAMIC, {"BEAGLE_MI355_NO_SLICE_SUMS": "0"}, moves=12)
    gv, _, gs, _, _ = chain(wl, RESCALE_DYNAMIC, {"BEAGLE_MI355_NO_SLICE_SUMS": "1"}, moves=12)
    assert sinfo0["slice_accumulations"] == 1                      # the first evaluation computes the factors; later ones read them
    for a, b in zip(sv, gv):
        assert helpers.rel_err(a, b) <= 1e-13
    assert np.max(np.abs(ss - gs) / np.abs(gs)) <= 1e-12
    o = BeagleTreeLikelihood(wl, library=oracle_lib, rescaling=RESCALE_DYNAMIC, delay_rescaling=False)
    ref = o.getLogLikelihood()
    o.close()
    assert helpers.rel_err(sv[0], ref) <= 1e-10


def test_a_list_that_names_other_buffers_takes_the_general_kernel(oracle_lib):
    """Raw calls: the walk writes the factors of every internal node; accumulate (i) all of them — covered —, (ii) all but one, (iii) all of
    them after one was overwritten by copyScaleFactors, (iv) one of them twice in a list of the right length: (ii)-(iv) must not take the
    shortcut, and every result must be what the general kernel gives."""
    wl = helpers.random_workload(60, 900, 4, 4, seed=9292, tree_kind="coalescent")
    res = {}
    for mode in ("0", "1"):
        os.environ["BEAGLE_MI355_NO_SLICE_SUMS"] = mode
        try:
            tl = BeagleTreeLikelihood(wl, rescaling=RESCALE_ALWAYS, delay_rescaling=False)
        finally:
            del os.environ["BEAGLE_MI355_NO_SLICE_SUMS"]
        raw = bm.beagle.Beagle.attach(tl)
        tl.getLogLikelihood()
        nodes = list(range(wl.tree.tip_count, wl.tree.node_count))
        idx = [tl.node_scale_index(n) for n in nodes]
        cum = tl.cumulative_scale_index()
        out = []
        n0 = raw.walkLaunchInfo()["slice_accumulations"]
        raw.resetScaleFactors(cum); raw.accumulateScaleFactors(idx, len(idx), cum); out.append(raw.getLogScaleFactors(cum).copy())
        n1 = raw.walkLaunchInfo()["slice_accumulations"]
        raw.resetScaleFactors(cum); raw.accumulateScaleFactors(idx[:-1], len(idx) - 1, cum); out.append(raw.getLogScaleFactors(cum).copy())
        dup = idx[:-1] + [idx[0]]
        raw.resetScaleFactors(cum); raw.accumulateScaleFactors(dup, len(dup), cum); out.append(raw.getLogScaleFactors(cum).copy())
        n2 = raw.walkLaunchInfo()["slice_accumulations"]
        raw.copyScaleFactors(idx[3], idx[5])
        raw.resetScaleFactors(cum); raw.accumulateScaleFactors(idx, len(idx), cum); out.append(raw.getLogScaleFactors(cum).copy())
        n3 = raw.walkLaunchInfo()["slice_accumulations"]
        res[mode] = (out, (n1 - n0, n2 - n1, n3 - n2))
        tl.close()
    assert res["0"][1] == (1, 0, 0) and res["1"][1] == (0, 0, 0)
    for a, b in zip(res["0"][0], res["1"][0]):
        assert np.max(np.abs(a - b)) <= 1e-12 * max(1.0, float(np.max(np.abs(b))))
