"""The root integration inside the walk's launch (4 states, unpartitioned instances).

A one-launch walk is held back until the next call; when that call is calculateRootLogLikelihoods on the walk's last result, the
slice that computes it finishes the evaluation from its registers (kernels_walk4.hip, root_site4.h) — no root kernel, no read-back.
It has to give the same BITS as the launch of its own (kernels.hip k_rootSite4W; BEAGLE_MI355_NO_ROOT_FUSION=1), whatever comes
between the two calls, and everything that is not such a call has to find the walk launched."""
import os

import numpy as np
import pytest

import beast_mcmc_amd as bm
import helpers
from beast_mcmc_amd.treelikelihood import BeagleTreeLikelihood, RESCALE_ALWAYS, RESCALE_DYNAMIC, RESCALE_NONE

pytestmark = pytest.mark.gpu
REL_TOL = 1e-10


def make(wl, fused, scheme):
    old = os.environ.get("BEAGLE_MI355_NO_ROOT_FUSION")
    os.environ["BEAGLE_MI355_NO_ROOT_FUSION"] = "0" if fused else "1"       # read when the instance is created
    try:
        tl = BeagleTreeLikelihood(wl, rescaling=scheme, delay_rescaling=False)
    finally:
        if old is None:
            del os.environ["BEAGLE_MI355_NO_ROOT_FUSION"]
        else:
            os.environ["BEAGLE_MI355_NO_ROOT_FUSION"] = old
    return tl, bm.beagle.Beagle.attach(tl)


def chain(tl, wl, rng, steps):
    """full evaluations, branch moves and their rejections; -> every value and the last site vector"""
    t_, n_ = wl.tree.tip_count, wl.tree.node_count
    height = np.array(wl.tree.height, dtype=float)
    out = [tl.getLogLikelihood()]
    for it in range(steps):
        tl.storeState()
        node = -1
        if it % 3 == 0 or n_ - t_ < 2:                        # (two taxa: the only internal node is the root — nothing to move)
            tl.makeDirty()
        else:
            node = int(rng.integers(t_, n_))
            while wl.tree.parent[node] < 0:
                node = int(rng.integers(t_, n_))
            lo = max(height[int(wl.tree.left[node])], height[int(wl.tree.right[node])])
            hi = height[wl.tree.parent[node]]
            old = float(height[node])
            height[node] = lo + (hi - lo) * float(rng.uniform(0.1, 0.9))
            tl.set_node_height(node, float(height[node]))
        out.append(tl.getLogLikelihood())
        if it % 4 == 1:
            tl.restoreState()
            if node >= 0:
                tl.restore_node_height(node, old)              # (the tree model's own restore)
                height[node] = old
            out.append(tl.getLogLikelihood())
    assert all(np.isfinite(v) for v in out), out
    return out, tl.getSiteLogLikelihoods().copy()


@pytest.mark.parametrize("C,T,P", [(4, 40, 1000), (1, 9, 127), (3, 25, 129), (8, 20, 700), (16, 12, 300), (4, 2, 1), (5, 90, 2049)])
@pytest.mark.parametrize("scheme", [RESCALE_NONE, RESCALE_DYNAMIC, RESCALE_ALWAYS])
def test_root_inside_the_walk_equals_the_root_kernel_bit_for_bit(C, T, P, scheme, oracle_lib):
    wl = helpers.random_workload(T, P, 4, C, seed=300 + C + T)
    a, ra = make(wl, True, scheme)
    b, rb = make(wl, False, scheme)
    va, sa = chain(a, wl, np.random.default_rng(5), 9)
    wl2 = helpers.random_workload(T, P, 4, C, seed=300 + C + T)          # (the chain moved wl's heights: the same start for b)
    b.close()
    b, rb = make(wl2, False, scheme)
    vb, sb = chain(b, wl2, np.random.default_rng(5), 9)
    assert va == vb, (va, vb)
    assert np.array_equal(sa, sb)
    # (ALWAYS accumulates the new scale factors between updatePartials and the root call: those launches find the walk launched,
    # nothing is left to hold; DYNAMIC: the same on the evaluations that recompute the factors only)
    assert rb.rootFusedCount() == 0 and (ra.rootFusedCount() > 0 or scheme == RESCALE_ALWAYS), (ra.rootFusedCount(), rb.rootFusedCount())
    wl3 = helpers.random_workload(T, P, 4, C, seed=300 + C + T)
    o = BeagleTreeLikelihood(wl3, library=oracle_lib, rescaling=scheme, delay_rescaling=False)    # the same chain on the oracle
    vo, so = chain(o, wl3, np.random.default_rng(5), 9)
    assert len(vo) == len(va) and max(helpers.rel_err(x, y) for x, y in zip(va, vo)) <= REL_TOL
    assert np.max(np.abs(sa - so) / np.maximum(np.abs(so), 1e-300)) <= REL_TOL
    a.close(); b.close(); o.close()


def raw_instance(wl, library=None):
    """the bare call sequence on one set of buffers: -> (instance, operation list, root buffer, evaluate())"""
    B = bm.beagle
    tree, T, n = wl.tree, wl.tree.tip_count, wl.tree.node_count
    b = B.Beagle(T, n, T, wl.state_count, wl.pattern_count, 1, n, wl.category_count, 1, library=library)
    for t in range(T):
        b.setTipStates(t, wl.tip_states[t])
    b.setPatternWeights(wl.weights)
    b.setEigenDecomposition(0, wl.eig.evec, wl.eig.ievc, wl.eig.evals)
    b.setCategoryRates(wl.cat_rates)
    b.setCategoryWeights(0, wl.cat_weights)
    b.setStateFrequencies(0, wl.freqs)
    branches = [x for x in range(n) if x != tree.root]
    b.updateTransitionMatrices(0, branches, None, None, [tree.branch_length(x) for x in branches], len(branches))
    ops = []
    for x in tree.postorder():
        if x >= T:
            l, r = int(tree.left[x]), int(tree.right[x])
            ops += [x, B.NONE, B.NONE, l, l, r, r]

    def root(buf=tree.root):
        out = [0.0]
        b.calculateRootLogLikelihoods([buf], [0], [0], [B.NONE], 1, out)
        return out[0]
    return b, ops, root


def test_calls_between_update_partials_and_the_root_find_the_walk_done(oracle_lib):
    """Raw call sequences: whatever is asked for while a launch is held back sees the result of updatePartials — a partials read-back,
    a changed tip (must NOT reach the held walk), a root call on ANOTHER buffer; the reference's own order (category weights and
    frequencies set between the two calls) keeps the launch held."""
    B = bm.beagle
    wl = helpers.random_workload(12, 300, 4, 4, seed=77)
    T, n = wl.tree.tip_count, wl.tree.node_count
    g, ops, groot = raw_instance(wl)
    o, _, oroot = raw_instance(wl, library=oracle_lib)
    count = len(ops) // 7
    o.updatePartials(ops, count, B.NONE)
    ref = oroot()
    # the plain sequence: fused
    g.updatePartials(ops, count, B.NONE)
    assert helpers.rel_err(groot(), ref) <= REL_TOL
    base = g.rootFusedCount()
    assert base == 1
    # (1) a read-back right behind updatePartials
    g.updatePartials(ops, count, B.NONE)
    for x in range(T, n):
        pa, po = g.getPartials(x, B.NONE), o.getPartials(x, B.NONE)
        assert np.max(np.abs(pa - po)) <= REL_TOL * max(1.0, float(np.max(np.abs(po)))), x
    assert helpers.rel_err(groot(), ref) <= REL_TOL
    assert g.rootFusedCount() == base                             # the read-back had launched the walk: nothing left to fuse
    # (2) a tip changes between updatePartials and the root call: the held walk is launched first, with the OLD tip
    g.updatePartials(ops, count, B.NONE)
    states = np.array(wl.tip_states[0])
    g.setTipStates(0, (states + 1) % 4)
    assert helpers.rel_err(groot(), ref) <= REL_TOL
    g.setTipStates(0, states)
    # (3) the reference's own order keeps the launch held
    g.updatePartials(ops, count, B.NONE)
    fused = g.rootFusedCount()
    g.setCategoryWeights(0, wl.cat_weights[::-1].copy())
    g.setStateFrequencies(0, wl.freqs[::-1].copy())
    o.setCategoryWeights(0, wl.cat_weights[::-1].copy())
    o.setStateFrequencies(0, wl.freqs[::-1].copy())
    v, w = groot(), oroot()
    assert abs(w - ref) > 1e-6 * abs(ref)                         # (the new weights and frequencies do matter)
    assert helpers.rel_err(v, w) <= REL_TOL
    assert g.rootFusedCount() == fused + 1
    # (4) the root log-likelihood of ANOTHER buffer of the list
    other = int(wl.tree.left[wl.tree.root]) if int(wl.tree.left[wl.tree.root]) >= T else int(wl.tree.right[wl.tree.root])
    g.updatePartials(ops, count, B.NONE)
    assert helpers.rel_err(groot(other), oroot(other)) <= REL_TOL
    assert helpers.rel_err(groot(), oroot()) <= REL_TOL
    g.finalize(); o.finalize()
