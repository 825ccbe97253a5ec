"""The two implementations of the 4-state pattern walk against each other and against the oracle.

k_walk4_fast (the generated assembly loop, tools/gen_walk4_fast.py) runs every launch of a default instance — read mode,
write-mode rescaling (LDS exchange across the category waves, two barriers, a true division) and no scaling alike; k_walk4
(C++) is the reference implementation (BEAGLE_MI355_NO_FAST_WALK=1).  They must produce the same bits: same mat-vec order,
same products, same factors and reciprocals — the difference is the instruction stream, the lane <-> pattern assignment and
the shape of the stores.  Checked here for every rate-category bracket (kernel instantiations for <= 4,
<= 8, <= 16 categories), ragged pattern counts (the masked last workgroup), all tree shapes (hold slots, partials
re-read from memory, both children in memory) and the read-mode steady state of DYNAMIC rescaling."""
import os

import numpy as np
import pytest

import beast_mcmc_amd as bm
import helpers
from beast_mcmc_amd.treelikelihood import (BeagleTreeLikelihood, RESCALE_ALWAYS, RESCALE_DYNAMIC, RESCALE_NONE, POST_ORDER,
                                           REVERSE_LEVEL_ORDER)

pytestmark = pytest.mark.gpu
REL_TOL = 1e-10


def evaluate(wl, fast, scheme, traversal, nodes):
    """lnL, site lnL and the partials of `nodes` after a write-mode and a read-mode evaluation."""
    old = os.environ.get("BEAGLE_MI355_NO_FAST_WALK")
    os.environ["BEAGLE_MI355_NO_FAST_WALK"] = "0" if fast else "1"      # read when the instance is created
    # (BEAGLE_MI355_NO_SLICE_SUMS=1: the cumulative buffer from the per-node factors — which ARE the same bits on every path compared here;
    # the per-slice products of round 6 depend on how a program is cut into slices, which differs between the kernels / launch forms:
    # tests/test_gpu_slice_sums.py holds those to rounding)
    os.environ["BEAGLE_MI355_NO_SLICE_SUMS"] = "1"
    try:
        tl = BeagleTreeLikelihood(wl, rescaling=scheme, delay_rescaling=False, traversal=traversal)
    finally:
        del os.environ["BEAGLE_MI355_NO_SLICE_SUMS"]
        if old is None:
            del os.environ["BEAGLE_MI355_NO_FAST_WALK"]
        else:
            os.environ["BEAGLE_MI355_NO_FAST_WALK"] = old
    raw = bm.beagle.Beagle.attach(tl)
    raw.kernelTimer(True)
    first = tl.getLogLikelihood()          # DYNAMIC / ALWAYS: rescaling in write mode
    tl.makeDirty()
    lnl = tl.getLogLikelihood()            # DYNAMIC: read mode;  ALWAYS: write mode again;  NONE: no scaling
    stats = raw.walkStats()
    raw.kernelTimer(False)
    site = tl.getSiteLogLikelihoods().copy()
    parts = [raw.getPartials(tl.node_buffer_index(n), bm.beagle.NONE).copy() for n in nodes]
    if scheme != RESCALE_NONE:             # the factors themselves (log of what the rescaling operations stored)
        parts += [raw.getLogScaleFactors(tl.node_scale_index(n)).copy() for n in nodes]
    tl.close()
    return first, lnl, site, parts, stats


@pytest.mark.parametrize("C,T,P", [(4, 60, 1000), (1, 33, 129), (3, 25, 127), (8, 20, 700), (16, 12, 300), (5, 90, 2049)])
@pytest.mark.parametrize("scheme", [RESCALE_NONE, RESCALE_DYNAMIC, RESCALE_ALWAYS])
@pytest.mark.parametrize("kind,traversal", [("coalescent", REVERSE_LEVEL_ORDER), ("yule", POST_ORDER), ("caterpillar", POST_ORDER)])
def test_assembly_loop_equals_cpp_kernel_bit_for_bit(C, T, P, scheme, kind, traversal, oracle_lib):
    wl = helpers.random_workload(T, P, 4, C, seed=900 + C + T, tree_kind=kind)
    nodes = list(range(wl.tree.tip_count, wl.tree.node_count))
    f0, fl, fs, fp, fstats = evaluate(wl, True, scheme, traversal, nodes)
    g0, gl, gs, gp, gstats = evaluate(wl, False, scheme, traversal, nodes)
    assert fstats["fast_walks"] == fstats["walks"] > 0 and gstats["fast_walks"] == 0          # each run really used its kernel
    assert f0 == g0 and fl == gl
    assert np.array_equal(fs, gs)
    for a, b in zip(fp, gp):
        assert np.array_equal(a, b)
    # and both are right
    o = BeagleTreeLikelihood(wl, library=oracle_lib, rescaling=scheme, delay_rescaling=False, traversal=traversal)
    ref = o.getLogLikelihood()
    so = o.getSiteLogLikelihoods()
    assert helpers.rel_err(fl, ref) <= REL_TOL
    assert np.max(np.abs(fs - so) / np.maximum(np.abs(so), 1e-300)) <= REL_TOL
    o.close()


def test_partial_updates_use_the_assembly_loop_and_match(oracle_lib):
    """An MCMC-like sequence of branch moves (short operation lists, buffers re-read from memory) on both kernels."""
    wl = helpers.random_workload(70, 900, 4, 4, seed=4242, tree_kind="coalescent")
    rng = np.random.default_rng(3)
    results = []
    for fast in (True, False):
        os.environ["BEAGLE_MI355_NO_FAST_WALK"] = "0" if fast else "1"
        os.environ["BEAGLE_MI355_NO_SLICE_SUMS"] = "1"           # (see evaluate)
        try:
            tl = BeagleTreeLikelihood(wl, rescaling=RESCALE_DYNAMIC, delay_rescaling=False)
        finally:
            del os.environ["BEAGLE_MI355_NO_FAST_WALK"], os.environ["BEAGLE_MI355_NO_SLICE_SUMS"]
        vals = [tl.getLogLikelihood()]
        r = np.random.default_rng(11)
        for step in range(25):
            node = int(r.integers(wl.tree.tip_count, wl.tree.node_count - 1))
            tl.storeState()
            tl.set_node_height(node, float(wl.tree.height[node]) * (1.0 + 0.01 * r.standard_normal()))
            vals.append(tl.getLogLikelihood())
            if r.random() < 0.4:
                tl.restoreState()
                vals.append(tl.getLogLikelihood())
        results.append(vals)
        tl.close()
    assert results[0] == results[1]


def _chain_with(wl, scheme, env, moves=10):
    """A chain of full evaluations, branch moves and rejections on an instance created under `env`; everything it computes."""
    env = dict(env, BEAGLE_MI355_NO_SLICE_SUMS="1")          # (see evaluate)
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
    raw.kernelTimer(True)
    vals = [tl.getLogLikelihood()]
    tl.makeDirty()
    vals.append(tl.getLogLikelihood())
    r = np.random.default_rng(77)
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
    info = raw.walkLaunchInfo()
    site = tl.getSiteLogLikelihoods().copy()
    nodes = list(range(wl.tree.tip_count, wl.tree.node_count))
    parts = [raw.getPartials(tl.node_buffer_index(n), bm.beagle.NONE).copy() for n in nodes]
    tl.close()
    return vals, site, parts, info


@pytest.mark.parametrize("scheme", [RESCALE_NONE, RESCALE_DYNAMIC, RESCALE_ALWAYS])
@pytest.mark.parametrize("C,T,P,kind", [(4, 200, 3000, "coalescent"), (1, 64, 129, "yule"), (8, 90, 700, "coalescent"), (4, 50, 1, "coalescent"),
                                         (4, 120, 640, "caterpillar"), (16, 30, 300, "yule")])
def test_fused_cherries_give_the_bits_of_the_unfused_program(C, T, P, kind, scheme, oracle_lib):
    """Round 6: a node over two compact tips that is not stored, pays no scale factors and is taken by the very next micro-operation
    is evaluated INSIDE that micro-operation's stage (kernels.h WK_CHERRY; engine_walk.cpp runPlan) — its own stage, a third of a
    tree's, disappears from the device program.  Same arithmetic in the same order: every log-likelihood, site value and node partial
    of a chain must equal, bit for bit, what the unfused program (BEAGLE_MI355_NO_CHERRY_FUSION=1) and the C++ kernel give, and the
    counters must say that cherries really were fused (read mode and no scaling; never in a program that rescales in write mode)."""
    wl = helpers.random_workload(T, P, 4, C, seed=8800 + T + C, tree_kind=kind)
    fv, fs, fp, finfo = _chain_with(wl, scheme, {"BEAGLE_MI355_NO_CHERRY_FUSION": "0"})
    uv, us, up, uinfo = _chain_with(wl, scheme, {"BEAGLE_MI355_NO_CHERRY_FUSION": "1"})
    cv, cs, cp, _ = _chain_with(wl, scheme, {"BEAGLE_MI355_NO_FAST_WALK": "1"})
    assert uinfo["fused_cherries"] == 0
    if scheme == RESCALE_ALWAYS:
        assert finfo["fused_cherries"] == 0                                  # every program rescales in write mode
    elif kind != "caterpillar":
        assert finfo["fused_cherries"] > 0                                   # (full evaluations fuse; a ladder has ONE cherry, branch moves under per-node factors none)
    assert fv == uv == cv
    assert np.array_equal(fs, us) and np.array_equal(fs, cs)
    for a, b, c in zip(fp, up, cp):
        assert np.array_equal(a, b) and np.array_equal(a, c)
    o = BeagleTreeLikelihood(wl, library=oracle_lib, rescaling=scheme, delay_rescaling=False)
    ref = o.getLogLikelihood()
    o.close()
    assert helpers.rel_err(fv[0], ref) <= REL_TOL
