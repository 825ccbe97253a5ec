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
    try:
        tl = BeagleTreeLikelihood(wl, rescaling=scheme, delay_rescaling=False, traversal=traversal)
    finally:
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
        try:
            tl = BeagleTreeLikelihood(wl, rescaling=RESCALE_DYNAMIC, delay_rescaling=False)
        finally:
            del os.environ["BEAGLE_MI355_NO_FAST_WALK"]
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
