"""The one-launch walk cannot deadlock, whatever order the hardware dispatches workgroups in.

All slices of a 4-state program run in ONE launch; a workgroup polls the flags of the slices it reads from
(kernels_walk4.hip k_walk4_fast).  On gfx950 those slices' workgroups are always dispatched first, but that is observed
behaviour, not a documented guarantee — so the wait is bounded (20 ms) and a workgroup whose wait runs out computes the slices in
front of its own itself (same 128 patterns; a result computed twice is stored twice with the same bits).
BEAGLE_MI355_WALK_SPIN_US=0 (read at instance creation) makes every wait run out at once: every workgroup with a dependency
serves itself.  The results must be the same BITS as the default's, over full evaluations in write and read mode, branch moves
and rejections, and the counter of self-served workgroups must say that the path really ran (and that by default it does not)."""
import os

import numpy as np
import pytest

import beast_mcmc_amd as bm
import helpers
from beast_mcmc_amd.treelikelihood import BeagleTreeLikelihood, RESCALE_ALWAYS, RESCALE_DYNAMIC, RESCALE_NONE

pytestmark = pytest.mark.gpu


def chain(wl, spin_us, scheme, moves=12):
    old = os.environ.get("BEAGLE_MI355_WALK_SPIN_US")
    if spin_us is None:
        os.environ.pop("BEAGLE_MI355_WALK_SPIN_US", None)
    else:
        os.environ["BEAGLE_MI355_WALK_SPIN_US"] = str(spin_us)
    try:
        tl = BeagleTreeLikelihood(wl, rescaling=scheme, delay_rescaling=False)
    finally:
        if old is None:
            os.environ.pop("BEAGLE_MI355_WALK_SPIN_US", None)
        else:
            os.environ["BEAGLE_MI355_WALK_SPIN_US"] = old
    raw = bm.beagle.Beagle.attach(tl)
    vals = [tl.getLogLikelihood()]
    tl.makeDirty()
    vals.append(tl.getLogLikelihood())
    r = np.random.default_rng(5)
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
    site = tl.getSiteLogLikelihoods().copy()
    nodes = list(range(wl.tree.tip_count, wl.tree.node_count))[::7]
    parts = [raw.getPartials(tl.node_buffer_index(n), bm.beagle.NONE).copy() for n in nodes]
    health = raw.walkHealth()
    stats = raw.rootFusedCount()
    tl.close()
    return vals, site, parts, health, stats


@pytest.mark.parametrize("scheme", [RESCALE_NONE, RESCALE_DYNAMIC, RESCALE_ALWAYS])
@pytest.mark.parametrize("T,P,C,kind", [(220, 3000, 4, "coalescent"), (150, 1100, 2, "yule"), (120, 700, 8, "caterpillar")])
def test_workgroups_that_stop_waiting_compute_the_same_bits(T, P, C, kind, scheme, oracle_lib):
    wl = helpers.random_workload(T, P, 4, C, seed=7100 + T, tree_kind=kind)
    dv, ds, dp, dh, dfused = chain(wl, None, scheme)
    fv, fs, fp, fh, ffused = chain(wl, 0, scheme)
    # gfx950 dispatches in order: nobody's wait runs out (a stray one — a 20 ms hiccup of the box under a workgroup — would only cost time)
    assert dh["spin_limit_us"] == 20000 and dh["self_served"] <= 2
    assert fh["spin_limit_us"] == 0                                      # ... and here everybody's did
    if kind != "caterpillar":                                            # (a ladder has no side subtrees to cut off: one slice, nobody waits)
        assert fh["self_served"] > 0
    assert dv == fv
    assert np.array_equal(ds, fs)
    for a, b in zip(dp, fp):
        assert np.array_equal(a, b)
    assert dfused == ffused                                              # the root slice still finished the evaluation in both
    if scheme != RESCALE_ALWAYS:                                         # (evaluations that accumulate new factors are not held back)
        assert dfused > 0
    o = BeagleTreeLikelihood(wl, library=oracle_lib, rescaling=scheme, delay_rescaling=False)
    ref = o.getLogLikelihood()
    o.close()
    assert helpers.rel_err(dv[0], ref) <= 1e-10


def test_a_short_limit_only_costs_time():
    """A limit of a few microseconds lets some waits run out and others not — a mix of owners and helpers computing the same
    slices at the same time: same bits."""
    wl = helpers.random_workload(300, 5000, 4, 4, seed=7301, tree_kind="coalescent")
    dv, ds, _, _, _ = chain(wl, None, RESCALE_DYNAMIC, moves=4)
    for us in (2, 10, 50):
        fv, fs, _, fh, _ = chain(wl, us, RESCALE_DYNAMIC, moves=4)
        assert dv == fv and np.array_equal(ds, fs), us
