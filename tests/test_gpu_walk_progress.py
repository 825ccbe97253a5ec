"""The one-launch walk cannot deadlock, whatever order the hardware dispatches workgroups in.

All slices of a 4-state program run in ONE launch; a workgroup polls the flags of the slices it reads from
(kernels_walk4.hip k_walk4_fast).  On gfx950 those slices' workgroups are always dispatched first, but that is observed
behaviour, not a documented guarantee — so the wait is bounded (20 ms) and a workgroup whose wait runs out computes the slices in
front of its own itself (same 128 patterns; a result computed twice is stored twice with the same bits).
BEAGLE_MI355_WALK_SPIN_US=0 (read at instance creation) makes every wait run out at once: every workgroup with a dependency
serves itself.  The results must be the same BITS as the default's, over full evaluations in write and read mode, branch moves
and rejections, and the counter of self-served workgroups must say that the path really ran (and that by default it does not).

Round 6: when a program's slices form a forest (a stored slice root has one reader — every list BEAST issues) the launch runs on
TICKETS instead: only the slices without dependencies get workgroups, a workgroup that finishes a slice counts itself in at the
slice above and the one that arrives last there runs it.  Nobody waits, so there is nothing to bound.  The flag form stays for
programs that are no forest and behind BEAGLE_MI355_NO_WALK_TICKETS=1 — which is how the tests below reach it — and the default
(tickets) must give the same BITS as both flag runs."""
import os

import numpy as np
import pytest

import beast_mcmc_amd as bm
import helpers
from beast_mcmc_amd.treelikelihood import BeagleTreeLikelihood, RESCALE_ALWAYS, RESCALE_DYNAMIC, RESCALE_NONE

pytestmark = pytest.mark.gpu


def chain(wl, spin_us, scheme, moves=12, tickets=False):
    # (BEAGLE_MI355_NO_SLICE_SUMS=1: the cumulative buffer from the per-node factors, which are the same bits on both launch forms; the per-slice
    # products depend on the slice sizes, which the two forms choose differently — tests/test_gpu_slice_sums.py holds those to rounding)
    env = {"BEAGLE_MI355_WALK_SPIN_US": None if spin_us is None else str(spin_us), "BEAGLE_MI355_NO_WALK_TICKETS": None if tickets else "1",
           "BEAGLE_MI355_NO_SLICE_SUMS": "1"}
    old = {k: os.environ.get(k) for k in env}
    for k, v in env.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
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
    health.update(raw.walkLaunchInfo())
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
    assert dh["ticket_walks"] == 0 and fh["ticket_walks"] == 0 and dh["flag_walks"] > 0
    # ... and the default form of the launch, on tickets: the same bits, the same evaluations finished by the root slice
    tv, ts, tp, th, tfused = chain(wl, None, scheme, tickets=True)
    assert th["ticket_walks"] > 0 and th["flag_walks"] == 0 and th["self_served"] == 0
    if kind != "caterpillar":
        assert th["rows"] < th["slices"] or th["slices"] == 1            # (the last launch: slices above the first wave have no workgroups of their own)
    assert tv == dv and np.array_equal(ts, ds) and tfused == dfused
    for a, b in zip(tp, dp):
        assert np.array_equal(a, b)
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


@pytest.mark.parametrize("P", [1, 127, 129, 640, 12500])
def test_tickets_over_ragged_pattern_counts_and_many_slices(P, oracle_lib):
    """The ticket form over pattern counts that leave the last group ragged (or alone), on a tree deep enough for several levels of
    slices: flags and tickets give the same bits, the oracle agrees, and the arrival counters are back at zero behind every launch
    (a second chain on the same instance would otherwise start slices early or never)."""
    wl = helpers.random_workload(400, P, 4, 4, seed=7400 + P, tree_kind="coalescent")
    fv, fs, fp, fh, _ = chain(wl, None, RESCALE_DYNAMIC, moves=8)
    tv, ts, tp, th, _ = chain(wl, None, RESCALE_DYNAMIC, moves=8, tickets=True)
    assert th["ticket_walks"] > 0 and fh["ticket_walks"] == 0
    assert tv == fv and np.array_equal(ts, fs)
    for a, b in zip(tp, fp):
        assert np.array_equal(a, b)
    o = BeagleTreeLikelihood(wl, library=oracle_lib, rescaling=RESCALE_DYNAMIC, delay_rescaling=False)
    ref = o.getLogLikelihood()
    o.close()
    assert helpers.rel_err(tv[0], ref) <= 1e-10
