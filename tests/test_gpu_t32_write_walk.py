"""Write-mode rescaling on the T32 walk (kernels_mfma.hip k_walkT32W1; 17..20 states, up to four rate categories).

PartialsRescalingScheme ALWAYS and DYNAMIC's rescaling evaluations (BeagleTreeLikelihood.java:1013-1026, 1059-1113) hand the
engine lists whose operations WRITE scale factors: a pattern's factor is the maximum over its states and rate categories
(GeneralLikelihoodCore.java:281-318).  Until round 6 such lists left the walk and ran level by level; now a workgroup holds all
categories of its tiles and the list stays on the walk.  Held here: the walk was taken; the log-likelihood, the site values, every
node's factors and the cumulative buffer follow the CPU oracle (1e-10) and the level kernels (BEAGLE_MI355_NO_T32_WRITE_WALK=1:
another order of additions, 1e-12; factors are maxima of those sums); ragged pattern counts (half-filled tiles, an odd tile count —
the workgroup's second tile missing —, a single pattern); more categories than the kernel takes fall back to the levels.
"""
import os

import numpy as np
import pytest

import beast_mcmc_amd as bm
import helpers
from beast_mcmc_amd.treelikelihood import BeagleTreeLikelihood, RESCALE_ALWAYS, RESCALE_DYNAMIC

pytestmark = pytest.mark.gpu


def evaluate(wl, scheme, env=None, library=None, evaluations=2):
    env = env or {}
    os.environ.update(env)
    try:
        tl = BeagleTreeLikelihood(wl, library=library, rescaling=scheme, delay_rescaling=False)
        raw = bm.beagle.Beagle.attach(tl)
        if library is None:
            raw.kernelTimer(True)
        lnl = []
        for _ in range(evaluations):                       # (the second one: the flipped buffers, the cached plan)
            lnl.append(tl.getLogLikelihood())
            tl.makeDirty()
        site = tl.getSiteLogLikelihoods()
        T = wl.tip_count
        factors = [raw.getLogScaleFactors(tl.node_scale_index(n)) for n in range(T, wl.tree.node_count)]
        cum = raw.getLogScaleFactors(tl.cumulative_scale_index())
        stats = raw.walkStats() if library is None else None
        tl.close()
    finally:
        for k in env:
            os.environ.pop(k, None)
    return lnl, site, np.array(factors), cum, stats


@pytest.mark.parametrize("S,C,P", [(20, 4, 2000), (20, 4, 1000 + 37), (17, 4, 700), (20, 2, 333), (20, 1, 96), (19, 3, 33), (20, 4, 1)])
def test_always_rescaling_stays_on_the_walk_and_follows_oracle_and_levels(S, C, P, oracle_lib):
    wl = helpers.random_workload(48, P, S, C, seed=1000 + S + C + P)
    lnl, site, fac, cum, st = evaluate(wl, RESCALE_ALWAYS)
    assert st["walks"] > 0 and st["scale_writes"] > 0, st                      # write-mode micro-operations ran on the walk
    assert st["stored"] < st["micro_ops"], st                                   # ... and not every node went to memory
    assert lnl[0] == lnl[1]
    lv_lnl, lv_site, lv_fac, lv_cum, lv_st = evaluate(wl, RESCALE_ALWAYS, {"BEAGLE_MI355_NO_T32_WRITE_WALK": "1"})
    assert lv_st["walks"] == 0, lv_st                                           # (level kernels only)
    assert helpers.rel_err(lnl[0], lv_lnl[0]) <= 1e-12
    assert np.max(np.abs(site - lv_site) / np.abs(lv_site)) <= 1e-12
    assert np.max(np.abs(fac - lv_fac)) <= 1e-11 and np.max(np.abs(cum - lv_cum) / np.maximum(1.0, np.abs(lv_cum))) <= 1e-12
    o_lnl, o_site, o_fac, o_cum, _ = evaluate(wl, RESCALE_ALWAYS, library=oracle_lib, evaluations=1)
    assert np.isfinite(o_lnl[0]) and helpers.rel_err(lnl[0], o_lnl[0]) <= 1e-10
    assert np.max(np.abs(site - o_site) / np.abs(o_site)) <= 1e-10
    assert np.max(np.abs(fac - o_fac)) <= 1e-9 and np.max(np.abs(cum - o_cum) / np.maximum(1.0, np.abs(o_cum))) <= 1e-10


def test_a_thousand_taxa(oracle_lib):
    """1 000 taxa (999 write-mode micro-operations in slices of two waves): value, site values and cumulative factors follow the oracle."""
    wl = helpers.random_workload(1000, 128, 20, 4, seed=5, root_to_tip=6.0)
    lnl, site, _, cum, st = evaluate(wl, RESCALE_ALWAYS, evaluations=1)
    o_lnl, o_site, _, o_cum, _ = evaluate(wl, RESCALE_ALWAYS, library=oracle_lib, evaluations=1)
    assert st["scale_writes"] == 999 and np.isfinite(lnl[0]) and helpers.rel_err(lnl[0], o_lnl[0]) <= 1e-10
    assert np.max(np.abs(site - o_site) / np.abs(o_site)) <= 1e-10
    assert np.min(cum) < -100.0 and np.max(np.abs(cum - o_cum) / np.maximum(1.0, np.abs(o_cum))) <= 1e-10


def test_dynamic_scheme_rescaling_evaluations_take_the_walk():
    wl = helpers.random_workload(40, 700, 20, 4, seed=9)
    lnl, _, _, _, st = evaluate(wl, RESCALE_DYNAMIC, evaluations=3)
    lv, _, _, _, _ = evaluate(wl, RESCALE_DYNAMIC, {"BEAGLE_MI355_NO_T32_WRITE_WALK": "1"}, evaluations=3)
    assert st["scale_writes"] > 0 and st["walks"] > 0
    for a, b in zip(lnl, lv):
        assert helpers.rel_err(a, b) <= 1e-12


def test_more_categories_than_the_kernel_takes_run_level_by_level(oracle_lib):
    wl = helpers.random_workload(24, 200, 20, 6, seed=12)
    lnl, _, _, _, st = evaluate(wl, RESCALE_ALWAYS, evaluations=1)
    o_lnl, _, _, _, _ = evaluate(wl, RESCALE_ALWAYS, library=oracle_lib, evaluations=1)
    assert st["walks"] == 0 and helpers.rel_err(lnl[0], o_lnl[0]) <= 1e-10


@pytest.mark.parametrize("S", [20, 17])
def test_chain_of_moves_under_always_rescaling(S, oracle_lib):
    """Scheme ALWAYS through what a chain does: node-height moves (partial lists that write the factors of one path and read their
    siblings from memory), rejections (restoreState: index flips only), a model change and rate changes in between — every value against
    the same chain on the level kernels (1e-12) and on the oracle (1e-10)."""
    from beast_mcmc_amd.inputs import substmodel
    wl = helpers.random_workload(40, 700, S, 4, seed=31)
    rng = np.random.default_rng(4)
    moves = []
    for step in range(12):
        kind = ("height", "height", "rates", "model")[step % 4]
        moves.append((kind, substmodel.random_reversible(S, rng)[0], rng.uniform(0.7, 1.4, size=wl.tree.node_count),
                      int(rng.integers(wl.tip_count, wl.tree.node_count)), step % 3 == 2))

    def chain(tl):
        out = [tl.getLogLikelihood()]
        height = wl.tree.height.copy()
        for kind, eig, rates, node, reject in moves:
            tl.storeState()
            saved = height.copy()
            if kind == "model":
                tl.set_substitution_model(eig, wl.freqs)
            elif kind == "rates":
                tl.set_branch_rates(rates)
            elif node != wl.tree.root:
                lo = max(height[wl.tree.left[node]], height[wl.tree.right[node]])
                hi = height[wl.tree.parent[node]]
                height[node] = lo + 0.41 * (hi - lo)
                tl.set_node_height(node, float(height[node]))
            else:
                tl.makeDirty()
            out.append(tl.getLogLikelihood())
            if reject:
                tl.restoreState()
                height = saved
                out.append(tl.getLogLikelihood())
        return out

    runs = {}
    for name, env, lib in (("walk", {}, None), ("levels", {"BEAGLE_MI355_NO_T32_WRITE_WALK": "1"}, None), ("oracle", {}, oracle_lib)):
        os.environ.update(env)
        try:
            tl = BeagleTreeLikelihood(wl, library=lib, rescaling=RESCALE_ALWAYS, delay_rescaling=False)
            if name == "walk":
                raw = bm.beagle.Beagle.attach(tl)
                raw.kernelTimer(True)
            runs[name] = chain(tl)
            if name == "walk":
                st = raw.walkStats()
                assert st["walks"] > 0 and st["scale_writes"] > 0
            tl.close()
        finally:
            for k in env:
                os.environ.pop(k, None)
    assert len(runs["walk"]) == len(runs["levels"]) == len(runs["oracle"])
    for a, b, c in zip(runs["walk"], runs["levels"], runs["oracle"]):
        assert helpers.rel_err(a, b) <= 1e-12 and helpers.rel_err(a, c) <= 1e-10
