"""The pattern walk at 21..64 states (kernels_mfma.hip k_walkT64; codon models: 61 states).

Until round 6 every list at these state counts ran level by level (k_pruneTiled<16>): every node stored and read back.  Now a wave
that owns (tile of 32 patterns, rate category) walks the planner's program — planned WITHOUT hold slots (planner.h: ladders are
definitions, a node over two evaluated children takes one of them through memory) — with its running result in registers, the two
branch matrices of a micro-operation as LDS-DMA'd fragment halves and a child's stored partials streamed row tile by row tile.
Reference arithmetic: GeneralLikelihoodCore.java:52-203 (updateStatesStates / StatesPartials / PartialsPartials), read-mode scaling
BeagleTreeLikelihood.java:1013-1026.

Held here: the walk was taken and left nodes unstored; its numbers equal the level kernels' BIT FOR BIT (same accumulation order per
result element: column tiles ascending in the matrix core) and follow the CPU oracle (1e-10) — log-likelihood, site values, every
internal node's partials (unstored ones materialised on demand); ragged pattern counts (half-filled tiles, fewer tiles than a
workgroup's four, one pattern), partial state tiles (21, 33, 60, 61, 63) and full ones (64), 1..5 rate categories; lists that rescale in
write mode leave the walk (level kernels on operands materialised first) and come back; a chain of node-height moves, rejections,
rate and model changes.  Partitioned instances at 61 states: tests/test_gpu_multipartition.py.
"""
import os

import numpy as np
import pytest

import beast_mcmc_amd as bm
import helpers
from beast_mcmc_amd.treelikelihood import BeagleTreeLikelihood, RESCALE_ALWAYS, RESCALE_DYNAMIC, RESCALE_NONE

pytestmark = pytest.mark.gpu
LEVELS = {"BEAGLE_MI355_NO_T64_WALK": "1"}
# bit-equality with the level kernels is a property of per-node scale factors; the default program folds the reciprocals of unstored nodes
# into one multiplication per stored node (planner.h FoldMap: other roundings, held to 1e-12 below and in tests/test_gpu_scale_fold.py)
PER_NODE = {"BEAGLE_MI355_NO_SCALE_FOLD": "1"}


def evaluate(wl, scheme, env=None, library=None, evaluations=2, partials=True):
    env = env or {}
    os.environ.update(env)
    try:
        tl = BeagleTreeLikelihood(wl, library=library, rescaling=scheme, delay_rescaling=False)
        raw = bm.beagle.Beagle.attach(tl)
        lnl = []
        for _ in range(evaluations):                       # (DYNAMIC: write mode first, then read mode on the flipped buffers)
            lnl.append(tl.getLogLikelihood())
            tl.makeDirty()
        stats = raw.walkStats() if library is None else None
        site = tl.getSiteLogLikelihoods()
        nodes = None
        if partials:                                        # every internal node as the last evaluation left it (unstored ones: on demand)
            tl.makeDirty(); tl.getLogLikelihood()
            nodes = np.array([raw.getPartials(tl.node_buffer_index(n)) for n in range(wl.tip_count, wl.tree.node_count)])
        tl.close()
    finally:
        for k in env:
            os.environ.pop(k, None)
    return lnl, site, nodes, stats


@pytest.mark.parametrize("S,C,P", [(61, 4, 600), (61, 4, 128 + 37), (64, 2, 333), (21, 3, 257), (33, 1, 96), (60, 4, 64), (63, 5, 31), (61, 2, 1)])
def test_walk_equals_level_kernels_bitwise_and_follows_oracle(S, C, P, oracle_lib):
    wl = helpers.random_workload(37, P, S, C, seed=2000 + S + C + P)
    lnl, site, nodes, st = evaluate(wl, RESCALE_DYNAMIC, PER_NODE)
    assert st["walks"] > 0 and st["fast_walks"] == 0, st                        # the read-mode evaluation ran on the walk ...
    assert st["stored"] < st["micro_ops"], st                                   # ... and not every node went to memory
    lv_lnl, lv_site, lv_nodes, lv_st = evaluate(wl, RESCALE_DYNAMIC, LEVELS)
    assert lv_st["walks"] == 0 and lv_st["stored"] == lv_st["micro_ops"], lv_st
    assert lnl == lv_lnl
    assert np.array_equal(site, lv_site)
    assert np.array_equal(nodes, lv_nodes)
    f_lnl, f_site, f_nodes, f_st = evaluate(wl, RESCALE_DYNAMIC)                # the default: folded factors
    assert f_st["walks"] > 0 and helpers.rel_err(f_lnl[1], lnl[1]) <= 1e-12 and np.max(np.abs(f_site - site) / np.abs(site)) <= 1e-12
    o_lnl, o_site, o_nodes, _ = evaluate(wl, RESCALE_DYNAMIC, library=oracle_lib)
    assert helpers.rel_err(lnl[1], o_lnl[1]) <= 1e-10 and helpers.rel_err(f_lnl[1], o_lnl[1]) <= 1e-10
    assert np.max(np.abs(site - o_site) / np.abs(o_site)) <= 1e-10
    scale = np.max(np.abs(o_nodes), axis=(1, 3), keepdims=True)                 # per node and pattern (getPartials: [category][pattern][state])
    assert np.max(np.abs(nodes - o_nodes) / scale) <= 1e-10


def test_without_rescaling_and_under_always(oracle_lib):
    """NONE: no scale buffers at all (the walk multiplies by nothing); ALWAYS: every list rescales in write mode — the level kernels, on
    operands the walk left unstored (materialised first), every evaluation."""
    wl = helpers.random_workload(30, 300, 61, 4, seed=77, root_to_tip=0.4)
    lnl, site, _, st = evaluate(wl, RESCALE_NONE, partials=False)
    lv_lnl, lv_site, _, _ = evaluate(wl, RESCALE_NONE, LEVELS, partials=False)
    assert st["walks"] > 0 and lnl == lv_lnl and np.array_equal(site, lv_site)
    o_lnl, o_site, _, _ = evaluate(wl, RESCALE_NONE, library=oracle_lib, partials=False)
    assert helpers.rel_err(lnl[0], o_lnl[0]) <= 1e-10
    a_lnl, a_site, _, a_st = evaluate(wl, RESCALE_ALWAYS, partials=False)
    oa_lnl, oa_site, _, _ = evaluate(wl, RESCALE_ALWAYS, library=oracle_lib, partials=False)
    assert a_st["scale_writes"] > 0 and helpers.rel_err(a_lnl[1], oa_lnl[1]) <= 1e-10
    assert np.max(np.abs(a_site - oa_site) / np.abs(oa_site)) <= 1e-10


def test_two_hundred_taxa_in_slices(oracle_lib):
    """Config C's tree size at a fraction of its patterns: the program is cut into slices that run side by side, launch after launch."""
    wl = helpers.random_workload(200, 1000, 61, 4, seed=6)
    lnl, site, _, st = evaluate(wl, RESCALE_DYNAMIC, PER_NODE, partials=False)
    lv_lnl, lv_site, _, _ = evaluate(wl, RESCALE_DYNAMIC, LEVELS, partials=False)
    assert st["walks"] >= 2 and st["stored"] < 0.8 * st["micro_ops"], st       # (the counters include the first, write-mode evaluation: every node stored)
    assert lnl == lv_lnl and np.array_equal(site, lv_site)
    o_lnl, o_site, _, _ = evaluate(wl, RESCALE_DYNAMIC, library=oracle_lib, partials=False)
    assert helpers.rel_err(lnl[1], o_lnl[1]) <= 1e-10 and np.max(np.abs(site - o_site) / np.abs(o_site)) <= 1e-10


@pytest.mark.parametrize("S", [61, 23])
def test_chain_of_moves(S, oracle_lib):
    """What a chain does, under DYNAMIC rescaling: node-height moves (partial lists: one path recomputed, its siblings read from memory or
    re-evaluated from their definitions), rejections (restoreState: index flips only), rate and model changes — every value against the
    same chain on the level kernels (bit for bit) and on the oracle (1e-10)."""
    from beast_mcmc_amd.inputs import substmodel
    wl = helpers.random_workload(40, 300, S, 4, seed=31)
    rng = np.random.default_rng(4)
    moves = []
    for step in range(14):
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
    for name, env, lib in (("walk", PER_NODE, None), ("folded", {}, None), ("levels", LEVELS, None), ("oracle", {}, oracle_lib)):
        os.environ.update(env)
        try:
            tl = BeagleTreeLikelihood(wl, library=lib, rescaling=RESCALE_DYNAMIC, delay_rescaling=False)
            runs[name] = chain(tl)
            if name == "walk":
                st = helpers.walk_stats(tl)
                assert st["walks"] > 0 and st["stored"] < st["micro_ops"], st
            tl.close()
        finally:
            for k in env:
                os.environ.pop(k, None)
    assert runs["walk"] == runs["levels"]
    for a, f, b in zip(runs["walk"], runs["folded"], runs["oracle"]):
        assert helpers.rel_err(a, b) <= 1e-10 and helpers.rel_err(f, a) <= 1e-12
