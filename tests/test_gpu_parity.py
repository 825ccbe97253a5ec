"""Parity of the HIP engine with the CPU oracle and with the reference's golden values, through the C ABI.

Tolerance (BASELINE.json north_star): per-tree log-likelihood within 1e-10 RELATIVE of the fp64 CPU path on the
same inputs.  Site log-likelihoods and partials are held to the same relative bound.
"""
import numpy as np
import pytest

import beast_mcmc_amd as bm
import helpers
from beast_mcmc_amd.inputs import synth
from beast_mcmc_amd.treelikelihood import (BeagleTreeLikelihood, POST_ORDER, RESCALE_ALWAYS, RESCALE_DELAYED,
                                           RESCALE_DYNAMIC, RESCALE_NONE, REVERSE_LEVEL_ORDER)
from test_oracle_golden import PRIMATES, fmt5, run_branch_specific, run_jar_smoke

pytestmark = pytest.mark.gpu

REL_TOL = 1e-10


def both(wl, oracle_lib, **kw):
    g = BeagleTreeLikelihood(wl, **kw)
    o = BeagleTreeLikelihood(wl, library=oracle_lib, **kw)
    return g, o


def assert_parity(g, o, what=""):
    a, b = g.getLogLikelihood(), o.getLogLikelihood()
    assert np.isfinite(b), what
    assert helpers.rel_err(a, b) <= REL_TOL, (what, a, b)
    sa, sb = g.getSiteLogLikelihoods(), o.getSiteLogLikelihoods()
    assert np.max(np.abs(sa - sb) / np.maximum(np.abs(sb), 1e-300)) <= REL_TOL, what
    return a, b


def test_small_and_ragged_shapes_on_the_walks(oracle_lib):
    """The corner shapes of the two pattern walks in one sweep — 4 states (kernels_walk4.hip) and 16..20 states on the T32 layout
    (kernels_mfma.hip k_walkT32): a single pattern, one pattern short of / past a tile and a workgroup, two taxa (one
    operation), one and sixteen rate categories, with and without rescaling, a second evaluation after a branch move on top
    (read mode, unstored nodes, the plan cache)."""
    from beast_mcmc_amd.treelikelihood import RESCALE_DYNAMIC
    checked = 0
    for S in (4, 16, 17, 19, 20):
        for C in (1, 3, 16):
            for T, P in ((2, 1), (2, 4), (3, 15), (5, 33), (9, 64), (9, 127), (14, 129)) + (((2, 33), (3, 31)) if S >= 16 else ()):
                if S != 4 and C == 16 and P > 64:
                    continue                                   # (keeps the sweep short: the 16-category workgroups are covered at small P)
                wl = helpers.random_workload(T, P, S, C, seed=1000 + 31 * S + 7 * C + T + P)
                for scheme in (RESCALE_NONE, RESCALE_DYNAMIC):
                    g, o = both(wl, oracle_lib, rescaling=scheme, delay_rescaling=False)
                    assert_parity(g, o, "S=%d C=%d T=%d P=%d scheme=%s" % (S, C, T, P, scheme))
                    rates = np.linspace(0.7, 1.4, wl.tree.node_count)
                    for t in (g, o):
                        t.storeState(); t.set_branch_rates(rates)
                    assert_parity(g, o, "S=%d C=%d T=%d P=%d scheme=%s, after a move" % (S, C, T, P, scheme))
                    g.close(); o.close()
                    checked += 1
    assert checked >= 170


# ---- the reference's golden vectors, through the HIP engine -----------------------------------

@pytest.mark.parametrize("case", PRIMATES["tree_data_likelihood_test"], ids=lambda c: c["name"])
def test_golden_tree_data_likelihood(case):
    wl = helpers.primates_case(case, site_model="new")
    tl = BeagleTreeLikelihood(wl)
    assert fmt5(tl.getLogLikelihood()) == fmt5(case["lnL"])
    tl.close()


@pytest.mark.parametrize("case", PRIMATES["likelihood_test"], ids=lambda c: c["name"])
def test_golden_likelihood_test(case):
    wl = helpers.primates_case(case, site_model="old")
    tl = BeagleTreeLikelihood(wl, rescaling=RESCALE_ALWAYS, delay_rescaling=False, traversal=POST_ORDER)
    assert fmt5(tl.getLogLikelihood()) == fmt5(case["lnL"])
    tl.close()


def test_golden_jar_smoke(engine_lib):
    lnl, g = run_jar_smoke(engine_lib)
    assert fmt5(lnl) == fmt5(g["lnL"])


@pytest.mark.parametrize("case", [0, 1])
def test_golden_branch_specific(case, engine_lib):
    g = helpers.golden("branch_specific.json")
    c = g["cases"][case]
    lnl, _ = run_branch_specific(engine_lib, c["stem_weight"])
    assert abs(lnl - c["lnL"]) < 5e-13, (lnl, c["lnL"])


def test_golden_transition_probabilities(engine_lib, oracle_lib):
    """The reference's known-answer transition-probability matrices (HKYTest / TN93Test / GeneralF81Test, 1e-10) through the HIP
    engine's updateTransitionMatrices, and the engine against the oracle on the same eigen systems."""
    from test_oracle_golden import check_transition_probabilities, run_transition_probabilities
    rows = run_transition_probabilities(engine_lib)
    check_transition_probabilities(rows)
    for (c, a0, a1), (_, o0, o1) in zip(rows, run_transition_probabilities(oracle_lib)):
        assert np.abs(a0 - o0).max() <= 1e-14 and np.abs(a1 - o1).max() <= 1e-14, c["source"]


def test_golden_epoch_convolution(engine_lib):
    """tests/TestXML/testEpochConvolutionOrder.xml:179-193 — pins convolveTransitionMatrices and its order."""
    from test_oracle_golden import run_epoch_convolution
    lnl, g = run_epoch_convolution(engine_lib)
    assert abs(lnl - g["lnL"]) < 1e-5, lnl


# ---- engine vs oracle on seeded workloads -------------------------------------------------------

@pytest.mark.parametrize("S,C,T,P", [(4, 4, 33, 1000), (4, 1, 17, 257), (4, 2, 9, 64), (4, 8, 12, 300), (4, 10, 8, 130),
                                     (20, 4, 12, 333), (20, 1, 7, 50), (61, 4, 9, 150), (61, 2, 5, 33), (3, 3, 6, 100),
                                     (2, 1, 12, 77), (7, 5, 10, 200),
                                     # MFMA / T32-layout path (16..64 states), incl. partial 4-tiles and ragged 32-tiles
                                     (16, 2, 6, 70), (21, 3, 6, 95), (60, 1, 5, 31), (64, 2, 5, 65), (20, 4, 30, 1000),
                                     # more than 64 states (large discrete-trait spaces): the general kernels, matrices staged in LDS
                                     # up to ~95 states, read from L2 above (kernels.hip k_pruneGeneral<false>, k_transitionBig)
                                     (65, 2, 6, 40), (90, 1, 5, 23), (100, 2, 5, 30), (128, 1, 6, 21), (200, 1, 4, 9)])
@pytest.mark.parametrize("scheme", [RESCALE_NONE, RESCALE_ALWAYS])
def test_engine_matches_oracle(S, C, T, P, scheme, oracle_lib):
    wl = helpers.random_workload(T, P, S, C, seed=100 + S + C)
    g, o = both(wl, oracle_lib, rescaling=scheme, delay_rescaling=False)
    assert_parity(g, o, "S=%d C=%d" % (S, C))
    # every internal node's partials, and the scale factors written for it
    for node in range(wl.tip_count, 2 * wl.tip_count - 1):
        pg = _partials(g, node)
        po = _partials(o, node)
        # per pattern, relative to that pattern's largest partial: tiny entries inherit the ABSOLUTE rounding error of the
        # transition-matrix entries they came from (exp() differs by an ulp between libm and the device), so an
        # element-wise relative bound is not meaningful for them
        scale = np.maximum(np.abs(po).max(axis=(0, 2), keepdims=True), 1e-300)
        assert np.max(np.abs(pg - po) / scale) <= REL_TOL, node
    g.close(); o.close()


def _raw(tl):
    """A Beagle binding object over the host driver's existing instance (no new instance)."""
    b = bm.beagle.Beagle.__new__(bm.beagle.Beagle)
    b.lib = tl.engine
    b._f = tl.engine.fn
    b.instance = tl.instance
    b.stateCount, b.patternCount, b.categoryCount = tl.state_count, tl.pattern_count, tl.category_count
    return b


def _partials(tl, node):
    return _raw(tl).getPartials(tl.node_buffer_index(node), bm.beagle.NONE)


@pytest.mark.parametrize("traversal", [POST_ORDER, REVERSE_LEVEL_ORDER])
@pytest.mark.parametrize("kind", ["coalescent", "yule", "caterpillar"])
def test_tree_shapes_and_traversals(traversal, kind, oracle_lib):
    wl = helpers.random_workload(40, 500, 4, 4, seed=7, tree_kind=kind)
    g, o = both(wl, oracle_lib, rescaling=RESCALE_ALWAYS, delay_rescaling=False, traversal=traversal)
    assert_parity(g, o, kind)
    g.close(); o.close()


def test_scale_factors_and_cumulative_buffer(oracle_lib):
    wl = helpers.random_workload(25, 400, 4, 4, seed=3)
    g, o = both(wl, oracle_lib, rescaling=RESCALE_ALWAYS, delay_rescaling=False)
    assert_parity(g, o)
    rg, ro = _raw(g), _raw(o)
    for node in range(wl.tip_count, 2 * wl.tip_count - 1):
        a = rg.getLogScaleFactors(g.node_scale_index(node))
        b = ro.getLogScaleFactors(o.node_scale_index(node))
        assert np.max(np.abs(a - b)) <= 1e-12
    ca = rg.getLogScaleFactors(g.cumulative_scale_index())
    cb = ro.getLogScaleFactors(o.cumulative_scale_index())
    assert np.max(np.abs(ca - cb)) <= 1e-9
    # un-scaled partials read back through getPartials(buffer, cumulativeScaleIndex)
    pa = rg.getPartials(g.root_buffer_index(), g.cumulative_scale_index())
    pb = ro.getPartials(o.root_buffer_index(), o.cumulative_scale_index())
    assert np.max(np.abs(pa - pb) / np.maximum(np.abs(pb), 1e-300)) <= 1e-9
    g.close(); o.close()


def test_underflow_triggers_rescaling_retry(oracle_lib):
    """DYNAMIC + delayed scaling: the first evaluation underflows (lnL = -inf), the host retries with
    rescaling on (BeagleTreeLikelihood.java:1059-1113) and later evaluations read the stored factors."""
    # Yule tree, 900 tips, saturated branches: about -960 log-units per pattern, so unscaled fp64 partials underflow
    wl = helpers.random_workload(900, 300, 4, 4, seed=11, root_to_tip=20.0, tree_kind="yule")
    g, o = both(wl, oracle_lib, rescaling=RESCALE_DYNAMIC, delay_rescaling=True)
    assert_parity(g, o)
    assert g.counters()["rescale_retries"] == 1 and g.counters()["ever_underflowed"] == 1
    assert o.counters()["rescale_retries"] == 1
    # steady state: read-mode evaluations after a parameter change
    for step in range(3):
        new = wl.tree.height[wl.tree.root] * (1.0 + 0.01 * (step + 1))
        for t in (g, o):
            t.set_node_height(wl.tree.root, new)
        assert_parity(g, o, "step %d" % step)
    ops = g.last_operations()
    assert len(ops) == 1 and ops[0][1] == -1 and ops[0][2] >= 0      # a single read-scale op: the root
    g.close(); o.close()


def test_partial_updates_store_restore(oracle_lib):
    """MCMC-style protocol: store, perturb one node (dirty path only), evaluate, reject -> restore, re-evaluate."""
    wl = helpers.random_workload(30, 700, 4, 4, seed=5)
    g, o = both(wl, oracle_lib, rescaling=RESCALE_ALWAYS, delay_rescaling=False)
    base_g, base_o = assert_parity(g, o)
    rng = np.random.default_rng(0)
    tree = wl.tree
    for it in range(6):
        node = int(rng.integers(wl.tip_count, 2 * wl.tip_count - 1))
        lo = max(tree.height[tree.left[node]], tree.height[tree.right[node]])
        hi = tree.height[tree.parent[node]] if tree.parent[node] >= 0 else tree.height[node] * 1.2
        h = lo + (hi - lo) * rng.random()
        for t in (g, o):
            t.storeState()
            t.set_node_height(node, h)
        assert_parity(g, o, "proposal %d" % it)
        assert g.counters()["last_op_count"] <= tree.depth()          # only the dirty path to the root
        for t in (g, o):
            t.restoreState()
            t.set_node_height(node, tree.height[node])    # the tree model restores itself in BEAST
        a, b = assert_parity(g, o, "after restore %d" % it)
        assert helpers.rel_err(a, base_g) <= 1e-12
    g.close(); o.close()


def test_run_to_run_determinism():
    """Checkpoint/resume in the reference re-evaluates lnL and requires identical digits
    (BeastCheckpointer.java:201-262): the engine must be bitwise reproducible."""
    wl = helpers.random_workload(50, 5000, 4, 4, seed=9)
    vals = []
    for _ in range(3):
        tl = BeagleTreeLikelihood(wl, rescaling=RESCALE_ALWAYS, delay_rescaling=False)
        vals.append(tl.getLogLikelihood())
        tl.makeDirty()
        vals.append(tl.getLogLikelihood())
        tl.close()
    assert len(set(vals)) == 1, vals


def test_post_order_list_is_levelised(oracle_lib):
    """BeagleTreeLikelihood always sends POST-ORDER op lists; the engine levelises them itself and must
    give the same answer as for BEAST's level-ordered list."""
    wl = helpers.random_workload(64, 900, 4, 4, seed=21)
    a = BeagleTreeLikelihood(wl, traversal=POST_ORDER, rescaling=RESCALE_ALWAYS, delay_rescaling=False)
    b = BeagleTreeLikelihood(wl, traversal=REVERSE_LEVEL_ORDER, rescaling=RESCALE_ALWAYS, delay_rescaling=False)
    assert a.getLogLikelihood() == b.getLogLikelihood()
    a.close(); b.close()


@pytest.mark.parametrize("S", [4, 20])
def test_tip_partials_and_ambiguity(S, oracle_lib):
    """setTipPartials (replicated over categories) must agree with compact states for unambiguous data
    (S = 20 also exercises the API <-> T32 layout conversion of the MFMA path)."""
    wl = helpers.random_workload(10, 300, S, 4, seed=2)
    g, o = both(wl, oracle_lib, rescaling=RESCALE_NONE)
    ref = g.getLogLikelihood()
    for t in (g, o):
        for tip in (0, 3, 7):
            st = wl.tip_states[tip]
            part = np.zeros((wl.pattern_count, S))
            known = st < S
            part[np.arange(wl.pattern_count)[known], st[known]] = 1.0
            part[~known] = 1.0
            t._chk(t.h.btlSetTipPartials(t.ptr, tip, part.ctypes.data_as(__import__("ctypes").POINTER(__import__("ctypes").c_double))), "setTipPartials")
        t.makeDirty()
    a, b = assert_parity(g, o)
    assert helpers.rel_err(a, ref) <= 1e-12
    g.close(); o.close()


def test_error_codes(engine_lib):
    B = bm.beagle
    b = B.Beagle(3, 5, 3, 4, 10, 1, 4, 1, 2)
    with pytest.raises(B.BeagleException) as e:
        b.updatePartials([9, -1, -1, 0, 0, 1, 1], 1, B.NONE)
    assert e.value.code == -5
    with pytest.raises(B.BeagleException) as e:
        b.updatePartials([3, -1, -1, 0, 0, 1, 1], 1, B.NONE)        # tips never set
    assert e.value.code == -5
    with pytest.raises(B.BeagleException) as e:
        b.setEigenDecomposition(7, np.eye(4), np.eye(4), np.zeros(4))
    assert e.value.code == -5
    b.setCPUThreadCount(8)                                          # must succeed on a GPU instance
    assert (b.details.flags & (1 << 27)) == 0                       # not FRAMEWORK_CPU -> BEAST sends level order
    assert (b.details.flags & (1 << 16)) != 0                       # PROCESSOR_GPU
    b.finalize()
    with pytest.raises(B.BeagleException):
        B.Beagle(3, 5, 3, 4, 10, 1, 4, 1, 2, resourceList=(0,))     # resource 0 = CPU: not provided
    rl = engine_lib.resource_list()
    assert rl[0][0] == "CPU" and len(rl) >= 2


# ---- full-size properties (BASELINE.json config A: 1000 taxa x 1e5 patterns) ---------------------

@pytest.fixture(scope="module")
def config_a():
    return synth.config_a()


def test_config_a_sampled_against_oracle(config_a, oracle_lib):
    """Patterns are independent given the tree: the oracle evaluates a random 1 % sample of the 1e5 patterns
    (seconds on the CPU) and must agree with the engine's site log-likelihoods for those patterns."""
    wl = config_a
    g = BeagleTreeLikelihood(wl, rescaling=RESCALE_DYNAMIC, delay_rescaling=True)
    lnl = g.getLogLikelihood()
    assert np.isfinite(lnl)
    site = g.getSiteLogLikelihoods()
    assert helpers.rel_err(float(np.dot(site, wl.weights)), lnl) <= 1e-12
    idx = helpers.sample_with_tail(wl.pattern_count, 1000, 4)           # 1 % at random + always the last 256 patterns
    sub = synth.Workload("A-sample", wl.tree, wl.eig, wl.freqs, wl.cat_rates, wl.cat_weights,
                         np.ascontiguousarray(wl.tip_states[:, idx]), wl.weights[idx], 4)
    o = BeagleTreeLikelihood(sub, library=oracle_lib, rescaling=RESCALE_DYNAMIC, delay_rescaling=True)
    o.getLogLikelihood()
    so = o.getSiteLogLikelihoods()
    assert np.max(np.abs(site[idx] - so) / np.abs(so)) <= REL_TOL
    # read-mode (steady-state DYNAMIC) evaluation must reproduce the recompute-mode value
    g.makeDirty()
    again = g.getLogLikelihood()
    assert helpers.rel_err(again, lnl) <= 1e-12
    o.close(); g.close()


def test_config_a_shards_sum_to_whole(config_a):
    """Multi-GPU row (e): contiguous pattern shards (Patterns.java:142-167) evaluated independently must sum
    to the unsharded lnL to ~1e-12 (the all-reduce is a plain sum of per-shard doubles)."""
    from beast_mcmc_amd.inputs import patterns
    wl = config_a
    whole = BeagleTreeLikelihood(wl)
    total = whole.getLogLikelihood()
    whole.close()
    parts = 0.0
    for (s, e) in patterns.shard_bounds(wl.pattern_count, 8)[:8]:
        tl = BeagleTreeLikelihood(wl.shard(s, e))
        parts += tl.getLogLikelihood()
        tl.close()
    assert helpers.rel_err(parts, total) <= 1e-11


def test_concurrent_instances_from_two_threads(oracle_lib):
    """BEAST drives different instances concurrently from CompoundLikelihood's thread pool
    (src/dr/inference/model/CompoundLikelihood.java:64-81, 208-214): two instances evaluated from two threads at the
    same time must give exactly what each gives alone."""
    import threading
    wls = [helpers.random_workload(40, 3000, 4, 4, seed=500), helpers.random_workload(25, 2500, 20, 2, seed=501)]

    def chain(wl, out):
        tl = BeagleTreeLikelihood(wl, rescaling=RESCALE_DYNAMIC, delay_rescaling=False)
        vals = []
        for k in range(25):
            tl.storeState()
            tl.set_branch_rates(np.full(wl.tree.node_count, 1.0 + 0.01 * k))
            vals.append(tl.getLogLikelihood())
        tl.close()
        out.append(vals)

    serial = []
    for wl in wls:
        chain(wl, serial)
    par = [[], []]
    threads = [threading.Thread(target=chain, args=(wls[i], par[i])) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert par[0][0] == serial[0] and par[1][0] == serial[1]
    o = BeagleTreeLikelihood(wls[0], library=oracle_lib, rescaling=RESCALE_DYNAMIC, delay_rescaling=False)
    o.storeState(); o.set_branch_rates(np.full(wls[0].tree.node_count, 1.0))
    assert helpers.rel_err(serial[0][0], o.getLogLikelihood()) <= REL_TOL
    o.close()


@pytest.mark.parametrize("traversal", [POST_ORDER, REVERSE_LEVEL_ORDER])
def test_ladder_tree_of_5000_taxa(traversal, oracle_lib):
    """4999 dependency levels: the planner's emission is iterative (no native-stack recursion under a JVM thread) and the
    whole ladder runs as one chain in registers; a first child must never be fetched from a buffer the micro-operation right
    before it stores (the kernels request it one stage early — ADVICE round 2, planner.cpp / engine_walk.cpp runPlan)."""
    rng = np.random.default_rng(5)
    pi = rng.dirichlet(np.full(4, 10.0))
    eig = bm.substmodel.gtr(rng.gamma(2.0, 1.0, size=6) + 0.1, pi)
    wl = synth.make_workload("ladder", 5000, 300, eig, pi, alpha=0.7, categories=4, seed=12, tree_kind="caterpillar",
                             root_to_tip=2.0, unknown_fraction=0.02)
    g, o = both(wl, oracle_lib, rescaling=RESCALE_ALWAYS, delay_rescaling=False, traversal=traversal)
    assert_parity(g, o, "ladder write mode")
    for tl in (g, o):
        tl.makeDirty()
    assert_parity(g, o, "ladder again")
    # the same with virtual buffers off: every node stored, every internal child the previous micro-operation's result
    g.close(); o.close()


def test_ladder_tree_every_node_stored(oracle_lib, monkeypatch):
    monkeypatch.setenv("BEAGLE_MI355_NO_VIRTUAL", "1")
    rng = np.random.default_rng(6)
    pi = rng.dirichlet(np.full(4, 10.0))
    eig = bm.substmodel.gtr(rng.gamma(2.0, 1.0, size=6) + 0.1, pi)
    wl = synth.make_workload("ladder", 4500, 200, eig, pi, alpha=0.7, categories=2, seed=13, tree_kind="caterpillar",
                             root_to_tip=2.0, unknown_fraction=0.02)
    g, o = both(wl, oracle_lib, rescaling=RESCALE_DYNAMIC, delay_rescaling=False)
    assert_parity(g, o, "ladder, all stored")
    for node in (wl.tip_count, wl.tip_count + 1, 2 * wl.tip_count - 3, 2 * wl.tip_count - 2):
        pg, po = _partials(g, node), _partials(o, node)
        scale = np.maximum(np.abs(po).max(axis=(0, 2), keepdims=True), 1e-300)
        assert np.max(np.abs(pg - po) / scale) <= REL_TOL, node
    g.close(); o.close()


def test_all_tips_as_partials_stay_unstored(oracle_lib):
    """A useAmbiguities-style analysis: EVERY tip is uploaded as partials (soft ambiguity codes;
    BeagleTreeLikelihood.java:497-509).  Definitions of unstored nodes read such a tip as a memory leaf (planner.h
    leafPartials), so the evaluation keeps the shape it has with compact states; a changed tip (sequence-error models re-send
    tips every evaluation, :917-930) materialises what was defined on it."""
    import ctypes as C
    wl = helpers.random_workload(60, 1500, 4, 4, seed=31)
    rng = np.random.default_rng(3)
    g, o = both(wl, oracle_lib, rescaling=RESCALE_DYNAMIC, delay_rescaling=False)

    def upload(tips, soft):
        for tip in tips:
            st = wl.tip_states[tip]
            part = np.full((wl.pattern_count, 4), soft)
            known = st < 4
            part[np.arange(wl.pattern_count)[known], st[known]] = 1.0
            part[~known] = 1.0
            part = np.ascontiguousarray(part)
            for t in (g, o):
                t._chk(t.h.btlSetTipPartials(t.ptr, tip, part.ctypes.data_as(C.POINTER(C.c_double))), "setTipPartials")
    upload(range(wl.tip_count), 0.02)
    raw = bm.beagle.Beagle.attach(g)
    for t in (g, o):
        t.makeDirty()
    assert_parity(g, o, "tips as partials, write mode")
    raw.kernelTimer(True)
    for t in (g, o):
        t.makeDirty()
    assert_parity(g, o, "tips as partials, read mode")
    stats = raw.walkStats()
    raw.kernelTimer(False)
    assert stats["stored"] * 4 < stats["micro_ops"], stats          # most nodes are defined, not stored
    upload([5, 17], 0.07)                                            # new data for two tips
    for t in (g, o):
        t.makeDirty()
    assert_parity(g, o, "two tips changed")
    for node in range(wl.tip_count, 2 * wl.tip_count - 1, 5):
        pg, po = _partials(g, node), _partials(o, node)
        scale = np.maximum(np.abs(po).max(axis=(0, 2), keepdims=True), 1e-300)
        assert np.max(np.abs(pg - po) / scale) <= REL_TOL, node
    g.close(); o.close()


@pytest.mark.parametrize("S,T,P", [(4, 40, 3000), (20, 16, 700)])
def test_batched_read_back_of_every_internal_node(S, T, P, oracle_lib):
    """beagleMi355GetPartialsBatch (what AncestralStateBeagleTreeLikelihood.java:414-542 does once per logged
    sample, in one call): every internal node with its scale factors folded in, against the oracle's getPartials node by node,
    and bitwise against the engine's own per-node getPartials.  The 4-state case reads back nodes that were never stored (one
    batched materialisation); sizes chosen so that the sweep spans several pipeline chunks."""
    wl = helpers.random_workload(T, P, S, 4, seed=70 + S)
    g, o = both(wl, oracle_lib, rescaling=RESCALE_ALWAYS, delay_rescaling=False)
    assert_parity(g, o)
    rg, ro = _raw(g), _raw(o)
    nodes = list(range(wl.tip_count, 2 * wl.tip_count - 1))
    bufs = [g.node_buffer_index(n) for n in nodes]
    scales = [g.node_scale_index(n) for n in nodes]
    batch = rg.getPartialsBatch(bufs, scales)
    assert batch.shape == (len(nodes), 4, P, S)
    for k, n in enumerate(nodes):
        po = ro.getPartials(o.node_buffer_index(n), o.node_scale_index(n))
        scale = np.maximum(np.abs(po).max(axis=(0, 2), keepdims=True), 1e-300)
        assert np.max(np.abs(batch[k] - po) / scale) <= REL_TOL, n
        assert np.array_equal(batch[k], rg.getPartials(bufs[k], scales[k])), n
    unscaled = rg.getPartialsBatch(bufs, None)
    assert np.array_equal(unscaled[3], rg.getPartials(bufs[3], bm.beagle.NONE))
    assert_parity(g, o, "after the read-back")
    g.close(); o.close()


@pytest.mark.parametrize("S", [4, 7, 20, 61])
def test_complex_eigen_instance_against_oracle(S, oracle_lib):
    """An EIGEN_COMPLEX instance (BeagleTreeLikelihood.java:353-355, ComplexSubstitutionModel.java:121-187: the asymmetric
    discrete-trait models of the phylogeography analyses): transition matrices of a model with complex-conjugate eigenvalue
    pairs against the oracle (itself pinned by scipy's matrix exponential, tests/test_oracle_golden.py) and against expm
    directly, then a whole tree likelihood with the stationary distribution at the root — on the 4-state walk, the general
    kernel (7) and the MFMA path (20, 61)."""
    import scipy.linalg
    from test_oracle_golden import _cyclic_model
    from beast_mcmc_amd.inputs import trees
    qn, pi, eig = _cyclic_model(S, 3 + S)
    b = bm.beagle.Beagle(3, 5, 3, S, 10, 1, 4, 2, 0, requirementFlags=bm.beagle.FLAG_EIGEN_COMPLEX)
    assert b.details.flags & bm.beagle.FLAG_EIGEN_COMPLEX and not b.details.flags & bm.beagle.FLAG_EIGEN_REAL
    b.setEigenDecomposition(0, eig.evec, eig.ievc, eig.evals)
    b.setCategoryRates([0.5, 1.7])
    b.updateTransitionMatrices(0, [0, 1], None, None, [0.3, 1.1], 2)
    got = b.getTransitionMatrix(1).reshape(2, S, S)
    for c, r in enumerate((0.5, 1.7)):
        assert np.max(np.abs(got[c] - scipy.linalg.expm(qn * 1.1 * r))) <= 1e-12
    # imaginary parts must come as adjacent conjugate pairs: a lone one (here: the second of a pair zeroed) is refused, and the
    # eigen system the instance holds stays what it was (round-3 advisor finding: the kernel would have read past the block)
    ev = np.array(eig.evals, dtype=float)
    im = np.nonzero(ev[S:])[0]
    if len(im):
        bad = ev.copy(); bad[S + im[1]] = 0.0
        with pytest.raises(bm.beagle.BeagleException):
            b.setEigenDecomposition(0, eig.evec, eig.ievc, bad)
        b.updateTransitionMatrices(0, [2], None, None, [1.1], 1)
        assert np.array_equal(b.getTransitionMatrix(2).reshape(2, S, S), got)
    b.finalize()
    rng = np.random.default_rng(S)
    T, P = 14, 333
    tree = trees.coalescent_tree(T, rng, root_height=0.7)
    tips = rng.integers(0, S, size=(T, P)).astype(np.int32)
    tips[rng.random(tips.shape) < 0.05] = S
    wl = synth.Workload("complex-S%d" % S, tree, eig, pi, [0.4, 1.0, 1.6], [0.3, 0.4, 0.3], tips, rng.integers(1, 5, size=P).astype(np.float64), S)
    g, o = both(wl, oracle_lib, rescaling=RESCALE_ALWAYS, delay_rescaling=False, requirement_flags=bm.beagle.FLAG_EIGEN_COMPLEX)
    assert_parity(g, o, "complex S=%d" % S)
    for node in (T, 2 * T - 2):
        pg, po = _partials(g, node), _partials(o, node)
        scale = np.maximum(np.abs(po).max(axis=(0, 2), keepdims=True), 1e-300)
        assert np.max(np.abs(pg - po) / scale) <= REL_TOL
    g.close(); o.close()
