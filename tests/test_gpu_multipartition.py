"""the multi-partition entry points, driven the way MultiPartitionDataLikelihoodDelegate drives them
(src/dr/evomodel/treedatalikelihood/MultiPartitionDataLikelihoodDelegate.java:520-553 partitions = contiguous pattern
ranges of ONE instance; :835 setCategoryRatesWithIndex; :880-887 updateTransitionMatricesWithMultipleModels;
:972-997 9-int op tuples; :1016-1017 scale factors by partition; :1074-1083 root by partition).

Expected values: each partition evaluated on its own by the CPU oracle through the single-partition protocol."""
import numpy as np
import pytest

import beast_mcmc_amd as bm
import helpers
from beast_mcmc_amd.inputs import substmodel, synth
from beast_mcmc_amd.inputs.siterates import GammaSiteRateModel
from beast_mcmc_amd.treelikelihood import BeagleTreeLikelihood, RESCALE_ALWAYS, RESCALE_NONE

pytestmark = pytest.mark.gpu
NONE = bm.beagle.NONE


def two_partitions(S, T, sizes, seed):
    rng = np.random.default_rng(seed)
    wls = []
    tree = None
    for k, n in enumerate(sizes):
        if S == 4:
            pi = rng.dirichlet(np.full(4, 8.0))
            eig = substmodel.gtr(rng.gamma(2.0, 1.0, size=6) + 0.1, pi)
        else:
            eig, pi = substmodel.random_reversible(S, rng)
        if tree is None:
            wl = synth.make_workload("part0", T, n, eig, pi, alpha=0.4 + 0.5 * k, categories=4, seed=seed)
            tree = wl.tree
        else:
            rates, props = GammaSiteRateModel(alpha=0.4 + 0.5 * k, gamma_categories=4).category_rates_and_proportions()
            tips = synth.simulate_unique_patterns(tree, eig, np.asarray(pi), rates, props, n, rng).astype(np.int32)
            tips[rng.random(tips.shape) < 0.03] = S
            wl = synth.Workload("part%d" % k, tree, eig, pi, rates, props, np.ascontiguousarray(tips),
                                rng.integers(1, 9, size=n).astype(np.float64), S)
        wls.append(wl)
    return tree, wls


@pytest.mark.parametrize("S", [4, 20, 61])
@pytest.mark.parametrize("scaling", [False, True])
def test_multi_partition_protocol(S, scaling, oracle_lib):
    T = 9
    tree, wls = two_partitions(S, T, [150, 77], seed=40 + S)
    K = len(wls)
    P = sum(w.pattern_count for w in wls)
    nodes = 2 * T - 1
    mat_per_part = nodes
    b = bm.beagle.Beagle(T, T + (T - 1), T, S, P, K, K * mat_per_part, 4, T + 1)
    try:
        for t in range(T):
            b.setTipStates(t, np.concatenate([w.tip_states[t] for w in wls]))
        b.setPatternWeights(np.concatenate([w.weights for w in wls]))
        part_of = np.concatenate([np.full(w.pattern_count, k, dtype=np.int32) for k, w in enumerate(wls)])
        b.setPatternPartitions(K, part_of)
        eig_idx, rate_idx, mat_idx, lens = [], [], [], []
        for k, w in enumerate(wls):
            b.setEigenDecomposition(k, w.eig.evec, w.eig.ievc, w.eig.evals)
            b.setCategoryRatesWithIndex(k, w.cat_rates)
            b.setCategoryWeights(k, w.cat_weights)
            b.setStateFrequencies(k, w.freqs)
            for n in range(nodes):
                if n != tree.root:
                    eig_idx.append(k); rate_idx.append(k); mat_idx.append(k * mat_per_part + n); lens.append(tree.branch_length(n))
        b.updateTransitionMatricesWithMultipleModels(eig_idx, rate_idx, mat_idx, None, None, lens, len(lens))
        ops = []
        scale_idx = []
        for n in tree.postorder():
            if n < T:
                continue
            l, r = int(tree.left[n]), int(tree.right[n])
            for k in range(K):
                ws = (n - T) if scaling else NONE
                ops += [n, ws, NONE, l, k * mat_per_part + l, r, k * mat_per_part + r, k, NONE]
            scale_idx.append(n - T)
        b.updatePartialsByPartition(ops, len(ops) // 9)
        cum = T - 1 if scaling else NONE                   # one cumulative buffer, each partition owns its pattern range
        if scaling:
            for k in range(K):
                b.resetScaleFactorsByPartition(cum, k)
                b.accumulateScaleFactorsByPartition(scale_idx, len(scale_idx), cum, k)
        by_part = np.zeros(K)
        total = [0.0]
        b.calculateRootLogLikelihoodsByPartition([tree.root] * K, list(range(K)), list(range(K)), [cum] * K,
                                                 list(range(K)), K, 1, by_part, total)
        site = b.getSiteLogLikelihoods()
    finally:
        b.finalize()
    expect = []
    off = 0
    for k, w in enumerate(wls):
        o = BeagleTreeLikelihood(w, library=oracle_lib, rescaling=RESCALE_ALWAYS if scaling else RESCALE_NONE,
                                 delay_rescaling=False)
        v = o.getLogLikelihood()
        so = o.getSiteLogLikelihoods()
        o.close()
        expect.append(v)
        assert helpers.rel_err(by_part[k], v) <= 1e-10, (k, by_part[k], v)
        assert np.max(np.abs(site[off:off + w.pattern_count] - so) / np.abs(so)) <= 1e-10
        off += w.pattern_count
    assert helpers.rel_err(total[0], sum(expect)) <= 1e-10


@pytest.mark.parametrize("S", [4, 20, 61])
def test_partitioned_instance_partial_updates_with_per_partition_flips(S, oracle_lib):
    """What MultiPartitionDataLikelihoodDelegate does between full evaluations: ONE partition's branch changes, only that
    partition's path to the root is re-evaluated, into ITS alternate buffers (partialBufferHelper[i], one per partition,
    MultiPartitionDataLikelihoodDelegate.java:972-997), and a rejected proposal flips back.  The engine keeps definitions of
    unstored nodes per (buffer, partition) (planner.h); the reference here is one oracle instance per partition given the
    same operations as 7-int tuples.  Checked after every step: the per-partition log-likelihoods; at the end: the partials
    of every internal node of every partition (the engine materialises what it had not stored).  20 states: the partitions
    (300, 77 and 140 patterns) do not end at tile boundaries — a tile of 32 patterns that straddles two partitions is walked
    once per partition (kernels_mfma.hip k_walkT32)."""
    T, C = 24, 4
    tree, wls = two_partitions(S, T, [300, 77, 140], seed=77)
    K = len(wls)
    P = sum(w.pattern_count for w in wls)
    nodes = 2 * T - 1
    starts = np.concatenate([[0], np.cumsum([w.pattern_count for w in wls])])
    eng = bm.beagle.Beagle(T, T + 2 * (T - 1), T, S, P, K, 2 * K * nodes, C, T)
    ora = [bm.beagle.Beagle(T, T + 2 * (T - 1), T, S, w.pattern_count, 1, 2 * nodes, C, T, library=oracle_lib) for w in wls]
    rng = np.random.default_rng(5)
    try:
        for t in range(T):
            eng.setTipStates(t, np.concatenate([w.tip_states[t] for w in wls]))
            for k, w in enumerate(wls):
                ora[k].setTipStates(t, w.tip_states[t])
        eng.setPatternWeights(np.concatenate([w.weights for w in wls]))
        eng.setPatternPartitions(K, np.concatenate([np.full(w.pattern_count, k, dtype=np.int32) for k, w in enumerate(wls)]))
        for k, w in enumerate(wls):
            ora[k].setPatternWeights(w.weights)
            for b_, e_ in ((eng, k), (ora[k], 0)):
                b_.setEigenDecomposition(e_, w.eig.evec, w.eig.ievc, w.eig.evals)
                b_.setCategoryWeights(e_, w.cat_weights)
                b_.setStateFrequencies(e_, w.freqs)
            eng.setCategoryRatesWithIndex(k, w.cat_rates)
            ora[k].setCategoryRates(w.cat_rates)
        lens = np.array([tree.branch_length(n) if n != tree.root else 0.0 for n in range(nodes)])
        pflip = np.zeros((K, nodes), dtype=int)
        mflip = np.zeros((K, nodes), dtype=int)
        pb = lambda k, n: n if n < T else T + 2 * (n - T) + pflip[k][n]
        mbe = lambda k, n: (k * nodes + n) * 2 + mflip[k][n]          # engine: all partitions' matrices in one index space
        mbo = lambda k, n: n * 2 + mflip[k][n]
        branches = [n for n in range(nodes) if n != tree.root]

        def matrices(k, which):
            eng.updateTransitionMatricesWithMultipleModels([k] * len(which), [k] * len(which), [mbe(k, n) for n in which], None, None,
                                                           [lens[n] for n in which], len(which))
            ora[k].updateTransitionMatrices(0, [mbo(k, n) for n in which], None, None, [lens[n] for n in which], len(which))

        def ops_for(k, order, write):
            e9, o7 = [], []
            for n in order:
                l, r = int(tree.left[n]), int(tree.right[n])
                ws, rs = (n - T, NONE) if write else (NONE, n - T)
                e9 += [pb(k, n), ws, rs, pb(k, l), mbe(k, l), pb(k, r), mbe(k, r), k, NONE]
                o7 += [pb(k, n), ws, rs, pb(k, l), mbo(k, l), pb(k, r), mbo(k, r)]
            return e9, o7

        def check(what):
            by_part, total = np.zeros(K), [0.0]
            eng.calculateRootLogLikelihoodsByPartition([pb(k, tree.root) for k in range(K)], list(range(K)), list(range(K)), [T - 1] * K,
                                                       list(range(K)), K, 1, by_part, total)
            for k in range(K):
                out = [0.0]
                ora[k].calculateRootLogLikelihoods([pb(k, tree.root)], [0], [0], [T - 1], 1, out)
                assert helpers.rel_err(by_part[k], out[0]) <= 1e-10, (what, k, by_part[k], out[0])

        internal = [n for n in tree.postorder() if n >= T]
        e_all = []
        for k in range(K):                                       # full evaluation, rescaling in write mode
            matrices(k, branches)
            e9, o7 = ops_for(k, internal, True)
            e_all += e9
            ora[k].updatePartials(o7, len(o7) // 7, NONE)
            ora[k].resetScaleFactors(T - 1)
            ora[k].accumulateScaleFactors([n - T for n in internal], len(internal), T - 1)
        eng.kernelTimer(True)
        eng.updatePartialsByPartition(e_all, len(e_all) // 9)
        for k in range(K):
            eng.resetScaleFactorsByPartition(T - 1, k)
            eng.accumulateScaleFactorsByPartition([n - T for n in internal], len(internal), T - 1, k)
        check("full")
        for step in range(25):
            k = int(rng.integers(K))
            n0 = int(rng.choice(branches))
            saved = (pflip[k].copy(), mflip[k].copy(), lens[n0])
            lens[n0] *= float(np.exp(0.3 * rng.standard_normal()))
            mflip[k][n0] ^= 1
            matrices(k, [n0])
            path, a = set(), tree.parent[n0]
            while a >= 0:
                path.add(int(a)); a = tree.parent[a]
            order = [n for n in internal if n in path]
            for n in order:
                pflip[k][n] ^= 1
            e9, o7 = ops_for(k, order, False)
            eng.updatePartialsByPartition(e9, len(e9) // 9)
            ora[k].updatePartials(o7, len(o7) // 7, NONE)
            check("step %d" % step)
            if rng.random() < 0.35:                              # rejected: the unflipped buffers still hold their values
                pflip[k], mflip[k], lens[n0] = saved
                check("step %d restored" % step)
        stats = eng.walkStats()
        # unstored nodes exist; 4 states: every launch on the assembly loop; 20 states: the T32 walk for the read-mode updates
        # (the write-mode evaluation at the start ran level by level: its operations are counted as stored)
        assert stats["walks"] > 0 and stats["stored"] < stats["micro_ops"]
        assert S != 4 or stats["fast_walks"] == stats["walks"]
        for k, w in enumerate(wls):
            for n in internal:
                pe = eng.getPartials(pb(k, n), NONE)[:, starts[k]:starts[k + 1], :]
                po = ora[k].getPartials(pb(k, n), NONE)
                scale = np.maximum(np.abs(po).max(axis=(0, 2), keepdims=True), 1e-300)
                assert np.max(np.abs(pe - po) / scale) <= 1e-10, (k, n)
        check("after read-back")
    finally:
        eng.finalize()
        for o in ora:
            o.finalize()


def test_partition_roots_inside_the_walk_launch_give_the_bits_of_the_launch_of_their_own(monkeypatch):
    """calculateRootLogLikelihoodsByPartition right behind updatePartialsByPartition on a 4-state instance: the top slice of every
    partition integrates its own root inside the walk's launch (kernels.h RootFusedParts) — the same functions in the same order as
    k_rootSite4WParts, the launch of its own that BEAGLE_MI355_NO_ROOT_PARTS_FUSION=1 keeps: per-partition sums, total and site values
    bit for bit, through flipped buffers, changing branch rates and node-height moves (whose lists are the partitions' paths to their roots)."""
    from beast_mcmc_amd.inputs import synth
    from beast_mcmc_amd.multipartition import MultiPartitionTreeLikelihood
    pw = synth.config_e(scale=0.08)

    heights = np.array(pw.tree.height, dtype=float).copy()

    def run():
        pw.tree.height[:] = heights                          # (the moves below change the shared tree)
        tl = MultiPartitionTreeLikelihood(pw)
        out = []
        for i in range(4):
            tl.set_branch_rates(np.ones(pw.tree.node_count) * (1.0 + 0.01 * i))
            by_part, total = tl.calculate()
            out.append((by_part.copy(), total, tl.getSiteLogLikelihoods().copy()))
        rng = np.random.default_rng(3)
        tree = pw.tree
        for _ in range(6):
            node = int(rng.integers(tree.tip_count, tree.node_count))
            if node == tree.root:
                continue
            lo = max(tree.height[int(tree.left[node])], tree.height[int(tree.right[node])]); hi = tree.height[tree.parent[node]]
            by_part, total = tl.move_node_height(node, lo + 0.4 * (hi - lo))
            out.append((by_part.copy(), total, tl.getSiteLogLikelihoods().copy()))
        by_part, total = tl.calculate()
        out.append((by_part.copy(), total, tl.getSiteLogLikelihoods().copy()))
        info = tl.b.walkLaunchInfo()
        tl.close()
        return out, info

    fused, info = run()
    assert info["partition_roots_in_walk"] >= 5, info
    monkeypatch.setenv("BEAGLE_MI355_NO_ROOT_PARTS_FUSION", "1")
    plain, info1 = run()
    assert info1["partition_roots_in_walk"] == 0
    for (a, ta, sa), (b, tb, sb) in zip(fused, plain):
        assert ta == tb and np.array_equal(a, b) and np.array_equal(sa, sb)


@pytest.mark.parametrize("always_rescale", [False, True])
def test_native_call_sequence_is_the_python_one(always_rescale):
    """calculate() issues its calls from C++ (tools/host mpEvaluate) by default and call by call from Python with
    native_sequence=False: the same calls in the same order, so the same bits — through rate changes, a model change of one partition
    (flags) and evaluations in between node-height moves (which are issued from Python either way)."""
    from beast_mcmc_amd.inputs import synth
    from beast_mcmc_amd.multipartition import MultiPartitionTreeLikelihood
    pw = synth.config_e(scale=0.06)
    heights = np.array(pw.tree.height, dtype=float).copy()
    kappa0 = pw.parts[1].eig

    def run(native):
        pw.tree.height[:] = heights
        pw.parts[1].eig = kappa0
        tl = MultiPartitionTreeLikelihood(pw, always_rescale=always_rescale, native_sequence=native)
        out = []
        for i in range(3):
            tl.set_branch_rates(np.ones(pw.tree.node_count) * (1.0 + 0.02 * i))
            by_part, total = tl.calculate()
            out.append((by_part.copy(), total))
        tl.set_substitution_model(1, eig=substmodel.hky(3.3, pw.parts[1].freqs))
        out.append(tuple(x if np.isscalar(x) else x.copy() for x in tl.calculate()))
        if not always_rescale:
            tl.move_node_height(pw.tree.tip_count + 3, float(0.5 * (pw.tree.height[pw.tree.tip_count + 3] + pw.tree.height[pw.tree.parent[pw.tree.tip_count + 3]]))) if pw.tree.parent[pw.tree.tip_count + 3] >= 0 else None
            out.append(tuple(x if np.isscalar(x) else x.copy() for x in tl.calculate()))
        assert (tl._fast["native"] is not None) == native
        tl.close()
        return out

    a, b = run(True), run(False)
    assert len(a) == len(b)
    for (pa, ta), (pb, tb) in zip(a, b):
        assert ta == tb and np.array_equal(pa, pb)
