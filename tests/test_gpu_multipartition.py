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


@pytest.mark.parametrize("S", [4, 20])
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
