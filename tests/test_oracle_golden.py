"""Pin the CPU oracle (oracle/beagle_cpu_oracle.c), the input front-end and the C++ host driver against
every golden vector the reference's own tests hold for the tree-likelihood path.
No GPU needed: these run the oracle through the SAME binding and host driver the HIP engine uses.
"""
import numpy as np
import pytest

import beast_mcmc_amd as bm
import helpers
import reference_quantile
from beast_mcmc_amd.inputs import patterns, siterates, substmodel, trees
from beast_mcmc_amd.treelikelihood import (BeagleTreeLikelihood, POST_ORDER, RESCALE_ALWAYS, RESCALE_DYNAMIC,
                                           RESCALE_NONE, REVERSE_LEVEL_ORDER)

PRIMATES = helpers.golden("primates.json")


def fmt5(x):
    """NumberFormat with 5 fraction digits (TreeDataLikelihoodTest.java:79) — HALF_EVEN on the decimal expansion."""
    return "%.5f" % x


@pytest.mark.parametrize("case", PRIMATES["tree_data_likelihood_test"], ids=lambda c: c["name"])
def test_tree_data_likelihood_test_values(case, oracle_lib):
    """src/test/dr/evomodel/treedatalikelihood/TreeDataLikelihoodTest.java:116-315 (10 PAUP* values)."""
    wl = helpers.primates_case(case, site_model="new")
    tl = BeagleTreeLikelihood(wl, library=oracle_lib, rescaling=RESCALE_DYNAMIC, traversal=REVERSE_LEVEL_ORDER)
    assert fmt5(tl.getLogLikelihood()) == fmt5(case["lnL"])
    tl.close()


@pytest.mark.parametrize("case", PRIMATES["likelihood_test"], ids=lambda c: c["name"])
def test_likelihood_test_values(case, oracle_lib):
    """src/test/dr/evomodel/treelikelihood/LikelihoodTest.java:86-345 (12 PAUP* values, older site model)."""
    wl = helpers.primates_case(case, site_model="old")
    tl = BeagleTreeLikelihood(wl, library=oracle_lib, rescaling=RESCALE_NONE, traversal=POST_ORDER)
    assert fmt5(tl.getLogLikelihood()) == fmt5(case["lnL"])
    tl.close()


def test_primates_pattern_compression():
    rows = np.stack([patterns.nucleotide_states(s) for s in PRIMATES["sequences"]])
    pats, w = patterns.site_patterns(rows)
    assert rows.shape == (6, 768)
    assert w.sum() == 768
    assert pats.shape[1] == len(w) < 768
    # newick of the fixture tree (TreeDataLikelihoodTest.testNewickTree, 6 fraction digits)
    tree = trees.from_nested(helpers._tuplify(PRIMATES["tree_nested"]), 6)
    assert abs(tree.branch_length(0) - 0.024003) < 1e-12
    assert abs(tree.branch_length(tree.parent[1]) - 0.013231) < 1e-12


def run_jar_smoke(library):
    g = helpers.golden("jar_smoke.json")
    n_sites = len(g["sequences"][0])
    inst = g["instance"]
    b = bm.beagle.Beagle(inst["tipCount"], inst["partialsBufferCount"], inst["compactBufferCount"], inst["stateCount"],
                         n_sites, inst["eigenBufferCount"], inst["matrixBufferCount"], inst["categoryCount"],
                         inst["scaleBufferCount"], library=library)
    try:
        for t, seq in enumerate(g["sequences"]):
            st = patterns.nucleotide_states(seq).copy()
            st[st > 3] = 4                                   # BeagleFactory#getStates: anything else -> 4
            b.setTipStates(t, st)
        b.setPatternWeights(np.ones(n_sites))
        b.setStateFrequencies(0, g["freqs"])
        b.setCategoryWeights(0, g["weights"])
        b.setCategoryRates(g["rates"])
        b.setEigenDecomposition(0, g["evec"], g["ivec"], g["eval"])
        b.updateTransitionMatrices(0, g["matrix_indices"], None, None, g["edge_lengths"], 4)
        b.updatePartials(g["operations"], 2, bm.beagle.NONE)
        out = [0.0]
        b.calculateRootLogLikelihoods([g["root"]], [0], [0], [bm.beagle.NONE], 1, out)
        return out[0], g
    finally:
        b.finalize()


def test_beagle_jar_smoke_value(oracle_lib):
    """lib/beagle.jar!beagle/BeagleFactory#main: 'PAUP logL = -1574.63623'."""
    lnl, g = run_jar_smoke(oracle_lib)
    assert fmt5(lnl) == fmt5(g["lnL"])


def run_branch_specific(library, stem_weight):
    """tests/TestXML/testBranchSpecificSubstitutionModel.xml: GTR2 on the branches leading to C and D
    (stem weight 0), plus the stem branch above (C,D) (stem weight 1); GTR1 elsewhere and at the root."""
    g = helpers.golden("branch_specific.json")
    rows = np.stack([patterns.nucleotide_states(s) for s in g["sequences"]])
    tree = trees.from_nested(helpers._tuplify(g["tree_nested"]), 4)
    tree.height[:4] = g["tip_heights"]
    n_sites = rows.shape[1]
    e1 = substmodel.gtr(g["gtr1"]["rates"], g["gtr1"]["pi"])
    e2 = substmodel.gtr(g["gtr2"]["rates"], g["gtr2"]["pi"])
    rates, props = siterates.GammaSiteRateModel(alpha=g["alpha"], gamma_categories=g["cats"],
                                                quantile=reference_quantile.gamma_quantile).category_rates_and_proportions()
    # patterns are NOT compressed here (<patterns strip="false"> over 14 sites, weights 1)
    b = bm.beagle.Beagle(4, 4 + 3, 4, 4, n_sites, 2, 7, len(rates), 0, library=library)
    try:
        for t in range(4):
            b.setTipStates(t, rows[t])
        b.setPatternWeights(np.ones(n_sites))
        b.setEigenDecomposition(0, e1.evec, e1.ievc, e1.evals)
        b.setEigenDecomposition(1, e2.evec, e2.ievc, e2.evals)
        b.setCategoryRates(rates)
        b.setCategoryWeights(0, props)
        b.setStateFrequencies(0, g["gtr1"]["pi"])
        cd = tree.parent[2]
        assert cd == tree.parent[3]
        clade_branches = [2, 3] + ([int(cd)] if stem_weight == 1.0 else [])
        other = [n for n in range(tree.node_count) if n != tree.root and n not in clade_branches]
        bl = lambda n: tree.branch_length(n, g["clock_rate"])
        b.updateTransitionMatrices(0, other, None, None, [bl(n) for n in other], len(other))
        b.updateTransitionMatrices(1, clade_branches, None, None, [bl(n) for n in clade_branches], len(clade_branches))
        ops = []
        for n in tree.postorder():
            if n >= 4:
                l, r = int(tree.left[n]), int(tree.right[n])
                ops += [n, -1, -1, l, l, r, r]
        b.updatePartials(ops, 3, bm.beagle.NONE)
        out = [0.0]
        b.calculateRootLogLikelihoods([tree.root], [0], [0], [bm.beagle.NONE], 1, out)
        return out[0], g
    finally:
        b.finalize()


def run_transition_probabilities(library):
    """tests/golden/transition_probabilities.json through setEigenDecomposition -> updateTransitionMatrices -> getTransitionMatrix:
    -> [(case, P(t) of category 0 from a branch of length t, P(t) of category 1 — rate 1/2 — from a branch of length 2 t)]"""
    g = helpers.golden("transition_probabilities.json")
    n = len(g["cases"])
    b = bm.beagle.Beagle(2, 3, 2, 4, 1, 1, 2 * n, 2, 0, library=library)
    try:
        b.setCategoryRates([1.0, 0.5])
        out = []
        for k, c in enumerate(g["cases"]):
            q = substmodel.reversible_q(c["relative_rates_ac_ag_at_cg_ct_gt"], c["pi"])
            e = substmodel.decompose_reversible(q, c["pi"])
            b.setEigenDecomposition(0, e.evec, e.ievc, e.evals)
            b.updateTransitionMatrices(0, [2 * k, 2 * k + 1], None, None, [c["distance"], 2.0 * c["distance"]], 2)
            out.append((c, b.getTransitionMatrix(2 * k)[0].reshape(-1), b.getTransitionMatrix(2 * k + 1)[1].reshape(-1)))
        return out
    finally:
        b.finalize()


def check_transition_probabilities(rows):
    assert len(rows) == 10
    for c, p_t, p_half_rate in rows:
        want = np.asarray(c["expected_row_major"])
        assert np.abs(p_t - want).max() <= c["tolerance"], (c["source"], np.abs(p_t - want).max())
        assert np.abs(p_half_rate - want).max() <= c["tolerance"], (c["source"], np.abs(p_half_rate - want).max())
        np.testing.assert_allclose(p_t.reshape(4, 4).sum(axis=1), 1.0, rtol=0, atol=1e-14)


def test_transition_probability_known_answers(oracle_lib):
    """HKYTest.java:151-156, TN93Test.java:167-172, GeneralF81Test.java:158-163: the ten scilab expm matrices, entry by entry to 1e-10
    (row a2 / a3 pinned on its own, not only through a tree's log-likelihood)."""
    check_transition_probabilities(run_transition_probabilities(oracle_lib))


@pytest.mark.parametrize("case", [0, 1])
def test_branch_specific_full_precision(case, oracle_lib):
    """The only full-precision lnL values the reference pins on this path (tolerance 1e-13, absolute)."""
    g = helpers.golden("branch_specific.json")
    c = g["cases"][case]
    lnl, _ = run_branch_specific(oracle_lib, c["stem_weight"])
    # the XML's tolerance is 1e-13; the eigen solver here is numpy's, not Colt's, so allow a few ulps more
    assert abs(lnl - c["lnL"]) < 5e-13, (lnl, c["lnL"])


def run_epoch_convolution(library):
    """tests/TestXML/testEpochConvolutionOrder.xml: a branch that spans several epochs gets the PRODUCT of the per-epoch
    transition matrices, root end first, built with convolveTransitionMatrices
    (src/dr/evomodel/treelikelihood/SubstitutionModelDelegate.java:271-405)."""
    g = helpers.golden("epoch_convolution.json")
    pi = [0.5, 0.5]
    b = bm.beagle.Beagle(2, 3, 2, 2, 1, 4, 24, 1, 0, library=library)
    try:
        for k, r in enumerate(g["epoch_rates"]):
            e = substmodel.decompose_general(substmodel.asymmetric_q(r, pi), pi)
            b.setEigenDecomposition(k, e.evec, e.ievc, e.evals)
        b.setCategoryRates([1.0]); b.setCategoryWeights(0, [1.0]); b.setStateFrequencies(0, g["root_freqs"])
        b.setPatternWeights([1.0])
        bounds = [0.0] + g["transition_times"] + [float("inf")]
        nxt = [4]                                   # next free matrix buffer (0..1 are the final branch matrices)

        def branch_matrix(tip, final_idx):
            lo, hi = g["tip_heights"][tip], g["root_height"]
            pieces = []
            for k in reversed(range(4)):            # root end (oldest epoch) first
                a, z = max(lo, bounds[k]), min(hi, bounds[k + 1])
                if z > a:
                    pieces.append((k, (z - a) * g["clock_rate"]))
            idx = []
            for k, t in pieces:
                b.updateTransitionMatrices(k, [nxt[0]], None, None, [t], 1)
                idx.append(nxt[0]); nxt[0] += 1
            acc = idx[0]
            for j, m in enumerate(idx[1:]):
                res = final_idx if j == len(idx) - 2 else nxt[0]
                if res != final_idx:
                    nxt[0] += 1
                b.convolveTransitionMatrices([acc], [m], [res], 1)
                acc = res
            return acc

        mx, my = branch_matrix(0, 0), branch_matrix(1, 1)
        b.setTipStates(0, [g["states"][0]]); b.setTipStates(1, [g["states"][1]])
        b.updatePartials([2, -1, -1, 0, mx, 1, my], 1, bm.beagle.NONE)
        out = [0.0]
        b.calculateRootLogLikelihoods([2], [0], [0], [bm.beagle.NONE], 1, out)
        return out[0], g
    finally:
        b.finalize()


def test_epoch_convolution_value(oracle_lib):
    lnl, g = run_epoch_convolution(oracle_lib)
    assert abs(lnl - g["lnL"]) < 1e-5, lnl          # the XML's tolerance is 1e-3; the value is known to 5 dp


def test_gamma_rates_match_reference_discretisation():
    """GammaSiteRateModel.java:445-472 with alpha = 0.5, 4 categories: AS 91 quantiles, mean-normalised."""
    rates, props = siterates.GammaSiteRateModel(alpha=0.5, gamma_categories=4,
                                                quantile=reference_quantile.gamma_quantile).category_rates_and_proportions()
    assert abs(sum(rates) / 4 - 1.0) < 1e-15
    assert props == [0.25] * 4
    from scipy import stats
    exact = stats.gamma.ppf([(2 * i + 1) / 8.0 for i in range(4)], 0.5, scale=2.0)
    exact = exact / exact.mean()
    # the reference's quantile is within ~1e-6 of the exact one, and NOT equal to it
    assert np.allclose(rates, exact, rtol=1e-5)
    assert not np.allclose(rates, exact, rtol=1e-12)
    # +I only: literal 2.0 for the variable class (GammaSiteRateModel.java:254)
    r, p = siterates.GammaSiteRateModel(p_inv=0.75).category_rates_and_proportions()
    assert r == [0.0, 2.0] and p == [0.75, 0.25]


# ---- pre-order partials and branch gradients ------------------------------------------------
# The reference holds no expected numbers for these natives (its own tests compare the analytic gradient with a
# numerical one, tests/TestXML/test*Gradient*.xml), so the oracle is pinned the same way: against central finite
# differences of the log-likelihood that the golden values above DO pin.

@pytest.mark.parametrize("S,C,T,P,seed", [(4, 4, 7, 60, 3), (4, 1, 12, 40, 5), (20, 2, 6, 30, 7), (61, 1, 5, 12, 9)])
def test_oracle_branch_gradient_matches_finite_differences(S, C, T, P, seed):
    from beast_mcmc_amd.gradient import BranchGradient
    wl = helpers.random_workload(T, P, S, C, seed)
    g = BranchGradient(wl, library=helpers.oracle_library())
    lnl, grad, hess = g.gradient(second=True)
    assert helpers.rel_err(lnl, g.log_likelihood()) < 1e-14
    for n in g.edges:
        t = g.branch_lengths[n]
        h = 1e-4 * t                       # short branches have large third derivatives: scale the step with t
        g.set_branch_length(n, t + h); up = g.log_likelihood()
        g.set_branch_length(n, t - h); dn = g.log_likelihood()
        g.set_branch_length(n, t); mid = g.log_likelihood()
        fd1 = (up - dn) / (2 * h)
        fd2 = (up - 2 * mid + dn) / (h * h)
        noise = 50 * np.finfo(float).eps * abs(lnl) / h       # rounding of lnL itself, amplified by 1/h (and 1/h^2)
        assert abs(fd1 - grad[n]) <= 1e-5 * max(1.0, abs(grad[n])) + noise, (n, fd1, grad[n])
        assert abs(fd2 - hess[n]) <= 1e-2 * max(1.0, abs(hess[n])) + 4 * noise / h, (n, fd2, hess[n])
    g.close()


def test_oracle_pre_order_partials_invariant():
    """sum_j pre[n][j] post[n][j], integrated over categories, is the site likelihood at EVERY node."""
    from beast_mcmc_amd.gradient import BranchGradient
    wl = helpers.random_workload(9, 50, 4, 3, 11)
    g = BranchGradient(wl, library=helpers.oracle_library())
    lnl, grad, per = g.gradient(per_pattern=True)
    site = None
    for n in range(g.T, g.N):
        pre = g.pre_partials(n).reshape(g.C, g.P, g.S)
        post = g.post_partials(n).reshape(g.C, g.P, g.S)
        like = np.einsum("c,cpi,cpi->p", wl.cat_weights, pre, post)
        if site is None:
            site = like
        assert np.max(np.abs(like - site) / site) < 1e-12
    assert helpers.rel_err(float(np.sum(wl.weights * np.log(site))), lnl) < 1e-13
    # per-pattern derivatives sum (weighted) to the per-edge totals
    assert np.allclose(per @ wl.weights, grad[g.edges], rtol=1e-12, atol=1e-12)
    # transposeTransitionMatrices round trip
    m = g.b.getTransitionMatrix(0).reshape(g.C, g.S, g.S)
    g.b.transposeTransitionMatrices([0], [g.q2_index], 1)
    mt = g.b.getTransitionMatrix(g.q2_index).reshape(g.C, g.S, g.S)
    assert np.array_equal(mt, np.transpose(m, (0, 2, 1)))
    g.close()


@pytest.mark.parametrize("S,C", [(4, 4), (20, 2), (61, 1)])
def test_oracle_cross_products_scaling_identity(S, C):
    """calculateCrossProductDifferentials is inferred (header of the oracle function).  The one direction in which the
    first-order form is exact pins its structure — category rates, branch lengths, pattern weights, the per-pattern
    denominator: scaling Q by (1 + a) is scaling every branch, so sum_ij Q_ij dlnL/dQ_ij = sum_b t_b dlnL/dt_b."""
    from beast_mcmc_amd.gradient import BranchGradient
    wl = helpers.random_workload(8, 45, S, C, seed=31 + S)
    g = BranchGradient(wl, library=helpers.oracle_library())
    lnl, grad = g.gradient()
    cp = g.cross_products()
    q = (wl.eig.evec * wl.eig.evals[None, :]) @ wl.eig.ievc
    assert helpers.rel_err(float(np.sum(q * cp)), float(np.dot(g.branch_lengths, grad))) < 1e-12
    # accumulates into the caller's array, as BEAST's zero-filled buffer expects
    post = np.asarray(g.edges, dtype=np.int32)
    acc = np.ones(S * S)
    g.b.calculateCrossProductDifferentials(post, post + g.pre_offset, [0], [0], g.branch_lengths[post], len(post), out=acc)
    assert np.allclose(acc - 1.0, cp.ravel(), rtol=1e-12, atol=1e-12)
    g.close()


@pytest.mark.parametrize("S,C", [(4, 2), (20, 1)])
def test_oracle_cross_products_are_the_first_order_generator_derivative(S, C):
    """Pins calculateCrossProductDifferentials beyond the scaling direction.  Its consumer asks for "a first-order"
    approximation of d lnL / d Q_ij (AbstractLogAdditiveSubstitutionModelGradient.java:74-93, ApproximationMode.FIRST_ORDER;
    AFFINE_CORRECTED adds a term computed in Java from the SAME cross products, :95-130), i.e. the exact derivative
    sum_b int_0^t_b pre_b(s)_i post_b(s)_j ds / L with the integrand frozen at the branch's lower end: exact up to O(t_b) per
    branch.  Check: central finite differences of the golden-pinned lnL along 10 random generator perturbations dQ — the
    perturbed lnL is evaluated through setTransitionMatrix with P = expm((Q + h dQ) t r_c), no eigen system involved — against
    sum_ij dQ_ij out_ij, on the same tree with all branch lengths scaled by 1, 1/2, ... 1/16: the relative error must fall in
    proportion to the branch lengths (first order: it halves each time) and be below 2 % at the shortest.  A transposed,
    mis-weighted or mis-indexed formula has an error that does not vanish."""
    from scipy.linalg import expm
    from beast_mcmc_amd.gradient import BranchGradient
    wl = helpers.random_workload(7, 40, S, C, seed=61 + S)
    rng = np.random.default_rng(5)
    dqs = [rng.normal(size=(S, S)) for _ in range(10)]
    e = wl.eig
    q = (e.evec * e.evals[None, :]) @ e.ievc
    errs = []
    for scale in (1.0, 0.5, 0.25, 0.125, 0.0625):
        g = BranchGradient(wl, library=helpers.oracle_library())
        g.branch_lengths = g.branch_lengths * scale
        lnl, _ = g.gradient()
        cp = g.cross_products()

        def lnl_of(qm):
            for n in g.edges:
                p = np.stack([expm(qm * g.branch_lengths[n] * r) for r in wl.cat_rates])
                g.b.setTransitionMatrix(n, p.ravel(), 0.0)
            g.b.updatePartials(g._post_ops, len(g._post_ops) // 7, bm.beagle.NONE)
            out = [0.0]
            g.b.calculateRootLogLikelihoods([g.tree.root], [0], [0], [bm.beagle.NONE], 1, out)
            return out[0]

        assert helpers.rel_err(lnl_of(q), lnl) < 1e-10          # expm(Q t) = the eigen-system matrices
        h = 1e-6
        fd = np.array([(lnl_of(q + h * d) - lnl_of(q - h * d)) / (2 * h) for d in dqs])
        ap = np.array([float(np.sum(d * cp)) for d in dqs])
        errs.append(float(np.linalg.norm(ap - fd) / np.linalg.norm(fd)))
        g.close()
    ratios = [errs[i] / errs[i + 1] for i in range(len(errs) - 1)]
    assert errs[-1] < 0.02, errs
    assert all(1.5 < r < 2.7 for r in ratios[1:]), (errs, ratios)


@pytest.mark.parametrize("name,taxa,sites,npatterns", [("benchmark1", 1441, 987, 593), ("benchmark2", 62, 10869, 5565)])
def test_real_benchmark_alignments_as_fixtures(name, taxa, sites, npatterns, oracle_lib):
    """examples/Benchmarks/benchmark{1,2}.xml transcribed as unique site patterns (make_fixtures.py --benchmarks): the pattern
    count equals the XML's own "npatterns=" comment (benchmark1.xml:10108, benchmark2.xml:641), the weights add up to the
    site count, and the oracle evaluates them (real ambiguity codes, real pattern weights).  No expected lnL is published
    for them — the XMLs draw a random starting tree — so what these fixtures pin is engine <-> oracle (tests/test_gpu_configs.py)."""
    import os
    from beast_mcmc_amd.inputs import synth
    wl = synth.from_pattern_fixture(os.path.join(helpers.GOLDEN, name + "_patterns.npz"))
    assert wl.tip_count == taxa and wl.pattern_count == npatterns and wl.weights.sum() == sites
    assert (wl.tip_states == 4).sum() > 0
    tl = BeagleTreeLikelihood(wl, library=oracle_lib, rescaling=RESCALE_DYNAMIC)
    lnl = tl.getLogLikelihood()
    assert np.isfinite(lnl) and lnl < 0
    site = tl.getSiteLogLikelihoods()
    assert abs(float((site * wl.weights).sum()) - lnl) <= 1e-9 * abs(lnl)
    tl.close()


def _cyclic_model(S, seed):
    """An asymmetric rate matrix with a strong directed cycle: its eigenvalues come in complex-conjugate pairs."""
    rng = np.random.default_rng(seed)
    q = substmodel.complex_q(rng.uniform(0.02, 0.1, size=S * (S - 1)), S)
    for i in range(S):
        q[i, (i + 1) % S] += 2.0
    np.fill_diagonal(q, 0.0); np.fill_diagonal(q, -q.sum(axis=1))
    pi, eig = substmodel.decompose_complex(q)
    assert np.count_nonzero(eig.evals[S:]) >= 2                       # at least one conjugate pair
    return q / float(-(np.diag(q) * pi).sum()), pi, eig


@pytest.mark.parametrize("S", [4, 7, 20])
def test_complex_eigen_systems_against_the_matrix_exponential(S, oracle_lib):
    """EIGEN_COMPLEX (BeagleTreeLikelihood.java:353-355: required whenever the model can return a complex
    diagonalisation — ComplexSubstitutionModel.java:232, the asymmetric discrete-trait models): the oracle's restatement of
    ComplexSubstitutionModel.java:121-187 (real block form, 2 S eigenvalue entries) reproduces exp(Q t r_c) computed
    independently (scipy.linalg.expm) for every rate category — this pins the complex path, for which the reference holds
    no golden value."""
    import scipy.linalg
    qn, pi, eig = _cyclic_model(S, 3 + S)
    b = bm.beagle.Beagle(3, 5, 3, S, 10, 1, 4, 2, 0, requirementFlags=bm.beagle.FLAG_EIGEN_COMPLEX, library=oracle_lib)
    assert b.details.flags & bm.beagle.FLAG_EIGEN_COMPLEX
    b.setEigenDecomposition(0, eig.evec, eig.ievc, eig.evals)
    b.setCategoryRates([0.5, 1.7])
    b.updateTransitionMatrices(0, [0, 1], None, None, [0.3, 1.1], 2)
    for m, t in ((0, 0.3), (1, 1.1)):
        got = b.getTransitionMatrix(m).reshape(2, S, S)
        for c, r in enumerate((0.5, 1.7)):
            assert np.max(np.abs(got[c] - scipy.linalg.expm(qn * t * r))) <= 1e-13
    b.finalize()


def test_precise_mode_is_the_same_algorithm_in_wider_arithmetic(oracle_lib):
    """oracle_set_precise(1) — long double sums, rounded once per entry: the third evaluation tests/test_gpu_configs.py arbitrates with —
    must reproduce the pinned 4-state value as well as the default path does, sit within the fp64 spread of it at 20 states, and leave the
    default path untouched when it is switched off again."""
    import beast_mcmc_amd as bm
    from beast_mcmc_amd.treelikelihood import BeagleTreeLikelihood, RESCALE_DYNAMIC
    wl4 = helpers.random_workload(40, 300, 4, 4, seed=12)
    wl20 = helpers.random_workload(30, 120, 20, 4, seed=13)
    out = {}
    for name, wl in (("4", wl4), ("20", wl20)):
        vals = []
        for precise in (0, 1, 0):
            oracle_lib.lib.oracle_set_precise(precise)
            try:
                t = BeagleTreeLikelihood(wl, library=oracle_lib, rescaling=RESCALE_DYNAMIC, delay_rescaling=False)
                vals.append((t.getLogLikelihood(), t.getSiteLogLikelihoods().copy()))
                t.close()
            finally:
                oracle_lib.lib.oracle_set_precise(0)
        out[name] = vals
        assert vals[0][0] == vals[2][0] and np.array_equal(vals[0][1], vals[2][1])                # switched off: the pinned path, bit for bit
        assert helpers.rel_err(vals[1][0], vals[0][0]) <= 1e-12
        assert np.max(np.abs(vals[1][1] - vals[0][1]) / np.abs(vals[0][1])) <= 1e-10
    assert oracle_lib.lib.oracle_precise() == 0
