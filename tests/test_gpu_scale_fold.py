"""Read mode, 4 states: the reciprocal scale factors of unstored nodes are applied once, at the stored node above them.

A node that is not stored is seen by nobody but its parent, and a partial is linear in each of its children — so a cached
full-evaluation program multiplies by ONE vector (the entry-wise product of the unstored nodes' reciprocals: a "fold",
engine_internal.h Instance::folds) where it used to multiply node by node.  The per-node buffers stay what
getLogScaleFactors / accumulateScaleFactors / partial updates / the gradient pass expect.  Held here:
  * folded against BEAGLE_MI355_NO_SCALE_FOLD=1 (per-node factors everywhere): lnL, site values and every node's partials agree
    to rounding, both agree with the oracle to 1e-10, and the program reads a fraction of the scale vectors;
  * folds follow the factors: a chain under DYNAMIC rescaling that recomputes its factors every few evaluations while the model
    changes (every write-mode evaluation makes the folds stale; they are rebuilt before the next read-mode one);
  * both walk kernels run the folded program to the same bits;
  * a fold whose products leave the safe range is refused: the plan falls back to per-node factors — the same BITS as with folding off."""
import os

import numpy as np
import pytest

import beast_mcmc_amd as bm
import helpers
from beast_mcmc_amd.inputs import substmodel, trees
from beast_mcmc_amd.inputs.synth import Workload
from beast_mcmc_amd.treelikelihood import BeagleTreeLikelihood, RESCALE_DYNAMIC

pytestmark = pytest.mark.gpu


class env:
    def __init__(self, **kv):
        self.kv = kv

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kv}
        for k, v in self.kv.items():
            os.environ[k] = v

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def make(wl, fold=True, fast=True):
    # (BEAGLE_MI355_NO_SLICE_SUMS=1: the first — write-mode — evaluation's cumulative buffer from the per-node factors, the same bits on both
    # kernels; the per-slice products of round 6 depend on the slicing, which the kernels choose differently: tests/test_gpu_slice_sums.py)
    with env(BEAGLE_MI355_NO_SCALE_FOLD="0" if fold else "1", BEAGLE_MI355_NO_FAST_WALK="0" if fast else "1", BEAGLE_MI355_NO_SLICE_SUMS="1"):
        return BeagleTreeLikelihood(wl, rescaling=RESCALE_DYNAMIC, delay_rescaling=False)


def steady_state(wl, fold, fast=True, nodes=()):
    tl = make(wl, fold, fast)
    raw = bm.beagle.Beagle.attach(tl)
    first = tl.getLogLikelihood()              # write mode
    raw.kernelTimer(True)
    vals = []
    for _ in range(3):                         # read mode, both buffer-flip states
        tl.makeDirty()
        vals.append(tl.getLogLikelihood())
    stats = raw.walkStats()
    raw.kernelTimer(False)
    site = tl.getSiteLogLikelihoods().copy()
    parts = [raw.getPartials(tl.node_buffer_index(n), tl.node_scale_index(n)).copy() for n in nodes]
    factors = [raw.getLogScaleFactors(tl.node_scale_index(n)).copy() for n in nodes]
    health = raw.walkHealth()
    tl.close()
    return first, vals, site, parts, factors, stats, health


@pytest.mark.parametrize("T,P,C,kind,S", [(260, 2100, 4, "coalescent", 4), (120, 333, 1, "yule", 4), (90, 1500, 8, "caterpillar", 4), (400, 9000, 4, "coalescent", 4),
                                          (70, 900, 4, "coalescent", 20), (45, 333, 2, "yule", 17)])      # (16..20 states: the same programs on k_walkT32, which divides by the factors)
def test_folded_and_per_node_factors_agree(T, P, C, kind, S, oracle_lib):
    wl = helpers.random_workload(T, P, S, C, seed=8200 + T, tree_kind=kind)
    nodes = list(range(wl.tree.tip_count, wl.tree.node_count))
    f0, fv, fs, fp, ff, fstats, fh = steady_state(wl, True, nodes=nodes)
    u0, uv, us, up, uf, ustats, uh = steady_state(wl, False, nodes=nodes)
    assert f0 == u0                                            # the write-mode evaluation is the same program
    assert fh["folded_vectors"] > 0 and fh["fold_builds"] > 0 and uh["folded_vectors"] == 0 and uh["fold_builds"] == 0
    assert ustats["scale_reads"] == 3 * (T - 1)                # one vector per node and evaluation ...
    # ... against one per stored node (and per 32 unstored in a row); a ladder stores most of its nodes (only the definition at its foot is unstored)
    # (... and an alignment whose partials buffer is under 64 KiB keeps definitions of two steps only since round 6 — many stored nodes, each paying)
    tiny = S == 4 and P * C * 32 < (64 << 10)
    assert fstats["scale_reads"] * (1 if kind == "caterpillar" or tiny else 2) < ustats["scale_reads"]
    assert fstats["stored"] == ustats["stored"] and fstats["micro_ops"] == ustats["micro_ops"]
    for a, b in zip(fv, uv):
        assert helpers.rel_err(a, b) <= 1e-13
    assert np.max(np.abs(fs - us) / np.maximum(np.abs(us), 1e-300)) <= 1e-12
    for a, b in zip(ff, uf):                                   # the per-node factors are untouched
        assert np.array_equal(a, b)
    for a, b in zip(fp, up):                                   # every node (unstored ones materialised): same values to rounding
        assert np.allclose(a, b, rtol=1e-12, atol=1e-300)
    o = BeagleTreeLikelihood(wl, library=oracle_lib, rescaling=RESCALE_DYNAMIC, delay_rescaling=False)
    ref = o.getLogLikelihood()
    so = o.getSiteLogLikelihoods()
    o.close()
    assert helpers.rel_err(fv[-1], ref) <= 1e-10
    assert np.max(np.abs(fs - so) / np.maximum(np.abs(so), 1e-300)) <= 1e-10


def test_both_walk_kernels_run_the_folded_program_to_the_same_bits():
    wl = helpers.random_workload(180, 1700, 4, 4, seed=8311, tree_kind="coalescent")
    a = steady_state(wl, True, fast=True)
    b = steady_state(wl, True, fast=False)
    assert a[0] == b[0] and a[1] == b[1] and np.array_equal(a[2], b[2])
    assert a[6]["folded_vectors"] == b[6]["folded_vectors"] > 0


def test_folds_follow_the_factors_through_a_chain(oracle_lib):
    """DYNAMIC rescaling with beagle.rescale = 3: factors are recomputed every few evaluations while kappa and the tree change;
    branch moves (partial updates: per-node factors) and rejections in between."""
    wl = helpers.random_workload(150, 1300, 4, 4, seed=8421, tree_kind="coalescent")
    tls = [make(wl, True), make(wl, False), BeagleTreeLikelihood(wl, library=oracle_lib, rescaling=RESCALE_DYNAMIC, delay_rescaling=False)]
    for tl in tls:
        tl.set_rescaling_frequency(3)
    rng = np.random.default_rng(17)
    builds = []
    for step in range(24):
        kind = step % 4
        if kind == 0:                         # a model move: everything dirty
            eig = substmodel.gtr(rng.gamma(2.0, 1.0, size=6) + 0.1, wl.freqs)
            for tl in tls:
                tl.storeState()
                tl.set_substitution_model(eig, wl.freqs)
        elif kind == 1:
            for tl in tls:
                tl.makeDirty()
        else:                                 # a height move
            node = int(rng.integers(wl.tree.tip_count, wl.tree.node_count - 1))
            h = helpers.proposed_height(wl.tree, node, rng)
            for tl in tls:
                tl.storeState()
                tl.set_node_height(node, h)
        v = [tl.getLogLikelihood() for tl in tls]
        assert helpers.rel_err(v[0], v[1]) <= 1e-13, step
        assert helpers.rel_err(v[0], v[2]) <= 1e-10, step
        if kind == 3 and rng.random() < 0.5:
            for tl in tls:
                tl.restoreState()
            v = [tl.getLogLikelihood() for tl in tls]
            assert helpers.rel_err(v[0], v[1]) <= 1e-13 and helpers.rel_err(v[0], v[2]) <= 1e-10
        builds.append(bm.beagle.Beagle.attach(tls[0]).walkHealth()["fold_builds"])
    assert builds[-1] > 0 and len(set(builds)) >= 3         # rebuilt after later write-mode evaluations, not only once
    for tl in tls:
        tl.close()


def test_a_fold_out_of_range_falls_back_to_per_node_factors(oracle_lib):
    """Branches of ~1e-12 substitutions and random tip states: nearly every node's factor is ~1e-12, and the unstored runs of this
    tree (buffers of 2 MiB: definitions of up to 24 nodes) would fold 17 and more of them — beyond the safe range (1e100): refused; the
    plan is resolved again with per-node factors: the same bits as with folding switched off."""
    T, P = 96, 17000
    rng = np.random.default_rng(99)
    tree = trees.coalescent_tree(T, rng, root_height=1e-10)
    pi = np.array([0.3, 0.2, 0.22, 0.28])
    eig = substmodel.gtr([1.0, 4.0, 0.8, 1.2, 4.5, 1.0], pi)
    tips = rng.integers(0, 4, size=(T, P)).astype(np.int32)
    wl = Workload("conflict", tree, eig, pi, [1.0, 1.0, 1.0, 1.0], [0.25, 0.25, 0.25, 0.25], np.ascontiguousarray(tips), np.ones(P), 4)
    f0, fv, fs, _, _, fstats, fh = steady_state(wl, True)
    u0, uv, us, _, _, ustats, uh = steady_state(wl, False)
    assert fh["fold_builds"] > 0 and fh["folded_vectors"] == 0           # tried, refused
    assert fstats["scale_reads"] == ustats["scale_reads"]
    assert fv == uv and np.array_equal(fs, us)
    o = BeagleTreeLikelihood(wl, library=oracle_lib, rescaling=RESCALE_DYNAMIC, delay_rescaling=False)
    ref = o.getLogLikelihood()
    o.close()
    assert helpers.rel_err(fv[-1], ref) <= 1e-4       # (matrix entries of 1e-12 are mostly the rounding of U exp(lambda t) U^-1 itself)
