"""Seeded synthetic workloads of the shapes BASELINE.json names.

A workload is everything the reference's caller would hand to the engine: a time-tree, an eigen
system, category rates/weights, root frequencies, compact tip states for exactly P *unique* site
patterns, and pattern weights.  Sequences are simulated down the tree under the same model, so the
per-pattern likelihoods have realistic magnitudes (and underflow without rescaling at T = 1000,
which is what forces the reference's rescaling protocol into the benchmark).
"""
import os

import numpy as np

from . import substmodel, trees
from .siterates import GammaSiteRateModel


class Workload:
    def __init__(self, name, tree, eig, freqs, cat_rates, cat_weights, tip_states, weights, state_count):
        self.name = name
        self.tree = tree
        self.eig = eig
        self.freqs = np.ascontiguousarray(freqs, dtype=np.float64)
        self.cat_rates = np.ascontiguousarray(cat_rates, dtype=np.float64)
        self.cat_weights = np.ascontiguousarray(cat_weights, dtype=np.float64)
        self.tip_states = tip_states            # int32 [T][P], value >= S means unknown
        self.weights = np.ascontiguousarray(weights, dtype=np.float64)
        self.state_count = state_count

    @property
    def tip_count(self):
        return self.tree.tip_count

    @property
    def pattern_count(self):
        return self.tip_states.shape[1]

    @property
    def category_count(self):
        return len(self.cat_rates)

    def shard(self, start, stop):
        """Contiguous pattern block [start, stop) — what one GPU of a pattern-sharded job owns."""
        return Workload(self.name, self.tree, self.eig, self.freqs, self.cat_rates, self.cat_weights,
                        np.ascontiguousarray(self.tip_states[:, start:stop]), self.weights[start:stop],
                        self.state_count)


def simulate_unique_patterns(tree, eig, freqs, cat_rates, cat_weights, pattern_count, rng, batch=None):
    """Simulate columns down ``tree`` until ``pattern_count`` distinct columns exist."""
    s = len(freqs)
    t = tree.tip_count
    cat_rates = np.asarray(cat_rates)
    order = [n for n in reversed(tree.postorder())]          # pre-order: parents before children
    # per-node cumulative transition rows, [C][S][S]
    cum = {}
    for n in order:
        if n == tree.root:
            continue
        bl = tree.branch_length(n)
        mats = np.stack([np.clip(eig.transition_probabilities(bl * r), 0.0, None) for r in cat_rates])
        mats /= mats.sum(axis=2, keepdims=True)
        cum[n] = np.cumsum(mats, axis=2)
    dtype = np.uint8
    h1 = rng.integers(1, 2 ** 62, size=t, dtype=np.int64) | 1
    h2 = rng.integers(1, 2 ** 62, size=t, dtype=np.int64) | 1
    if t * np.log(s) < np.log(pattern_count * 4.0):
        raise ValueError("%d unique patterns requested but only %d^%d distinct columns exist" % (pattern_count, s, t))
    seen = set()
    cols = []
    have = 0
    rounds = 0
    while have < pattern_count:
        rounds += 1
        if rounds > 200:
            raise RuntimeError("could not simulate %d unique patterns (have %d)" % (pattern_count, have))
        n_sites = batch or max(1024, int((pattern_count - have) * 1.25) + 64)
        cats = rng.choice(len(cat_rates), size=n_sites, p=np.asarray(cat_weights) / np.sum(cat_weights))
        states = np.empty((tree.node_count, n_sites), dtype=dtype)
        states[tree.root] = rng.choice(s, size=n_sites, p=freqs / freqs.sum())
        for n in order:
            if n == tree.root:
                continue
            rows = cum[n][cats, states[tree.parent[n]]]      # [n_sites][S]
            u = rng.random(n_sites)[:, None]
            states[n] = np.minimum((u > rows).sum(axis=1), s - 1)
        tips = states[:t]
        with np.errstate(over="ignore"):
            k1 = (tips.astype(np.int64) * h1[:, None]).sum(axis=0)
            k2 = (tips.astype(np.int64) * h2[:, None]).sum(axis=0)
        keep = []
        for i in range(n_sites):
            key = (int(k1[i]), int(k2[i]))
            if key not in seen:
                seen.add(key)
                keep.append(i)
                if have + len(keep) == pattern_count:
                    break
        cols.append(tips[:, keep])
        have += len(keep)
    return np.concatenate(cols, axis=1)


def make_workload(name, tip_count, pattern_count, eig, freqs, alpha=0.5, categories=4, seed=1,
                  tree_kind="coalescent", root_to_tip=0.5, unknown_fraction=0.01):
    """Tree from seed, data from seed+1 (weights are random integers 1..20,
    ~1 % of tip states unknown)."""
    rng_tree = np.random.default_rng(seed)
    rng_data = np.random.default_rng(seed + 1)
    if tree_kind == "coalescent":
        tree = trees.coalescent_tree(tip_count, rng_tree, root_height=root_to_tip)
    elif tree_kind == "yule":
        tree = trees.yule_tree(tip_count, rng_tree, root_height=root_to_tip)
    elif tree_kind == "caterpillar":
        tree = trees.caterpillar_tree(tip_count, root_height=root_to_tip)
    else:
        raise ValueError(tree_kind)
    if categories > 1:
        rates, props = GammaSiteRateModel(alpha=alpha, gamma_categories=categories).category_rates_and_proportions()
    else:
        rates, props = [1.0], [1.0]
    s = len(freqs)
    tips = simulate_unique_patterns(tree, eig, np.asarray(freqs), rates, props, pattern_count, rng_data)
    tip_states = tips.astype(np.int32)
    if unknown_fraction > 0:
        mask = rng_data.random(tip_states.shape) < unknown_fraction
        tip_states[mask] = s                                  # code >= S: missing / ambiguous
    weights = rng_data.integers(1, 21, size=pattern_count).astype(np.float64)
    return Workload(name, tree, eig, freqs, rates, props, np.ascontiguousarray(tip_states), weights, s)


def cached(path, maker):
    """Generate a workload once per machine: ``maker()`` on a miss, unpickle ``path`` on a hit (the seeded
    simulation of 1000 x 1e5 states takes ~10-30 s; several bench invocations on one box share it)."""
    import os
    import pickle
    if path and os.path.exists(path):
        with open(path, "rb") as fh:
            return pickle.load(fh)
    wl = maker()
    if path:
        tmp = "%s.%d.tmp" % (path, os.getpid())
        with open(tmp, "wb") as fh:
            pickle.dump(wl, fh, protocol=4)
        os.replace(tmp, path)
    return wl


# BASELINE.json configs.  `scale` shrinks taxa and patterns for parity-test sizes.
def config_a(scale=1.0, seed=1, tree_kind="coalescent"):
    """GTR+G4 nucleotide, 1000 taxa x 1e5 unique patterns (the metric's config)."""
    pi = np.array([0.30, 0.20, 0.22, 0.28])
    eig = substmodel.gtr([1.0, 4.0, 0.8, 1.2, 4.5, 1.0], pi)
    return make_workload("A:GTR+G4", max(4, int(1000 * scale)), max(64, int(100000 * scale)), eig, pi,
                         seed=seed, tree_kind=tree_kind)


def config_b(scale=1.0, seed=11):
    """20-state reversible + G4, 500 taxa x 5e4 patterns (seeded exchangeabilities, not the WAG table)."""
    eig, pi = substmodel.random_reversible(20, np.random.default_rng(seed + 100))
    return make_workload("B:AA20+G4", max(4, int(500 * scale)), max(64, int(50000 * scale)), eig, pi, seed=seed)


def config_c(scale=1.0, seed=21):
    """GY94 codon (61 states) + G4, 200 taxa x 2e4 patterns."""
    rng = np.random.default_rng(seed + 100)
    pi = rng.dirichlet(np.full(61, 20.0))
    eig, pi = substmodel.gy94(2.0, 0.2, pi)
    return make_workload("C:GY94+G4", max(4, int(200 * scale)), max(64, int(20000 * scale)), eig, pi, seed=seed)


def config_d(categories=1, seed=31):
    """benchmark1-like: 1441 taxa x 593 patterns, HKY kappa=2, equal frequencies (launch-latency case)."""
    pi = np.full(4, 0.25)
    eig = substmodel.hky(2.0, pi)
    return make_workload("D:benchmark1-like", 1441, 593, eig, pi, categories=categories, seed=seed,
                         root_to_tip=0.05)


def simulate_sites(tree, eig, freqs, cat_rates, cat_weights, n_sites, rng):
    """Simulate exactly ``n_sites`` alignment columns down ``tree`` (no uniqueness filter): uint8 [T][n_sites]."""
    s = len(freqs)
    cat_rates = np.asarray(cat_rates)
    order = [n for n in reversed(tree.postorder())]
    cats = rng.choice(len(cat_rates), size=n_sites, p=np.asarray(cat_weights) / np.sum(cat_weights))
    states = np.empty((tree.node_count, n_sites), dtype=np.uint8)
    states[tree.root] = rng.choice(s, size=n_sites, p=np.asarray(freqs) / np.sum(freqs))
    for n in order:
        if n == tree.root:
            continue
        bl = tree.branch_length(n)
        mats = np.stack([np.clip(eig.transition_probabilities(bl * r), 0.0, None) for r in cat_rates])
        mats /= mats.sum(axis=2, keepdims=True)
        rows = np.cumsum(mats, axis=2)[cats, states[tree.parent[n]]]
        u = rng.random(n_sites)[:, None]
        states[n] = np.minimum((u > rows).sum(axis=1), s - 1)
    return states[:tree.tip_count]


class PartitionedWorkload:
    """Several data partitions on ONE tree, each with its own substitution and site model — what
    MultiPartitionDataLikelihoodDelegate hands to a single instance (patterns concatenated, partition = contiguous
    pattern range, src/dr/evomodel/treedatalikelihood/MultiPartitionDataLikelihoodDelegate.java:520-553)."""

    def __init__(self, name, tree, parts):
        self.name = name
        self.tree = tree
        self.parts = parts                      # list of Workload on the same tree

    @property
    def tip_count(self):
        return self.tree.tip_count

    @property
    def pattern_count(self):
        return sum(w.pattern_count for w in self.parts)

    @property
    def pattern_counts(self):
        return [w.pattern_count for w in self.parts]

    def shard(self, rank, world):
        """What one GPU of a pattern-sharded job owns: block `rank` of every partition's pattern range."""
        from . import patterns
        parts = []
        for w in self.parts:
            s, e = patterns.shard_bounds(w.pattern_count, world)[rank]
            parts.append(w.shard(s, e))
        return PartitionedWorkload(self.name, self.tree, parts)


def config_e(scale=1.0, seed=41):
    """Makona-like (BASELINE.json config 5; the real alignment is absent from the reference tree): 1610 taxa,
    an 18 992-nt genome as four nucleotide partitions — codon positions 1, 2, 3 and non-coding — HKY+G4 each with its own
    kappa / frequencies / alpha / relative rate, simulated at low divergence (an outbreak: root-to-tip ~8e-3
    substitutions per site) and compressed to unique site patterns per partition."""
    from . import patterns
    t = max(8, int(1610 * scale))
    rng = np.random.default_rng(seed)
    tree = trees.coalescent_tree(t, rng, root_height=0.008)
    n_sites = [max(40, int(x * scale)) for x in (4900, 4900, 4900, 4292)]
    kappas, alphas, mus = (4.0, 3.5, 8.0, 6.0), (0.6, 0.4, 1.2, 0.8), (0.6, 0.4, 2.2, 0.9)
    pis = ([0.30, 0.20, 0.28, 0.22], [0.32, 0.22, 0.18, 0.28], [0.28, 0.24, 0.22, 0.26], [0.31, 0.21, 0.20, 0.28])
    parts = []
    for k in range(4):
        pi = np.asarray(pis[k])
        eig = substmodel.hky(kappas[k], pi)
        rates, props = GammaSiteRateModel(alpha=alphas[k], gamma_categories=4).category_rates_and_proportions()
        rates = np.asarray(rates) * mus[k]                  # the partition's relative rate folded into its category rates
        cols = simulate_sites(tree, eig, pi, rates, props, n_sites[k], rng).astype(np.int32)
        for taxon in rng.choice(t, size=max(1, t // 40), replace=False):      # sequencing gaps: a stretch of a few genomes
            a = int(rng.integers(0, n_sites[k])); cols[taxon, a:a + max(1, n_sites[k] // 10)] = 4
        pats, weights = patterns.site_patterns(cols, unique=True)
        parts.append(Workload("E:part%d" % k, tree, eig, pi, rates, props, pats, weights, 4))
    return PartitionedWorkload("E:Makona-like 4 x HKY+G4", tree, parts)


def from_pattern_fixture(path, seed=51):
    """A workload from a unique-site-pattern fixture (tests/golden/benchmark{1,2}_patterns.npz: the REAL alignments of the
    reference's examples/Benchmarks inputs, written by tests/golden/make_fixtures.py --benchmarks) with the model the XML
    names.  The XMLs draw a random coalescent starting tree; here it is a seeded coalescent of the same scale."""
    import json
    z = np.load(path, allow_pickle=False)
    model = json.loads(str(z["model"]))
    pats = z["patterns"].astype(np.int32)
    pats[pats > 3] = 4                                        # useAmbiguities="false": every ambiguity code is "unknown"
    pi = np.asarray(model["pi"], dtype=float)
    eig = substmodel.hky(model["kappa"], pi) if model["model"] == "hky" else substmodel.gtr(model["rates"], pi)
    cats = int(model.get("gamma_categories", 1))
    if cats > 1:
        rates, props = GammaSiteRateModel(alpha=model["alpha"], gamma_categories=cats).category_rates_and_proportions()
    else:
        rates, props = [1.0], [1.0]
    # substitutions from root to tip: the XML's rootHeight, or clock rate x ~2 N_e years of a constant-size coalescent
    height = model.get("root_height") or model["clock_rate"] * 2.0 * model["popSize_years"]
    tree = trees.coalescent_tree(pats.shape[0], np.random.default_rng(seed), root_height=height)
    return Workload(os.path.basename(path).replace("_patterns.npz", "") + " (real alignment)", tree, eig, pi, rates, props,
                    np.ascontiguousarray(pats), z["weights"].astype(np.float64), 4)
