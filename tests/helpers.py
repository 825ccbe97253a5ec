"""Shared test plumbing: the CPU oracle as an EngineLibrary, golden fixtures as workloads."""
import json
import os

import numpy as np

import beast_mcmc_amd as bm
import reference_quantile
from beast_mcmc_amd.inputs import patterns, siterates, substmodel, trees
from beast_mcmc_amd.inputs.synth import Workload

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
ORACLE_SO = os.path.join(ROOT, "oracle", "liboracle_beagle.so")

_oracle = None


def granted_cpus():
    """CPUs this process may actually use: affinity mask capped by the cgroup quota (a container can see 256 cores and be
    granted 16 — OpenMP's default thread count then oversubscribes the oracle badly)."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except Exception:
            pass
    return n


def oracle_library():
    """The CPU oracle (oracle/beagle_cpu_oracle.c) behind the same binding class as the engine."""
    global _oracle
    if _oracle is None:
        _oracle = bm.beagle.EngineLibrary(ORACLE_SO, prefix="oracle_")
        n = granted_cpus()
        if n < _oracle.lib.oracle_threads():
            _oracle.lib.oracle_set_threads(int(n))
    return _oracle


def golden(name):
    return json.load(open(os.path.join(GOLDEN, name)))


def rel_err(a, b):
    return abs(a - b) / max(abs(b), 1e-300)


def primates_case(case, site_model="new"):
    """One TreeDataLikelihoodTest / LikelihoodTest case of tests/golden/primates.json as a Workload."""
    g = golden("primates.json")
    rows = np.stack([patterns.nucleotide_states(s) for s in g["sequences"]])
    pats, weights = patterns.site_patterns(rows, unique=True)
    tree = trees.from_nested(_tuplify(g["tree_nested"]), len(g["taxa"]))
    pi = np.full(4, 0.25) if case["pi"] == "equal" else patterns.empirical_frequencies(rows)
    if case["model"] == "hky":
        eig = substmodel.hky(case["kappa"], pi)
    else:
        eig = substmodel.gtr(case["rates"], pi)
    cls = siterates.GammaSiteRateModel if site_model == "new" else siterates.OldGammaSiteModel
    sm = cls(alpha=case.get("alpha"), gamma_categories=case.get("cats", 1), p_inv=case.get("pinv"),
             quantile=reference_quantile.gamma_quantile)      # the golden values were computed with the reference's quantile
    rates, props = sm.category_rates_and_proportions()
    return Workload("primates:" + case["name"], tree, eig, pi, rates, props, pats, weights, 4)


def _tuplify(x):
    if isinstance(x, list):
        return tuple(_tuplify(y) for y in x)
    return x


def random_workload(tip_count, pattern_count, state_count, categories, seed, tree_kind="coalescent",
                    root_to_tip=0.5, unknown_fraction=0.05):
    """Small seeded workload with arbitrary state count (random reversible model for S != 4)."""
    rng = np.random.default_rng(seed)
    if state_count == 4:
        pi = rng.dirichlet(np.full(4, 10.0))
        eig = substmodel.gtr(rng.gamma(2.0, 1.0, size=6) + 0.1, pi)
    else:
        eig, pi = substmodel.random_reversible(state_count, rng)
    from beast_mcmc_amd.inputs import synth
    return synth.make_workload("rand-S%d" % state_count, tip_count, pattern_count, eig, pi, alpha=0.7,
                               categories=categories, seed=seed, tree_kind=tree_kind,
                               root_to_tip=root_to_tip, unknown_fraction=unknown_fraction)


def proposed_height(tree, node, rng, spread=0.3):
    """A new height for internal node `node` that keeps every branch length positive: somewhere between its higher child and its
    parent (the root: up to 5 % above itself), `spread` of that interval around the current height at most."""
    lo = max(float(tree.height[int(tree.left[node])]), float(tree.height[int(tree.right[node])]))
    parent = [n for n in range(tree.tip_count, tree.node_count) if int(tree.left[n]) == node or int(tree.right[n]) == node]
    h = float(tree.height[node])
    hi = float(tree.height[parent[0]]) if parent else h * 1.05
    a, b = max(lo + 1e-3 * (hi - lo), h - spread * (hi - lo)), min(hi - 1e-3 * (hi - lo), h + spread * (hi - lo))
    return float(rng.uniform(a, b)) if b > a else h


def sample_with_tail(pattern_count, n_sample, seed):
    """Pattern indices for a sampled full-size check: a seeded random sample PLUS, always, the last 256 patterns — the ragged
    last 128-pattern group of a buffer (and the last 32-pattern tile) must be in every sample, not left to the seed."""
    rnd = np.random.default_rng(seed).choice(pattern_count, size=min(n_sample, pattern_count), replace=False)
    tail = np.arange(max(0, pattern_count - 256), pattern_count)
    return np.unique(np.concatenate([rnd, tail]))


def raw_binding(tl):
    """A Beagle binding object over a BeagleTreeLikelihood's existing instance (no new instance): the getters the host driver does
    not wrap."""
    import beast_mcmc_amd as bm
    b = bm.beagle.Beagle.__new__(bm.beagle.Beagle)
    b.lib = tl.engine
    b._f = tl.engine.fn
    b.instance = tl.instance
    b.stateCount, b.patternCount, b.categoryCount = tl.state_count, tl.pattern_count, tl.category_count
    return b


def walk_stats(tl):
    """The engine's walk counters for the instance behind a BeagleTreeLikelihood (include/beagle_mi355.h beagleMi355WalkStats):
    which kernel a list ran on — 'walks' pattern-walk launches, of them 'fast_walks' on the assembly loop k_walk4_fast."""
    import beast_mcmc_amd as bm
    raw = bm.beagle.Beagle.__new__(bm.beagle.Beagle)
    raw.lib, raw._f, raw.instance = tl.engine, tl.engine.fn, tl.instance
    return raw.walkStats()
