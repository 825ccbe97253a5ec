"""The 4-state engine — and, on the T32 layout, the <= 20-state one — never writes tip-tip nodes' partials to HBM unless
somebody needs the real data ("virtual cherries", engine_levels.cpp; kernels_mfma.hip cherryOperands).  BEAGLE semantics must survive that: a partials buffer keeps the value its op gave it even
if the tip states, the matrices or the scale buffer it was computed from are changed afterwards.  Every step below is
issued identically to the HIP engine and to the CPU oracle, and every buffer is compared after every step."""
import numpy as np
import pytest

import beast_mcmc_amd as bm
import helpers
from beast_mcmc_amd.inputs import substmodel

pytestmark = pytest.mark.gpu
NONE = bm.beagle.NONE
T, P, C = 4, 300, 4          # tips 0..3; internal buffers 4 = (0,1) cherry, 5 = (2,3) cherry, 6 = (4,5) root, 7 = (4,2)


def make(lib, states, eig, rates, S=4):
    b = bm.beagle.Beagle(T, 8, T, S, P, 1, 8, C, 6, library=lib)
    for t in range(T):
        b.setTipStates(t, states[t])
    b.setEigenDecomposition(0, eig.evec, eig.ievc, eig.evals)
    b.setCategoryRates(rates)
    b.updateTransitionMatrices(0, [0, 1, 2, 3, 4, 5], None, None, [0.1, 0.3, 0.2, 0.05, 0.4, 0.15], 6)
    return b


def same(g, o, bufs, what):
    for x in bufs:
        a, b = g.getPartials(x, NONE), o.getPartials(x, NONE)
        assert np.max(np.abs(a - b) / np.maximum(np.abs(b).max(axis=(0, 2), keepdims=True), 1e-300)) <= 1e-12, (what, x)


@pytest.fixture
def cherries_at_61(monkeypatch):
    """21..64 states define tip-tip nodes only on request (an experiment switch, read at instance creation)."""
    monkeypatch.setenv("BEAGLE_MI355_CHERRY61", "1")


@pytest.mark.parametrize("S", [4, 20, 16, 61])
def test_virtual_cherries_keep_beagle_semantics(S, oracle_lib, engine_lib, cherries_at_61):
    rng = np.random.default_rng(5)
    states = rng.integers(0, S + 1, size=(T, P)).astype(np.int32)       # S = missing
    if S == 4:
        pi = np.array([0.3, 0.2, 0.25, 0.25])
        eig = substmodel.gtr([1.0, 3.0, 0.7, 1.1, 4.0, 1.0], pi)
    else:
        eig, pi = substmodel.random_reversible(S, rng)
    rates = [0.1, 0.5, 1.0, 2.4]
    g, o = make(engine_lib, states, eig, rates, S), make(oracle_lib, states, eig, rates, S)
    try:
        both = (g, o)
        # 1. write-mode rescaling: cherries 4, 5 (scale buffers 0, 1), root 6 (scale 2)
        ops1 = [4, 0, NONE, 0, 0, 1, 1,   5, 1, NONE, 2, 2, 3, 3,   6, 2, NONE, 4, 4, 5, 5]
        for b in both:
            b.updatePartials(ops1, 3, NONE)
        same(g, o, [6], "root after write-mode ops")
        assert np.max(np.abs(g.getLogScaleFactors(0) - o.getLogScaleFactors(0))) < 1e-12
        # 2. read-mode: recompute 4 (reads scale 0) and a second parent 7 = (4, tip 2) that fuses it
        ops2 = [4, NONE, 0, 0, 0, 1, 1,   7, NONE, 2, 4, 4, 2, 2]
        for b in both:
            b.updatePartials(ops2, 2, NONE)
        same(g, o, [7], "parent of a read-mode virtual cherry")
        # 3. change a tip of cherry 4: buffers 4 and 7 must keep their OLD values (no op recomputed them) ...
        new0 = rng.integers(0, S, size=P).astype(np.int32)
        for b in both:
            b.setTipStates(0, new0)
        same(g, o, [4, 7, 6], "after setTipStates")
        # ... and a later parent that uses buffer 4 WITHOUT recomputing it sees those old values
        for b in both:
            b.updatePartials([7, NONE, NONE, 4, 4, 3, 3], 1, NONE)
        same(g, o, [7], "later parent of a materialised cherry")
        # 4. recompute 4 from the new states, then overwrite its scale buffer: 4 keeps its value, parents still agree
        for b in both:
            b.updatePartials([4, NONE, 0, 0, 0, 1, 1], 1, NONE)
            b.resetScaleFactors(0)
            b.updatePartials([7, NONE, NONE, 4, 4, 2, 2], 1, NONE)
        same(g, o, [4, 7], "after the cherry's scale buffer was reset")
        # 5. new matrices for the cherry's branches: buffer 5 (computed in step 1, never read back so far) is unchanged
        for b in both:
            b.updateTransitionMatrices(0, [2, 3], None, None, [0.9, 0.8], 2)
            b.updatePartials([6, NONE, NONE, 4, 4, 5, 5], 1, NONE)
        same(g, o, [5, 6], "after the cherry's branch matrices were rewritten")
        # 6. root likelihood straight from a cherry buffer
        for b in both:
            b.setStateFrequencies(0, pi); b.setCategoryWeights(0, [0.25] * 4); b.setPatternWeights(np.ones(P))
            b.updatePartials([5, NONE, NONE, 2, 2, 3, 3], 1, NONE)
        out_g, out_o = [0.0], [0.0]
        g.calculateRootLogLikelihoods([5], [0], [0], [NONE], 1, out_g)
        o.calculateRootLogLikelihoods([5], [0], [0], [NONE], 1, out_o)
        assert helpers.rel_err(out_g[0], out_o[0]) <= 1e-12
    finally:
        g.finalize(); o.finalize()


@pytest.mark.parametrize("S", [4, 20, 61])
def test_virtual_and_stored_cherries_agree_bitwise(S, engine_lib, cherries_at_61):
    """BEAGLE_MI355_NO_VIRTUAL is read at instance creation: the same evaluation with virtual cherries on and off
    must give the same lnL to the last bit (the fused recomputation repeats the cherry op's own arithmetic)."""
    import os
    from beast_mcmc_amd.treelikelihood import BeagleTreeLikelihood, RESCALE_ALWAYS, RESCALE_DYNAMIC
    wl = helpers.random_workload(60, 2000, S, 4, seed=77) if S <= 20 else helpers.random_workload(30, 300, S, 2, seed=77)
    vals = {}
    # (4 states: with read-mode folding the unstored nodes' factors are applied in one multiplication — the same numbers in another
    # order, tests/test_gpu_scale_fold.py; bit-equality with the all-stored evaluation is a property of per-node factors)
    # ... and so is the cumulative buffer's: from the per-node factors (the per-slice products of round 6 follow the slicing, which the
    # two plans do differently: tests/test_gpu_slice_sums.py holds them to rounding)
    os.environ["BEAGLE_MI355_NO_SCALE_FOLD"] = "1"
    os.environ["BEAGLE_MI355_NO_SLICE_SUMS"] = "1"
    for flag in ("0", "1"):
        os.environ["BEAGLE_MI355_NO_VIRTUAL"] = flag
        try:
            for scheme, delay in ((RESCALE_ALWAYS, False), (RESCALE_DYNAMIC, False)):
                tl = BeagleTreeLikelihood(wl, rescaling=scheme, delay_rescaling=delay)
                a = tl.getLogLikelihood()
                tl.storeState(); tl.makeDirty()
                b = tl.getLogLikelihood()
                vals.setdefault((scheme, "first"), []).append(a)
                vals.setdefault((scheme, "second"), []).append(b)
                tl.close()
        finally:
            os.environ.pop("BEAGLE_MI355_NO_VIRTUAL", None)
            if flag == "1":
                os.environ.pop("BEAGLE_MI355_NO_SCALE_FOLD", None)
                os.environ.pop("BEAGLE_MI355_NO_SLICE_SUMS", None)
    for k, v in vals.items():
        assert v[0] == v[1], (k, v)


@pytest.mark.parametrize("S", [4, 20, 17])
def test_steady_state_chain_matches_stored_buffers_bitwise(S, oracle_lib, monkeypatch):
    """An MCMC-like chain: the same op lists come back every other evaluation (buffer flips), which is what the
    engine's steady-state fast path keys on (definitions re-confirmed, only the matrix snapshots refreshed).  Model
    parameters, branch rates and node heights change between evaluations, some moves are rejected (restoreState).
    Every evaluation must equal, to the last bit, the same chain with virtual buffers off — and the oracle to 1e-10.
    4 states: the pattern walk (kernels_walk4.hip); 17 and 20 states: the same programs on the T32 layout
    (kernels_mfma.hip k_walkT32) — there also against the level kernels (BEAGLE_MI355_NO_T32_WALK=1; other arithmetic
    order: 1e-12), with the rescaling evaluations of the DYNAMIC scheme taking the level path in between."""
    import os
    from beast_mcmc_amd.treelikelihood import BeagleTreeLikelihood, RESCALE_DYNAMIC
    # (bit-equality with the all-stored chain is a property of per-node scale factors: tests/test_gpu_scale_fold.py holds the folded ones)
    monkeypatch.setenv("BEAGLE_MI355_NO_SCALE_FOLD", "1")
    monkeypatch.setenv("BEAGLE_MI355_NO_SLICE_SUMS", "1")       # (... and of cumulative buffers formed from them: tests/test_gpu_slice_sums.py holds the per-slice products)
    wl = helpers.random_workload(80, 1500, 4, 4, seed=99) if S == 4 else helpers.random_workload(40, 700, S, 4, seed=99)
    rng = np.random.default_rng(3)
    moves = []
    for step in range(14):
        kind = ("model", "rates", "height", "none")[step % 4]
        eig = substmodel.gtr(rng.gamma(2.0, 1.0, size=6) + 0.1, wl.freqs) if S == 4 else substmodel.random_reversible(S, rng)[0]
        moves.append((kind, eig, rng.uniform(0.5, 1.5, size=wl.tree.node_count),
                      int(rng.integers(wl.tip_count, wl.tree.node_count)), step % 5 == 4))

    def chain(tl):
        out = [tl.getLogLikelihood()]
        height = wl.tree.height.copy()          # the chain's own view of the node heights (moves accumulate)
        for kind, eig, rates, node, reject in moves:
            tl.storeState()
            saved = height.copy()
            if kind == "model":
                tl.set_substitution_model(eig, wl.freqs)
            elif kind == "rates":
                tl.set_branch_rates(rates)
            elif kind == "height" and node != wl.tree.root:
                lo = max(height[wl.tree.left[node]], height[wl.tree.right[node]])
                hi = height[wl.tree.parent[node]]
                height[node] = lo + 0.37 * (hi - lo)
                tl.set_node_height(node, float(height[node]))
            else:
                tl.makeDirty()
            out.append(tl.getLogLikelihood())
            if reject:
                tl.restoreState()
                height = saved
                out.append(tl.getLogLikelihood())
        return out

    runs, stats = {}, {}
    for name, env in (("virtual", {}), ("stored", {"BEAGLE_MI355_NO_VIRTUAL": "1"})) + ((("levels", {"BEAGLE_MI355_NO_T32_WALK": "1"}),) if S != 4 else ()):
        os.environ.update(env)
        try:
            tl = BeagleTreeLikelihood(wl, rescaling=RESCALE_DYNAMIC, delay_rescaling=False)
            raw = bm.beagle.Beagle.attach(tl)
            raw.kernelTimer(True)
            runs[name] = chain(tl)
            stats[name] = raw.walkStats()
            tl.close()
        finally:
            for k in env:
                os.environ.pop(k, None)
    assert runs["virtual"] == runs["stored"]
    assert stats["virtual"]["walks"] > 0 and stats["virtual"]["stored"] < stats["stored"]["stored"] == stats["stored"]["micro_ops"]
    if S != 4:
        assert stats["levels"]["walks"] == 0                       # (the level kernels count their operations, not walks)
        for a, b in zip(runs["virtual"], runs["levels"]):
            assert helpers.rel_err(a, b) <= 1e-12
    o = BeagleTreeLikelihood(wl, library=oracle_lib, rescaling=RESCALE_DYNAMIC, delay_rescaling=False)
    ref = chain(o)
    o.close()
    assert len(ref) == len(runs["virtual"])
    for a, b in zip(runs["virtual"], ref):
        assert helpers.rel_err(a, b) <= 1e-10
