"""Pre-order partials and branch gradients on the HIP engine, through the C ABI, against the CPU
oracle driven by the identical call sequence (beast-mcmc_amd/gradient.py mirrors the reference's gradient delegates).

The oracle's gradients are pinned on the CPU tier against finite differences of the golden-pinned log-likelihood
(tests/test_oracle_golden.py); here the bound is the path's 1e-10 relative tolerance.
"""
import numpy as np
import pytest

import beast_mcmc_amd as bm
import helpers
from beast_mcmc_amd.gradient import BranchGradient

pytestmark = pytest.mark.gpu

REL_TOL = 1e-10


def close(a, b, what):
    a, b = np.asarray(a, dtype=float).ravel(), np.asarray(b, dtype=float).ravel()
    scale = max(1.0, float(np.max(np.abs(b))))
    err = float(np.max(np.abs(a - b))) / scale
    assert err <= REL_TOL, (what, err)


@pytest.mark.parametrize("S,C,T,P,rescale", [
    (4, 4, 9, 300, False),       # 4-state path; small tree: most internal nodes are virtual subtrees
    (4, 1, 40, 1000, False),     # deeper tree: virtual siblings have to be materialised for the pre-order pass
    (4, 3, 25, 257, True),       # post-order pass with scale factors (the derivative ratio is scale-free)
    (20, 2, 8, 100, False),      # T32 layout, 5 state tiles
    (61, 1, 6, 70, False),       # T32 layout, 16 state tiles
    (7, 2, 6, 50, False),        # general-S layout
])
def test_gradient_matches_oracle(S, C, T, P, rescale, oracle_lib):
    wl = helpers.random_workload(T, P, S, C, seed=100 + S + T)
    g = BranchGradient(wl, rescale=rescale)
    o = BranchGradient(wl, rescale=rescale, library=oracle_lib)
    # Sums only.  With 4 states the engine holds the pre-order list back (no scale indices in it: the derivative ratio is scale-
    # free) and answers from it.  The first evaluation of an instance runs the list together with the derivatives (the engine
    # starts to keep track of which scale factor went into which post-order partial only when a pre-order list has arrived);
    # from the second one on no pre-order partial is written and the list stays held — with or without rescaling in the
    # post-order pass.  Other state counts go operation by operation.  Same numbers every way.
    lo0, go0 = o.gradient()
    lf0, gf0 = g.gradient()
    assert helpers.rel_err(lf0, lo0) <= REL_TOL
    close(gf0, go0, "gradient (first evaluation)")
    lf, gf = g.gradient()
    none = {"fused": 0, "by_operation": 0, "walked": 0, "late": 0}
    assert g.b.gradientStats() == dict(none, **({"by_operation": 2} if S != 4 else {"fused": 1, "walked": 1}))
    lo, go, ho, po = o.gradient(second=True, per_pattern=True)
    assert helpers.rel_err(lf, lo) <= REL_TOL
    close(gf, go, "gradient (sums only)")
    for n in range(g.N):
        if n != wl.tree.root:
            a, b = g.pre_partials(n).reshape(C, P, S), o.pre_partials(n).reshape(C, P, S)
            ref = np.max(np.abs(b), axis=(0, 2), keepdims=True)
            assert np.max(np.abs(a - b) / np.maximum(ref, 1e-300)) <= REL_TOL, n
    assert g.b.gradientStats()["late"] == (1 if S == 4 else 0)      # the read made the held list run
    lg, gg, hg, pg = g.gradient(second=True, per_pattern=True)
    assert helpers.rel_err(lg, lo) <= REL_TOL
    close(gg, go, "gradient")
    close(hg, ho, "second derivatives")
    close(pg, po, "per-pattern derivatives")
    cg, co = g.cross_products(), o.cross_products()
    close(cg, co, "cross products")
    q = (wl.eig.evec * wl.eig.evals[None, :]) @ wl.eig.ievc
    assert helpers.rel_err(float(np.sum(q * cg)), float(np.dot(g.branch_lengths, gg))) <= 1e-9      # exact direction (scaling Q)
    for n in range(g.N):
        if n == wl.tree.root:
            continue
        a = g.pre_partials(n).reshape(C, P, S)
        b = o.pre_partials(n).reshape(C, P, S)
        ref = np.max(np.abs(b), axis=(0, 2), keepdims=True)
        assert np.max(np.abs(a - b) / np.maximum(ref, 1e-300)) <= REL_TOL, n
    # the likelihood path is undisturbed by the gradient pass (virtual buffers were materialised, not corrupted)
    assert g.log_likelihood() == lg
    g.close(); o.close()


def test_gradient_matches_finite_differences_on_the_engine():
    wl = helpers.random_workload(10, 400, 4, 4, seed=77)
    g = BranchGradient(wl)
    lnl, grad = g.gradient()
    for n in g.edges[:6]:
        t = g.branch_lengths[n]
        h = 1e-4 * t
        g.set_branch_length(n, t + h); up = g.log_likelihood()
        g.set_branch_length(n, t - h); dn = g.log_likelihood()
        g.set_branch_length(n, t)
        noise = 50 * np.finfo(float).eps * abs(lnl) / h
        assert abs((up - dn) / (2 * h) - grad[n]) <= 1e-5 * max(1.0, abs(grad[n])) + noise
    g.close()


def test_pre_order_entry_points_and_errors(oracle_lib):
    wl = helpers.random_workload(6, 90, 4, 2, seed=5)
    g = BranchGradient(wl)
    o = BranchGradient(wl, library=oracle_lib)
    g.log_likelihood(); o.log_likelihood()
    root = wl.tree.root
    # setRootPrePartials == setPartials(frequencies replicated)
    for d in (g, o):
        d.b.setRootPrePartials([d.pre_offset + root], [0], 1)
    close(g.pre_partials(root), np.tile(wl.freqs, g.P * g.C), "root pre-partials")
    close(g.pre_partials(root), o.pre_partials(root), "root pre-partials vs oracle")
    # transposeTransitionMatrices
    g.b.transposeTransitionMatrices([0, 1], [g.q_index, g.q2_index], 2)
    for k, idx in enumerate((g.q_index, g.q2_index)):
        m = g.b.getTransitionMatrix(k).reshape(g.C, g.S, g.S)
        assert np.array_equal(g.b.getTransitionMatrix(idx).reshape(g.C, g.S, g.S), np.transpose(m, (0, 2, 1)))
    # pre-order ops with a write-scale index: rescaled partials times the factor reproduce the unscaled ones
    wl2 = helpers.random_workload(6, 90, 4, 2, seed=5)
    s = BranchGradient(wl2, rescale=True)      # instance with scale buffers
    s.log_likelihood()
    s.b.setRootPrePartials([s.pre_offset + root], [0], 1)
    ops = s._pre_ops.copy().reshape(-1, 7)
    ops[:, 1] = 0                                # every op rescales into scale buffer 0 (overwritten op after op)
    s.b.updatePrePartials(ops[:1].ravel(), 1, bm.beagle.NONE)
    child = int(ops[0, 0])
    scaled = s.b.getPartials(child, bm.beagle.NONE).reshape(s.C, s.P, s.S)
    logf = s.b.getLogScaleFactors(0)
    assert np.allclose(np.max(scaled, axis=(0, 2)), 1.0, rtol=0, atol=1e-15)
    g.b.setRootPrePartials([g.pre_offset + root], [0], 1)
    g.b.updatePrePartials(g._pre_ops[:7], 1, bm.beagle.NONE)
    plain = g.pre_partials(child - g.pre_offset).reshape(g.C, g.P, g.S)
    close(scaled * np.exp(logf)[None, :, None], plain, "rescaled pre-order partials")
    # errors: out-of-range indices, destination aliasing an input
    bad = g._pre_ops[:7].copy(); bad[0] = 10 ** 6
    with pytest.raises(bm.beagle.BeagleException) as e:
        g.b.updatePrePartials(bad, 1, bm.beagle.NONE)
    assert e.value.code == -5
    bad = g._pre_ops[:7].copy(); bad[0] = bad[3]
    with pytest.raises(bm.beagle.BeagleException) as e:
        g.b.updatePrePartials(bad, 1, bm.beagle.NONE)
    assert e.value.code == -5
    with pytest.raises(bm.beagle.BeagleException) as e:
        g.b.calculateEdgeDifferentials([0], [10 ** 6], [g.q_index], [0], 1)
    assert e.value.code == -5
    g.close(); o.close(); s.close()


def test_held_back_pre_order_list_is_seen_by_every_other_call(oracle_lib):
    """The 4-state engine holds an unscaled pre-order list back (engine_internal.h HeldPreList) and answers the edge-derivative
    sums from it without writing a pre-order partial.  Whatever the caller does next has to see the list as executed: reading
    a pre-order partial, rewriting a branch matrix the list uses, asking for derivatives of a subset of the edges, in another
    order, or for edges the list does not produce."""
    wl = helpers.random_workload(17, 333, 4, 3, seed=909)
    g = BranchGradient(wl)
    o = BranchGradient(wl, library=oracle_lib)
    lo, go, ho = o.gradient(second=True)
    root = wl.tree.root
    post = np.asarray(g.edges, dtype=np.int32)
    pre = post + g.pre_offset
    n = len(post)

    def prepare():
        g.log_likelihood()
        g.b.setPartials(g.pre_offset + root, g._root_pre)
        g.b.updatePrePartials(g._pre_ops, len(g._pre_ops) // 7, bm.beagle.NONE)
        g.b.setDifferentialMatrix(g.q_index, g.infinitesimal(1))

    stats = g.b.gradientStats
    g.gradient()                                                   # (an instance's first evaluation runs its list with the derivatives)
    assert stats() == {"fused": 1, "by_operation": 0, "walked": 0, "late": 0}
    # 1. a subset of the edges, shuffled: one walk, the list stays held; then all of them, then the second derivatives (with the
    #    sums of squares: the list runs together with that call); every pre-order partial exists afterwards
    prepare()
    pick = np.random.default_rng(1).permutation(n)[: n // 2]
    s1, s1sq, _ = g.b.calculateEdgeDifferentials(post[pick], pre[pick], [g.q_index] * len(pick), [0], len(pick), want_squared=False)
    assert s1sq is None and stats() == {"fused": 1, "by_operation": 0, "walked": 1, "late": 0}
    close(s1, go[post[pick]], "subset of the edges")
    s1, _, _ = g.b.calculateEdgeDifferentials(post, pre, [g.q_index] * n, [0], n, want_squared=False)
    assert stats() == {"fused": 1, "by_operation": 0, "walked": 2, "late": 0}
    close(s1, go[post], "all edges, list still held")
    g.b.setDifferentialMatrix(g.q2_index, g.infinitesimal(2))
    s2, _, _ = g.b.calculateEdgeDifferentials(post, pre, [g.q2_index] * n, [0], n, want_squared=False)
    s1, s1sq, _ = g.b.calculateEdgeDifferentials(post, pre, [g.q_index] * n, [0], n)
    assert stats() == {"fused": 2, "by_operation": 0, "walked": 3, "late": 0}
    close(s1, go[post], "first derivatives again, with their squares")
    close(s2 - s1sq, ho[post], "second derivatives")
    for node in (int(post[0]), int(post[-1])):
        close(g.pre_partials(node), o.pre_partials(node), "pre-order partial after the fused call")
    assert stats()["late"] == 0
    # 2. a read of a pre-order partial in between: the list runs at the read
    prepare()
    close(g.pre_partials(int(post[3])), o.pre_partials(int(post[3])), "pre-order partial read before the edge call")
    assert stats()["late"] == 1
    s1, _, _ = g.b.calculateEdgeDifferentials(post, pre, [g.q_index] * n, [0], n)
    close(s1, go[post], "after an intervening read")
    # 3. a call that is not on the short list of calls that leave it held; a branch matrix of the list rewritten (with its own
    #    values): the list runs first.  The differential matrix is not one of the list's: it stays held
    prepare()
    m = g.b.getTransitionMatrix(int(post[0]))
    assert stats()["late"] == 2
    prepare()
    g.b.setTransitionMatrix(int(post[0]), m, 0.0)
    assert stats()["late"] == 3
    s1, _, _ = g.b.calculateEdgeDifferentials(post, pre, [g.q_index] * n, [0], n)
    close(s1, go[post], "after a branch matrix was rewritten")
    # 4. the next evaluation's calls — new branch matrices into the SAME indices (this driver does not double-buffer): the list
    #    reads them, so it runs first, with the old matrices
    prepare()
    before = [g.pre_partials(int(x)).copy() for x in post[:3]]
    prepare()
    t_old = g.branch_lengths.copy()
    g.branch_lengths[:] = t_old * 1.7
    g.log_likelihood()                                             # updateTransitionMatrices hits the held list's matrices
    for x, was in zip(post[:3], before):
        assert np.array_equal(g.pre_partials(int(x)), was)
    g.branch_lengths[:] = t_old
    # 5. edges the held list does not produce: list first, then the derivatives on what is stored
    prepare()
    g.b.getTransitionMatrix(int(post[0]))                          # (runs it)
    first_two = g._pre_ops[:14]                                    # the root's two children again: held
    late = stats()["late"]
    g.b.updatePrePartials(first_two, 2, bm.beagle.NONE)
    s1, _, _ = g.b.calculateEdgeDifferentials(post, pre, [g.q_index] * n, [0], n, want_squared=False)
    assert stats()["late"] == late + 1
    close(s1, go[post], "edges beyond the held list")
    # 6. half a node (one of two siblings): not the shape, runs at once
    by_op = stats()["by_operation"]
    g.b.updatePrePartials(first_two[:7], 1, bm.beagle.NONE)
    assert stats()["by_operation"] == by_op + 1
    e0 = int(first_two[0]) - g.pre_offset
    s1, _, _ = g.b.calculateEdgeDifferentials([e0], [e0 + g.pre_offset], [g.q_index], [0], 1)
    close(s1, go[[e0]], "one sibling only")
    g.close(); o.close()


@pytest.mark.parametrize("rescale", [False, True])
def test_gradient_chain_with_alternating_buffers(rescale, oracle_lib):
    """What a gradient-driven chain does (HMC over branch lengths): every evaluation writes the OTHER set of post-order
    buffers and branch matrices (BufferIndexHelper), rewrites the root's pre-order partial, sends the same pre-order
    destinations and asks for the sums.  On the engine no held list ever has to run (each is replaced by the next one
    unexecuted) and no pre-order partial is written until somebody reads one; the numbers are the oracle's throughout.
    rescale: the post-order pass rescales every node in write mode, every evaluation (new factors every time) — the walk follows
    them (kernels_preorder4.hip: a step into an internal node multiplies by the reciprocal of that node's factor)."""
    wl = helpers.random_workload(33, 500, 4, 4, seed=31)
    g = BranchGradient(wl, double_buffer=True, rescale=rescale)
    o = BranchGradient(wl, double_buffer=True, rescale=rescale, library=oracle_lib)
    rng = np.random.default_rng(3)
    steps = 6
    for step in range(steps):
        scale = np.exp(0.2 * rng.standard_normal(g.N))
        g.branch_lengths *= scale; o.branch_lengths *= scale
        second = step == 3
        rg, ro = g.gradient(second=second), o.gradient(second=second)
        assert helpers.rel_err(rg[0], ro[0]) <= REL_TOL
        close(rg[1], ro[1], "gradient, step %d" % step)
        if second:
            close(rg[2], ro[2], "second derivatives, step %d" % step)
    st = g.b.gradientStats()
    # (the first step, and step 3, which asks for the sums of squares: the list runs with that call — its second derivatives find
    # stored partials; every other step: the walk, with the reciprocals of the post-order pass's scale factors when it rescales)
    assert st == {"fused": 2, "by_operation": 0, "walked": steps - 2, "late": 0}
    for n_ in g.edges[:5]:
        close(g.pre_partials(n_), o.pre_partials(n_), "pre-order partial at the end of the chain")
    g.close(); o.close()


def test_gradient_on_a_deep_ladder_with_rescaling(oracle_lib):
    """A caterpillar of 250 taxa (249 levels), rescaling in the post-order pass: what the pre-order walk carries down the ladder
    is the pre-order partial times the root's 1 / likelihood times the reciprocal of every scale factor on the way.  Second
    and third evaluation (the walk) against the oracle (whose unscaled pre-order partials are still inside the double range at
    this depth — on deeper ladders they are not, while the walk's products shrink and grow together and stay finite); should
    the walk's sums ever come out non-finite the call goes to the path that forms each edge's denominator itself
    (engine_preorder.cpp walkedGradient)."""
    wl = helpers.random_workload(250, 300, 4, 4, seed=8, tree_kind="caterpillar", root_to_tip=0.6)
    g = BranchGradient(wl, rescale=True, double_buffer=True)
    o = BranchGradient(wl, rescale=True, double_buffer=True, library=oracle_lib)
    for step in range(3):
        lg, gg = g.gradient()
        lo, go = o.gradient()
        assert np.isfinite(lo) and np.all(np.isfinite(go)) and helpers.rel_err(lg, lo) <= REL_TOL
        close(gg, go, "ladder, evaluation %d" % step)
    assert g.b.gradientStats() == {"fused": 1, "by_operation": 0, "walked": 2, "late": 0}
    g.close(); o.close()


def test_gradient_corner_shapes(oracle_lib):
    """The pre-order walk (second evaluation of every instance) and the sweep per level (first evaluation) over corner shapes:
    two taxa (a list of one node), a single pattern, ragged pattern counts, 1 .. 16 rate categories (the walk's 256-, 512- and
    1024-thread instantiations), with and without rescaling in the post-order pass."""
    checked = 0
    for C in (1, 3, 5, 8, 11, 16):
        for T, P in ((2, 1), (3, 15), (5, 33), (9, 64), (14, 129), (40, 700)):
            if C > 8 and P > 200:
                continue
            wl = helpers.random_workload(T, P, 4, C, seed=500 + 13 * C + T)
            for rescale in (False, True):
                g = BranchGradient(wl, rescale=rescale, double_buffer=True)
                o = BranchGradient(wl, rescale=rescale, double_buffer=True, library=oracle_lib)
                for step in range(2):
                    (lg, gg), (lo, go) = g.gradient(), o.gradient()
                    assert helpers.rel_err(lg, lo) <= REL_TOL, (C, T, P, rescale, step)
                    close(gg, go, "C=%d T=%d P=%d rescale=%s evaluation %d" % (C, T, P, rescale, step))
                st = g.b.gradientStats()                          # (two taxa: no internal operand whose scale factor could be unknown — walked twice)
                assert st == {"fused": 0 if T == 2 else 1, "by_operation": 0, "walked": 2 if T == 2 else 1, "late": 0}, (C, T, P, rescale)
                g.close(); o.close()
                checked += 1
    assert checked >= 60


def test_gradient_corner_shapes_on_the_matrix_cores(oracle_lib):
    """16..64 states: the one-pass pre-order kernel and the edge-derivative kernel (kernels_mfma.hip k_preOpTiled / k_edgeTiled) over
    corner shapes — partial state tiles (17, 33, 61), exact ones (16, 20, 64), a single pattern, ragged tiles of 32 and blocks of 64,
    two taxa, 1 .. 7 rate categories, with and without rescaling in the post-order pass; sums, sums of squares and per-pattern
    derivatives against the oracle, and against the older two-pass route of the same engine where that still exists."""
    checked = 0
    for S, C in ((16, 3), (17, 1), (20, 4), (33, 2), (61, 3), (64, 1), (20, 7)):
        for T, P in ((2, 1), (3, 31), (5, 32), (6, 65), (9, 130)):
            if S > 33 and P > 65 and C > 1:
                continue
            wl = helpers.random_workload(T, P, S, C, seed=900 + 7 * S + T)
            for rescale in (False, True):
                g = BranchGradient(wl, rescale=rescale, double_buffer=True)
                o = BranchGradient(wl, rescale=rescale, double_buffer=True, library=oracle_lib)
                (lg, gg), (lo, go) = g.gradient(), o.gradient()
                close(lg, lo, "lnL S=%d C=%d T=%d P=%d rescale=%s" % (S, C, T, P, rescale))      # (two taxa, one pattern: lnL ~ -1e-15)
                close(gg, go, "S=%d C=%d T=%d P=%d rescale=%s" % (S, C, T, P, rescale))
                lg, gg, hg, pg = g.gradient(second=True, per_pattern=True)
                lo, go, ho, po = o.gradient(second=True, per_pattern=True)
                close(gg, go, "gradient"); close(hg, ho, "second derivatives"); close(pg, po, "per-pattern derivatives")
                assert g.b.gradientStats()["by_operation"] == 2
                g.close(); o.close()
                checked += 1
    assert checked >= 60


def test_gradient_with_tips_sent_as_partials(oracle_lib):
    """A useAmbiguities-style instance (every tip uploaded with setTipPartials): the walks treat such a tip as a memory operand
    without a scale factor; the second evaluation is answered by the pre-order walk."""
    wl = helpers.random_workload(30, 400, 4, 2, seed=21)
    g = BranchGradient(wl, double_buffer=True)
    o = BranchGradient(wl, double_buffer=True, library=oracle_lib)
    eye = np.vstack([np.eye(4), np.ones((1, 4))])                  # state 4 = missing: all ones
    for t in range(wl.tip_count):
        part = np.ascontiguousarray(eye[wl.tip_states[t]]).ravel()
        for d in (g, o):
            d.b.setTipPartials(t, part)
    for step in range(2):
        (lg, gg), (lo, go) = g.gradient(), o.gradient()
        assert helpers.rel_err(lg, lo) <= REL_TOL
        close(gg, go, "tips as partials, evaluation %d" % step)
    assert g.b.gradientStats() == {"fused": 1, "by_operation": 0, "walked": 1, "late": 0}
    g.close(); o.close()


def test_cumulative_index_in_update_partials_equals_accumulate(oracle_lib):
    """updatePartials' own cumulativeScaleIndex (the factors of the list's rescaling operations folded into that buffer by the
    call itself — one accumulation launch for the whole list, engine_levels.cpp foldCumulative) against the explicit
    resetScaleFactors + accumulateScaleFactors the reference's delegates use, and against the oracle."""
    wl = helpers.random_workload(40, 700, 4, 4, seed=12)
    vals = []
    for lib in (None, oracle_lib):
        g = BranchGradient(wl, rescale=True, library=lib)
        a = g.log_likelihood()                                    # explicit protocol
        idx = np.asarray(g.edges, dtype=np.int32)
        g.b.updateTransitionMatrices(0, g._edge_matrix[g._set], None, None, g.branch_lengths[idx], len(idx))
        g.b.resetScaleFactors(g.cum_scale)
        g.b.updatePartials(g._post_ops, len(g._post_ops) // 7, g.cum_scale)
        out = [0.0]
        g.b.calculateRootLogLikelihoods([g.post_index(wl.tree.root)], [0], [0], [g.cum_scale], 1, out)
        vals.append((a, out[0]))
        g.close()
    assert helpers.rel_err(vals[0][0], vals[0][1]) <= 1e-13 and helpers.rel_err(vals[0][1], vals[1][1]) <= REL_TOL
    assert helpers.rel_err(vals[1][0], vals[1][1]) <= 1e-13


def test_gradient_benchmark_sized_tree(oracle_lib):
    """A 1000-taxon tree (the benchmark's size), 2 000 patterns: the gradient of all 1 998 branches vs the oracle."""
    wl = helpers.random_workload(1000, 2000, 4, 4, seed=4242)
    g = BranchGradient(wl)
    o = BranchGradient(wl, library=oracle_lib)
    lg, gg = g.gradient()
    lo, go = o.gradient()
    assert helpers.rel_err(lg, lo) <= REL_TOL
    close(gg, go, "gradient, 1000 taxa")
    g.close(); o.close()


@pytest.mark.parametrize("S", [20, 61])
def test_gradient_after_set_pattern_partitions(S, oracle_lib):
    """setPatternPartitions re-allocates the matrix block of an instance with virtual buffers; on the T32 layout the identity
    matrix and the transposed-matrix scratch of the two-pass pre-order path sit behind the snapshot slots and have to move
    with it (round-3 advisor finding: they were left behind and the next pre-order pass read and wrote out of bounds)."""
    wl = helpers.random_workload(8, 100, S, 2, seed=77 + S)
    g = BranchGradient(wl)
    o = BranchGradient(wl, library=oracle_lib)
    g.b.setPatternPartitions(1, np.zeros(wl.pattern_count, dtype=np.int32))
    for _ in range(2):
        lo, go = o.gradient()
        lg, gg = g.gradient()
        assert helpers.rel_err(lg, lo) <= REL_TOL
        close(gg, go, "gradient after setPatternPartitions")
    for n in range(g.N):
        if n != wl.tree.root:
            a, b = g.pre_partials(n).reshape(2, wl.pattern_count, S), o.pre_partials(n).reshape(2, wl.pattern_count, S)
            ref = np.max(np.abs(b), axis=(0, 2), keepdims=True)
            assert np.max(np.abs(a - b) / np.maximum(ref, 1e-300)) <= REL_TOL, n


@pytest.mark.parametrize("S,C,T,P,rescale", [(65, 1, 6, 70, False), (100, 2, 5, 45, True), (130, 1, 4, 33, False)])
def test_gradient_above_64_states(S, C, T, P, rescale, oracle_lib):
    """65..255 states (discrete-trait models with many locations: SubstitutionModelCrossProductDelegate.java:153-178 takes whatever
    stateCount the data type has): pre-order partials, edge derivatives (sums, squares, per pattern) and cross products on the
    general kernels' large-state forms (kernels_preorder.hip k_prePartialsBig / k_edgeDifferentialsBig / k_crossProductsBig: the
    matrices read from L2, sixteen patterns' columns in LDS at a time) — round 4 answered -7 here."""
    wl = helpers.random_workload(T, P, S, C, seed=300 + S)
    g = BranchGradient(wl, rescale=rescale)
    o = BranchGradient(wl, rescale=rescale, library=oracle_lib)
    lo, go, ho, po = o.gradient(second=True, per_pattern=True)
    lg, gg, hg, pg = g.gradient(second=True, per_pattern=True)
    assert helpers.rel_err(lg, lo) <= REL_TOL
    close(gg, go, "gradient")
    close(hg, ho, "second derivatives")
    close(pg, po, "per-pattern derivatives")
    close(g.cross_products(), o.cross_products(), "cross products")
    for n in range(g.N):
        if n != wl.tree.root:
            a, b = g.pre_partials(n).reshape(C, P, S), o.pre_partials(n).reshape(C, P, S)
            ref = np.max(np.abs(b), axis=(0, 2), keepdims=True)
            assert np.max(np.abs(a - b) / np.maximum(ref, 1e-300)) <= REL_TOL, n
    # a pre-order operation that rescales (write-scale index): rescaled partials times the factor give the unscaled ones
    if rescale:
        root = wl.tree.root
        ops = g._pre_ops.copy().reshape(-1, 7)
        child = int(ops[0, 0])
        plain = g.b.getPartials(child, bm.beagle.NONE).reshape(C, P, S).copy()
        ops[:, 1] = 0
        g.b.setPartials(g.pre_offset + root, g._root_pre)
        g.b.updatePrePartials(ops[:1].ravel(), 1, bm.beagle.NONE)
        scaled = g.b.getPartials(child, bm.beagle.NONE).reshape(C, P, S)
        f = np.exp(g.b.getLogScaleFactors(0))
        assert np.allclose(scaled * f[None, :, None], plain, rtol=1e-12, atol=0) and np.allclose(scaled.max(axis=(0, 2)), 1.0)
    g.close(); o.close()


@pytest.mark.parametrize("S", [4, 7, 20])
def test_gradient_on_a_partitioned_instance(S, oracle_lib):
    """A partitioned instance (setPatternPartitions with several partitions; the post-order pass issued partition by partition, as
    MultiPartitionDataLikelihoodDelegate does) takes the pre-order / gradient entry points over the whole pattern range — round 4
    answered -7 —, and updatePrePartialsByPartition (9-int tuples, declared by BeagleJNIWrapper) covers one partition's patterns:
    the partitions' lists together leave what the whole-range list leaves."""
    wl = helpers.random_workload(9, 333, S, 2, seed=410 + S)
    P = wl.pattern_count
    parts = np.zeros(P, dtype=np.int32); parts[120:200] = 1; parts[200:] = 2
    g = BranchGradient(wl)
    o = BranchGradient(wl, library=oracle_lib)
    g.b.setPatternPartitions(3, parts)
    NONE = bm.beagle.NONE

    def by_partition(ops7, order=(0, 1, 2)):
        ops7 = ops7.reshape(-1, 7)
        return np.concatenate([np.concatenate([ops7, np.full((len(ops7), 1), k, dtype=np.int32), np.full((len(ops7), 1), NONE, dtype=np.int32)], axis=1)
                               for k in order]).astype(np.int32)

    def partitioned_gradient(second=False, per_pattern=False):
        idx = g._nodes
        g.b.updateTransitionMatrices(0, g._edge_matrix[0], None, None, g.branch_lengths[idx], len(idx))
        ops9 = by_partition(g._post_ops)
        g.b.updatePartialsByPartition(ops9.ravel(), len(ops9))
        byp, tot = np.zeros(3), np.zeros(1)
        root = g.post_index(wl.tree.root)
        g.b.calculateRootLogLikelihoodsByPartition([root] * 3, [0] * 3, [0] * 3, [NONE] * 3, [0, 1, 2], 3, 1, byp, tot)
        g.b.setPartials(g.pre_offset + wl.tree.root, g._root_pre)
        g.b.updatePrePartials(g._pre_ops, len(g._pre_ops) // 7, NONE)            # whole range, on the partitioned instance
        g.b.setDifferentialMatrix(g.q_index, g.infinitesimal(1))
        n = len(g._nodes)
        s1, s1sq, per = g.b.calculateEdgeDifferentials(g._edge_post[0], g._pre_idx, g._q1_idx, g._w0, n, want_per_pattern=per_pattern, want_squared=second)
        grad = np.zeros(g.N); grad[g._nodes] = s1
        return float(tot[0]), grad, s1sq, per

    for _ in range(2):
        lo, go = o.gradient()
        lg, gg, _, _ = partitioned_gradient()
        assert helpers.rel_err(lg, lo) <= REL_TOL
        close(gg, go, "gradient on the partitioned instance")
    lo, go, ho, po = o.gradient(second=True, per_pattern=True)
    lg, gg, sq, pg = partitioned_gradient(second=True, per_pattern=True)
    close(gg, go, "gradient"); close(pg, po, "per-pattern derivatives")
    close(g.cross_products(), o.cross_products(), "cross products")
    whole = {n: g.pre_partials(n).copy() for n in g.edges}
    for n in g.edges:
        close(whole[n], o.pre_partials(n), "pre-order partial %d" % n)
    # the same pre-order pass partition by partition
    g.b.setPartials(g.pre_offset + wl.tree.root, g._root_pre)
    for n in g.edges:                       # wipe the destinations
        g.b.setPartials(g.pre_offset + n, np.zeros(g.C * P * S))
    ops9 = by_partition(g._pre_ops, order=(2, 0, 1))
    # (a parent's operation precedes its children's within a partition; the partitions are independent)
    g.b.updatePrePartialsByPartition(ops9.ravel(), len(ops9))
    for n in g.edges:
        assert np.array_equal(g.pre_partials(n), whole[n]), n
    g.close(); o.close()


def test_add_transition_matrices():
    wl = helpers.random_workload(5, 40, 4, 3, seed=12)
    g = BranchGradient(wl)
    g.log_likelihood()
    a, b = g.b.getTransitionMatrix(0).copy(), g.b.getTransitionMatrix(1).copy()
    g.b.addTransitionMatrices([0, g.q_index], [1, 2], [g.q_index, g.q2_index], 2)      # the second sum reads the first one's result
    assert np.array_equal(g.b.getTransitionMatrix(g.q_index), a + b)
    assert np.array_equal(g.b.getTransitionMatrix(g.q2_index), (a + b) + g.b.getTransitionMatrix(2))
    g.close()


@pytest.mark.parametrize("rescale", [False, True])
@pytest.mark.parametrize("steps", ["1", "2"])
def test_gradient_chain_leaves_short_definitions_unstored(steps, rescale, oracle_lib, monkeypatch):
    """Round 5 (BEAGLE_MI355_GRADIENT_VIRTUAL): the post-order passes of a gradient chain need not store every node.  A node over two
    compact tips (1, the default) — and such a node under one more tip (2) — stay definitions (planner.h stepLimit), and the pre-order
    walk re-evaluates them from the tips where it needs them: the first kind inside its parent's descriptor (kernels.h PW_CHERRY), the
    second by descriptors of its own (PW_POSTOP); a third to a half of the nodes of a coalescent tree are neither written by the one pass
    nor read by the other.  Held here: the stored-node count of the chain's post-order passes, the numbers against the oracle and
    against the same chain with every node stored (0), and that reading partials afterwards — post-order ones of unstored nodes,
    pre-order ones of a list that never ran — still finds the right values."""
    wl = helpers.random_workload(150, 1800, 4, 4, seed=61)
    T = wl.tree.tip_count
    runs = {}
    for name, env in (("virtual", steps), ("stored", "0")):
        monkeypatch.setenv("BEAGLE_MI355_GRADIENT_VIRTUAL", env)
        g = BranchGradient(wl, double_buffer=True, rescale=rescale)
        monkeypatch.delenv("BEAGLE_MI355_GRADIENT_VIRTUAL", raising=False)
        o = BranchGradient(wl, double_buffer=True, rescale=rescale, library=oracle_lib) if name == "virtual" else None
        rng = np.random.default_rng(4)
        out = []
        for step in range(6):
            if step == 2:
                g.b.kernelTimerRestart()                                # (resets the walk's counters; the held list stays held)
            scale = np.exp(0.1 * rng.standard_normal(g.N))
            g.branch_lengths *= scale
            rg = g.gradient()
            out.append(rg)
            if o is not None:
                o.branch_lengths *= scale
                ro = o.gradient()
                assert helpers.rel_err(rg[0], ro[0]) <= REL_TOL
                close(rg[1], ro[1], "gradient, step %d" % step)
        st = g.b.walkStats()
        stored_per_pass = st["stored"] / 4.0
        how = g.b.gradientStats()
        assert how["walked"] == 5 and how["late"] == 0 and how["by_operation"] == 0, how
        if name == "virtual":
            assert stored_per_pass < (0.76 if steps == "1" else 0.72) * (T - 1), stored_per_pass        # (a quarter / a third of the nodes at the very least stays unstored)
            # partials afterwards: an unstored post-order node (materialised on demand), pre-order partials of the held list
            for n_ in list(range(T, g.N))[::11]:
                close(g.post_partials(n_ + (g._set * g._partial_set if n_ >= T else 0)), o.post_partials(n_ + (o._set * o._partial_set if n_ >= T else 0)), "post-order partial %d" % n_)
            for n_ in g.edges[:7]:
                close(g.pre_partials(n_), o.pre_partials(n_), "pre-order partial %d" % n_)
            assert g.b.gradientStats()["late"] == 1
            o.close()
        else:
            assert stored_per_pass == T - 1
        runs[name] = out
        g.close()
    for a, b in zip(runs["virtual"], runs["stored"]):
        assert helpers.rel_err(a[0], b[0]) <= 1e-13
        close(a[1], b[1], "virtual against stored")


def test_gradient_corner_shapes_with_unstored_definitions(oracle_lib, monkeypatch):
    """The corner shapes of test_gradient_corner_shapes (two taxa, single patterns, ragged counts, 1 .. 16 categories) with the
    gradient chain's short definitions left unstored: third evaluation of every instance (the first sets the chain's hint, the second
    plans under the step limit) against the oracle."""
    checked = 0
    for C in (1, 4, 8, 16):
        for T, P in ((2, 1), (3, 15), (6, 33), (14, 129), (40, 700)):
            if C > 8 and P > 200:
                continue
            monkeypatch.setenv("BEAGLE_MI355_GRADIENT_VIRTUAL", "2" if C in (1, 8) else "1")      # (both kinds of unstored operand)
            wl = helpers.random_workload(T, P, 4, C, seed=700 + 13 * C + T)
            for rescale in (False, True):
                g = BranchGradient(wl, rescale=rescale, double_buffer=True)
                o = BranchGradient(wl, rescale=rescale, double_buffer=True, library=oracle_lib)
                for step in range(3):
                    lg, gg = g.gradient()
                    lo, go = o.gradient()
                    assert helpers.rel_err(lg, lo) <= REL_TOL, (C, T, P, rescale, step)
                    close(gg, go, "corner shape C=%d T=%d P=%d rescale=%s step %d" % (C, T, P, rescale, step))
                g.close(); o.close()
                checked += 1
    assert checked >= 30

