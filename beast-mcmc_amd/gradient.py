"""Branch-length gradients through pre-order partials: the host side of SURVEY.md 8f row f1.

Mirror of the call sequence the reference issues for a gradient evaluation
(src/dr/evomodel/treedatalikelihood/preorder/AbstractBeagleGradientDelegate.java:115-151 ``simulate``/``simulateRoot``,
:207-221 ``vectorizeNodeOperations``; AbstractBeagleBranchGradientDelegate.java:52-95 ``getNodeDerivatives``;
discrete/DiscreteTraitBranchRateDelegate.java:49-89 the rate-scaled infinitesimal matrix) over the ``beagle.Beagle``
method set, with the reference's buffer plan (BeagleDataLikelihoodDelegate.java:236-244: pre-order partials sit right
after the post-order ones, one per node; HomogenousSubstitutionModelDelegate.java:142-149: the differential matrices sit
right after the branch matrices).  It holds no arithmetic of its own: everything numeric happens in the engine library it
is given — the HIP engine by default, the CPU oracle in tests.
"""
import numpy as np

from . import beagle as _b


class BranchGradient:
    """One full evaluation = post-order partials, root lnL, pre-order partials, d lnL / d(branch length) per node."""

    def __init__(self, workload, library=None, rescale=False, resource_list=(1,)):
        wl = self.wl = workload
        tr = self.tree = wl.tree
        self.T, self.N = tr.tip_count, tr.node_count
        self.S, self.P, self.C = wl.state_count, wl.pattern_count, wl.category_count
        self.rescale = rescale
        # buffer plan (no double buffering: a gradient evaluation recomputes the whole tree)
        self.pre_offset = self.N                       # pre-order partial of node n = pre_offset + n
        self.q_index = self.N                          # differential matrix (first order) after the N branch matrices
        self.q2_index = self.N + 1
        self.cum_scale = self.T - 1 if rescale else _b.NONE
        self.b = _b.Beagle(self.T, 2 * self.N, self.T, self.S, self.P, 1, self.N + 2, self.C,
                           (self.T if rescale else 0), resourceList=resource_list, library=library)
        for t in range(self.T):
            self.b.setTipStates(t, wl.tip_states[t])
        self.b.setPatternWeights(wl.weights)
        self.b.setStateFrequencies(0, wl.freqs)
        self.b.setCategoryRates(wl.cat_rates)
        self.b.setCategoryWeights(0, wl.cat_weights)
        self.b.setEigenDecomposition(0, wl.eig.evec, wl.eig.ievc, wl.eig.evals)
        self.branch_lengths = np.array([tr.branch_length(n) if n != tr.root else 0.0 for n in range(self.N)])
        self._post_ops = self._build_post_ops()
        self._pre_ops = self._build_pre_ops()
        self.edges = [n for n in range(self.N) if n != tr.root]

    # -- op lists ----------------------------------------------------------------------------------------------
    def _build_post_ops(self):
        tr, ops = self.tree, []
        for n in tr.postorder():
            if n < self.T:
                continue
            l, r = int(tr.left[n]), int(tr.right[n])
            ws = (n - self.T) if self.rescale else _b.NONE
            ops += [n, ws, _b.NONE, l, l, r, r]
        return np.asarray(ops, dtype=np.int32)

    def _build_pre_ops(self):
        """AbstractBeagleGradientDelegate.java:207-221, in pre-order (TreeTraversal pre-order: parent first)."""
        tr, ops, stack = self.tree, [], [self.tree.root]
        while stack:
            n = stack.pop()
            if n < self.T:
                continue
            l, r = int(tr.left[n]), int(tr.right[n])
            for child, sib in ((l, r), (r, l)):
                ops += [self.pre_offset + child, _b.NONE, _b.NONE, self.pre_offset + n, child, sib, sib]
            stack += [r, l]
        return np.asarray(ops, dtype=np.int32)

    # -- evaluation --------------------------------------------------------------------------------------------
    def set_branch_length(self, node, t):
        self.branch_lengths[node] = t

    def log_likelihood(self):
        idx = np.asarray(self.edges, dtype=np.int32)
        self.b.updateTransitionMatrices(0, idx, None, None, self.branch_lengths[idx], len(idx))
        if self.rescale:
            self.b.resetScaleFactors(self.cum_scale)
        self.b.updatePartials(self._post_ops, len(self._post_ops) // 7, self.cum_scale)
        out = [0.0]
        self.b.calculateRootLogLikelihoods([self.tree.root], [0], [0], [self.cum_scale], 1, out)
        return out[0]

    def infinitesimal(self, power=1):
        """Q (or Q^2) scaled per category rate (rate^power), category-major — what the delegate caches."""
        e = self.wl.eig
        q = (e.evec * e.evals[None, :]) @ e.ievc
        if power == 2:
            q = q @ q
        return np.concatenate([(q * r ** power).ravel() for r in self.wl.cat_rates])

    def gradient(self, second=False, per_pattern=False):
        """-> lnL, d lnL/d t_n for every non-root node n (array indexed by node; root entry 0), and optionally the
        diagonal second derivatives (AbstractBeagleBranchGradientDelegate.java:82-95: second - firstSquared)."""
        lnl = self.log_likelihood()
        root_pre = np.tile(self.wl.freqs, self.P * self.C)                       # simulateRoot, :142-151
        self.b.setPartials(self.pre_offset + self.tree.root, root_pre)
        self.b.updatePrePartials(self._pre_ops, len(self._pre_ops) // 7, _b.NONE)
        self.b.setDifferentialMatrix(self.q_index, self.infinitesimal(1))
        post = np.asarray(self.edges, dtype=np.int32)
        pre = post + self.pre_offset
        n = len(post)
        s1, s1sq, per = self.b.calculateEdgeDifferentials(post, pre, [self.q_index] * n, [0], n, want_per_pattern=per_pattern)
        grad = np.zeros(self.N)
        grad[post] = s1
        result = [lnl, grad]
        if second:
            self.b.setDifferentialMatrix(self.q2_index, self.infinitesimal(2))
            s2, _, _ = self.b.calculateEdgeDifferentials(post, pre, [self.q2_index] * n, [0], n)
            hess = np.zeros(self.N)
            hess[post] = s2 - s1sq
            result.append(hess)
        if per_pattern:
            result.append(per)
        return tuple(result)

    def cross_products(self):
        """SubstitutionModelCrossProductDelegate.java:149-162 (one substitution model): d lnL / d Q_ij, first-order form.
        Call after gradient() (needs the pre-order partials)."""
        post = np.asarray(self.edges, dtype=np.int32)
        return self.b.calculateCrossProductDifferentials(post, post + self.pre_offset, [0], [0], self.branch_lengths[post],
                                                         len(post)).reshape(self.S, self.S)

    def pre_partials(self, node):
        return self.b.getPartials(self.pre_offset + node, _b.NONE)

    def post_partials(self, node):
        return self.b.getPartials(node, _b.NONE)

    def close(self):
        self.b.finalize()
