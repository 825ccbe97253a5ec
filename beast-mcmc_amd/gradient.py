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

    def __init__(self, workload, library=None, rescale=False, resource_list=(1,), double_buffer=False):
        """double_buffer: lay the post-order partials and the branch matrices out in two sets and alternate between them from
        one evaluation to the next, as BufferIndexHelper does for the reference (BufferIndexHelper.java:65-85 index = i + 0 or
        i + doubleBufferCount; BeagleDataLikelihoodDelegate.java:853 flipOffset per updated node; a gradient chain updates
        every node, so here the whole set flips)."""
        wl = self.wl = workload
        tr = self.tree = wl.tree
        self.T, self.N = tr.tip_count, tr.node_count
        self.S, self.P, self.C = wl.state_count, wl.pattern_count, wl.category_count
        self.rescale = rescale
        self.double_buffer = double_buffer
        internal = self.N - self.T
        # buffer plan: pre-order partial of node n = pre_offset + n, right after the post-order ones
        # (BeagleDataLikelihoodDelegate.java:236-244); the differential matrices right after the branch matrices
        self._partial_set, self._matrix_set = (internal, self.N) if double_buffer else (0, 0)
        self.pre_offset = self.T + (2 if double_buffer else 1) * internal
        matrices = (2 if double_buffer else 1) * self.N
        self.q_index = matrices
        self.q2_index = matrices + 1
        # scale buffers: one per internal node and the cumulative one; with double buffering two such sets, flipped together with
        # the partials (BeagleDataLikelihoodDelegate.java:237 scaleBufferHelper, :874 / :917 flipOffset)
        self._scale_set = internal + 1 if (double_buffer and rescale) else 0
        self.b = _b.Beagle(self.T, self.pre_offset + self.N, self.T, self.S, self.P, 1, matrices + 2, self.C,
                           ((2 if double_buffer else 1) * (internal + 1) if rescale else 0), resourceList=resource_list, library=library)
        for t in range(self.T):
            self.b.setTipStates(t, wl.tip_states[t])
        self.b.setPatternWeights(wl.weights)
        self.b.setStateFrequencies(0, wl.freqs)
        self.b.setCategoryRates(wl.cat_rates)
        self.b.setCategoryWeights(0, wl.cat_weights)
        self.b.setEigenDecomposition(0, wl.eig.evec, wl.eig.ievc, wl.eig.evals)
        self.branch_lengths = np.array([tr.branch_length(n) if n != tr.root else 0.0 for n in range(self.N)])
        self.edges = [n for n in range(self.N) if n != tr.root]
        self._infinitesimal = {}
        self._root_pre = np.tile(wl.freqs, self.P * self.C)                      # simulateRoot, :142-151
        self._set = 0                                    # which of the two sets the current evaluation lives in
        sets = (0, 1) if double_buffer else (0,)
        self._post_ops_by_set = [self._build_post_ops(v) for v in sets]
        self._pre_ops_by_set = [self._build_pre_ops(v) for v in sets]
        e = np.asarray(self.edges, dtype=np.int32)
        self._edge_post = [np.asarray([self.post_index(n, v) for n in self.edges], dtype=np.int32) for v in sets]
        self._edge_matrix = [e + v * self._matrix_set for v in sets]
        self._nodes = e
        self._pre_idx = e + self.pre_offset
        self._q1_idx = np.full(len(e), self.q_index, dtype=np.int32)
        self._q2_idx = np.full(len(e), self.q2_index, dtype=np.int32)
        self._w0 = np.zeros(1, dtype=np.int32)
        self._scale_indices_by_set = [np.asarray([self.scale_index(n, v) for n in range(self.T, self.N)], dtype=np.int32) for v in sets]

    def post_index(self, node, which=None):
        v = self._set if which is None else which
        return node if node < self.T else node + v * self._partial_set

    def scale_index(self, node, which=None):
        """scale buffer of internal node `node` (node = None: the cumulative one)"""
        if not self.rescale:
            return _b.NONE
        v = self._set if which is None else which
        return (self.N - self.T if node is None else node - self.T) + v * self._scale_set

    @property
    def cum_scale(self):
        return self.scale_index(None)

    def matrix_index(self, node, which=None):
        return node + (self._set if which is None else which) * self._matrix_set

    @property
    def _post_ops(self):
        return self._post_ops_by_set[self._set]

    @property
    def _pre_ops(self):
        return self._pre_ops_by_set[self._set]

    # -- op lists ----------------------------------------------------------------------------------------------
    def _build_post_ops(self, v=0):
        tr, ops = self.tree, []
        for n in tr.postorder():
            if n < self.T:
                continue
            l, r = int(tr.left[n]), int(tr.right[n])
            ws = self.scale_index(n, v)
            ops += [self.post_index(n, v), ws, _b.NONE, self.post_index(l, v), self.matrix_index(l, v),
                    self.post_index(r, v), self.matrix_index(r, v)]
        return np.asarray(ops, dtype=np.int32)

    def _build_pre_ops(self, v=0):
        """AbstractBeagleGradientDelegate.java:207-221, in pre-order (TreeTraversal pre-order: parent first)."""
        tr, ops, stack = self.tree, [], [self.tree.root]
        while stack:
            n = stack.pop()
            if n < self.T:
                continue
            l, r = int(tr.left[n]), int(tr.right[n])
            for child, sib in ((l, r), (r, l)):
                ops += [self.pre_offset + child, _b.NONE, _b.NONE, self.pre_offset + n, self.matrix_index(child, v),
                        self.post_index(sib, v), self.matrix_index(sib, v)]
            stack += [r, l]
        return np.asarray(ops, dtype=np.int32)

    # -- evaluation --------------------------------------------------------------------------------------------
    def set_branch_length(self, node, t):
        self.branch_lengths[node] = t

    def log_likelihood(self):
        if self.double_buffer:
            self._set ^= 1
        idx = self._nodes
        self.b.updateTransitionMatrices(0, self._edge_matrix[self._set], None, None, self.branch_lengths[idx], len(idx))
        # BeagleDataLikelihoodDelegate.java:863-917: the operations with NONE as the cumulative index, then (rescaling) the
        # per-node factors reset and accumulated into the cumulative buffer
        self.b.updatePartials(self._post_ops, len(self._post_ops) // 7, _b.NONE)
        if self.rescale:
            self.b.resetScaleFactors(self.cum_scale)
            idx = self._scale_indices_by_set[self._set]
            self.b.accumulateScaleFactors(idx, len(idx), self.cum_scale)
        out = [0.0]
        self.b.calculateRootLogLikelihoods([self.post_index(self.tree.root)], [0], [0], [self.cum_scale], 1, out)
        return out[0]

    def infinitesimal(self, power=1):
        """Q (or Q^2) scaled per category rate (rate^power), category-major — what the delegate caches."""
        if power not in self._infinitesimal:
            e = self.wl.eig
            q = (e.evec * e.evals[None, :]) @ e.ievc
            if power == 2:
                q = q @ q
            self._infinitesimal[power] = np.concatenate([(q * r ** power).ravel() for r in self.wl.cat_rates])
        return self._infinitesimal[power]

    def gradient(self, second=False, per_pattern=False):
        """-> lnL, d lnL/d t_n for every non-root node n (array indexed by node; root entry 0), and optionally the
        diagonal second derivatives (AbstractBeagleBranchGradientDelegate.java:82-95: second - firstSquared)."""
        lnl = self.log_likelihood()
        self.b.setPartials(self.pre_offset + self.tree.root, self._root_pre)
        self.b.updatePrePartials(self._pre_ops, len(self._pre_ops) // 7, _b.NONE)
        self.b.setDifferentialMatrix(self.q_index, self.infinitesimal(1))
        nodes, post, pre = self._nodes, self._edge_post[self._set], self._pre_idx
        n = len(post)
        # (firstSquared is only asked for when the second derivatives are: AbstractBeagleBranchGradientDelegate.java:80-84)
        s1, s1sq, per = self.b.calculateEdgeDifferentials(post, pre, self._q1_idx, self._w0, n, want_per_pattern=per_pattern,
                                                          want_squared=second)
        grad = np.zeros(self.N)
        grad[nodes] = s1
        result = [lnl, grad]
        if second:
            self.b.setDifferentialMatrix(self.q2_index, self.infinitesimal(2))
            s2, _, _ = self.b.calculateEdgeDifferentials(post, pre, self._q2_idx, self._w0, n, want_squared=False)
            hess = np.zeros(self.N)
            hess[nodes] = s2 - s1sq
            result.append(hess)
        if per_pattern:
            result.append(per)
        return tuple(result)

    def cross_products(self):
        """SubstitutionModelCrossProductDelegate.java:149-162 (one substitution model): d lnL / d Q_ij, first-order form.
        Call after gradient() (needs the pre-order partials)."""
        nodes = np.asarray(self.edges, dtype=np.int32)
        return self.b.calculateCrossProductDifferentials(self._edge_post[self._set], nodes + self.pre_offset, [0], [0],
                                                         self.branch_lengths[nodes], len(nodes)).reshape(self.S, self.S)

    def pre_partials(self, node):
        return self.b.getPartials(self.pre_offset + node, _b.NONE)

    def post_partials(self, node):
        return self.b.getPartials(node, _b.NONE)

    def close(self):
        self.b.finalize()
