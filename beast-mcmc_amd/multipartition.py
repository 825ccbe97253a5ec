"""Host-side mirror of the reference's multi-partition caller for ONE engine instance.

What ``MultiPartitionDataLikelihoodDelegate`` does per evaluation, call for call
(src/dr/evomodel/treedatalikelihood/MultiPartitionDataLikelihoodDelegate.java): partitions are contiguous pattern
ranges of one instance (:520-553 ``setPatternPartitions``); every partition has its own eigen system, category rates,
weights and frequencies (:835 ``setCategoryRatesWithIndex``); all branch matrices of all partitions go out in ONE
``updateTransitionMatricesWithMultipleModels`` (:880-887); the operation list carries 9-int tuples
{dest, writeScale, readScale, child1, matrix1, child2, matrix2, partition, cumulativeScale} (:972-997) through
``updatePartialsByPartition``; scale factors are accumulated per partition (:1016-1017) and the root is integrated with
``calculateRootLogLikelihoodsByPartition`` (:1074-1083).  Partials and matrix indices flip between two buffers per node
(``BufferIndexHelper``), the traversal is the reverse level order of ``LikelihoodTreeTraversal.java:133-205``.

Python because it only sequences calls; the arithmetic is in the engine behind ``beagle.Beagle``.
"""
import numpy as np

from . import beagle as _b

NONE = _b.NONE


_MP_EVALUATION = None


def _mp_evaluation_type():
    """ctypes image of tools/host/tree_likelihood.cpp MpEvaluation (one type per process: the host function's argtypes name it)."""
    global _MP_EVALUATION
    if _MP_EVALUATION is None:
        import ctypes as C
        DP, IP = C.POINTER(C.c_double), C.POINTER(C.c_int)
        DPP = C.POINTER(DP)

        class MpEvaluation(C.Structure):
            _fields_ = [(n, C.c_int) for n in ("K", "always_rescale", "n_matrices", "n_ops9", "n_ops7", "n_scale", "cum", "n_branches")] + \
                       [("update_substitution_model", IP), ("update_site_rate_model", IP),
                        ("U", DPP), ("Uinv", DPP), ("lambda_", DPP), ("rates", DPP), ("weights", DPP), ("freqs", DPP),
                        ("eig_idx", IP), ("mat_idx", IP), ("lens0", DP), ("branch", IP), ("lens", DP),
                        ("ops9", IP), ("ops7", IP), ("scale_idx", IP), ("roots", IP), ("range", IP), ("cum_idx", IP),
                        ("by_part", DP), ("total", DP)]
        _MP_EVALUATION = MpEvaluation
    return _MP_EVALUATION


class MultiPartitionTreeLikelihood:
    def __init__(self, pw, *, library=None, resource_list=(1,), always_rescale=False, native_sequence=True):
        self.pw = pw
        # the per-evaluation call sequence of calculate() in C++ (tools/host/tree_likelihood.cpp mpEvaluate: the same calls in the same
        # order; the reference's caller is compiled code) — False: call by call from here (what a test that watches the calls wants)
        self.native_sequence = native_sequence
        self.tree = tree = pw.tree
        self.T = T = tree.tip_count
        self.K = K = len(pw.parts)
        self.nodes = nodes = tree.node_count
        self.C = len(pw.parts[0].cat_rates)
        self.P = pw.pattern_count
        self.always_rescale = always_rescale
        self.flip = np.zeros(nodes, dtype=np.int32)          # BufferIndexHelper offsets of the internal nodes
        self.mflip = 0
        self.mf = np.zeros(nodes, dtype=np.int32)            # ... and of the branch matrices (one offset per node, all partitions)
        self._saved = None
        self.evaluations = 0
        # partials: tips 0..T-1, internal node n -> T + 2 (n - T) + flip; matrices: (partition, node, flip); scale: node + cumulative
        self.b = _b.Beagle(T, T + 2 * (T - 1), T, pw.parts[0].state_count, self.P, K, 2 * K * nodes, self.C, (T - 1) + 1,
                           resourceList=list(resource_list), library=library)
        for t in range(T):
            self.b.setTipStates(t, np.concatenate([w.tip_states[t] for w in pw.parts]))
        self.b.setPatternWeights(np.concatenate([w.weights for w in pw.parts]))
        if K > 1:
            self.b.setPatternPartitions(K, np.concatenate([np.full(w.pattern_count, k, dtype=np.int32) for k, w in enumerate(pw.parts)]))
        # reverse level order: deepest internal nodes first (children always before parents)
        depth = np.zeros(nodes, dtype=np.int64)
        for n in reversed(tree.postorder()):
            if n != tree.root:
                depth[n] = depth[tree.parent[n]] + 1
        self.order = [int(n) for n in sorted((n for n in range(T, nodes)), key=lambda n: -depth[n])]
        self.branch_rates = np.ones(nodes)
        # the reference's model flags (MultiPartitionDataLikelihoodDelegate.java:579-583, :621-647): set at construction, by a model
        # change (make_dirty / set_substitution_model / set_site_model) — cleared when an evaluation is through (:1116-1117)
        self.update_substitution_models = [True] * K
        self.update_site_rate_models = [True] * K

    def make_dirty(self):
        """The reference's makeDirty (:1229-1234): every model goes out again with the next evaluation."""
        self.update_substitution_models = [True] * self.K
        self.update_site_rate_models = [True] * self.K

    def set_substitution_model(self, k, eig=None, freqs=None):
        """Partition k's substitution model changed (its eigen system and / or root frequencies): the reference's
        modelChangedEvent -> updateSubstitutionModels (:626-631)."""
        w = self.pw.parts[k]
        if eig is not None:
            w.eig = eig
        if freqs is not None:
            w.freqs = np.asarray(freqs, dtype=np.float64)
        self.update_substitution_models[k] = True
        if hasattr(self, "_fast"):
            del self._fast

    def set_site_model(self, k, cat_rates=None, cat_weights=None):
        """Partition k's site-rate model changed (:640-645)."""
        w = self.pw.parts[k]
        if cat_rates is not None:
            w.cat_rates = np.asarray(cat_rates, dtype=np.float64)
        if cat_weights is not None:
            w.cat_weights = np.asarray(cat_weights, dtype=np.float64)
        self.update_site_rate_models[k] = True
        if hasattr(self, "_fast"):
            del self._fast

    def pbuf(self, n):
        return n if n < self.T else self.T + 2 * (n - self.T) + int(self.flip[n])

    def mbuf(self, k, n):
        return (k * self.nodes + n) * 2 + int(self.mf[n])

    def set_branch_rates(self, rates):
        self.branch_rates = np.asarray(rates, dtype=np.float64)

    def _static_tables(self):
        """Index tables of both buffer flips, built once: the per-evaluation work of the host is then a handful of array
        copies (the reference rebuilds its operation arrays in Java at a comparable cost per node)."""
        tree, K, T, nodes = self.tree, self.K, self.T, self.nodes
        branch = np.array([n for n in range(nodes) if n != tree.root], dtype=np.int64)
        self._branch = branch
        self._lens0 = np.array([tree.branch_length(int(n)) for n in branch])
        self._eig_idx = np.repeat(np.arange(K, dtype=np.int32), len(branch))
        self._mat_idx = [np.concatenate([(k * nodes + branch) * 2 + f for k in range(K)]).astype(np.int32) for f in (0, 1)]
        order = np.array(self.order, dtype=np.int64)
        left = np.array([int(tree.left[n]) for n in order]); right = np.array([int(tree.right[n]) for n in order])
        self._scale_idx = (order - T).astype(np.int32)
        self._ops = {}
        for pf in (0, 1):                               # partials flip of the internal nodes (all flip together here)
            pb = lambda x: np.where(x < T, x, T + 2 * (x - T) + pf)
            for mf in (0, 1):
                ops = np.empty((len(order), K, 9), dtype=np.int32)
                for k in range(K):
                    ops[:, k, 0] = pb(order)
                    ops[:, k, 1] = (order - T) if self.always_rescale else NONE
                    ops[:, k, 2] = NONE
                    ops[:, k, 3] = pb(left); ops[:, k, 4] = (k * nodes + left) * 2 + mf
                    ops[:, k, 5] = pb(right); ops[:, k, 6] = (k * nodes + right) * 2 + mf
                    ops[:, k, 7] = k; ops[:, k, 8] = NONE
                self._ops[(pf, mf)] = (np.ascontiguousarray(ops.reshape(-1)), np.ascontiguousarray(ops[:, 0, :7].reshape(-1)))

    def _fast_tables(self):
        """Everything calculate() hands the engine as ready-made C pointers, built once: the per-evaluation host work is then the
        call sequence itself (the ctypes conversions of ~25 calls cost more than the evaluation's kernels, 100 of 190 us)."""
        import ctypes as C
        DP, IP = C.POINTER(C.c_double), C.POINTER(C.c_int)
        d = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        i = lambda a: np.ascontiguousarray(a, dtype=np.int32)
        keep = []

        def dp(a):
            a = d(a); keep.append(a); return a.ctypes.data_as(DP)

        def ip(a):
            a = i(a); keep.append(a); return a.ctypes.data_as(IP)
        K, T = self.K, self.T
        f = {}
        f["eig"] = [(dp(w.eig.evec), dp(w.eig.ievc), dp(w.eig.evals)) for w in self.pw.parts]
        f["rates"] = [dp(w.cat_rates) for w in self.pw.parts]
        f["weights"] = [dp(w.cat_weights) for w in self.pw.parts]
        f["freqs"] = [dp(w.freqs) for w in self.pw.parts]
        f["eig_idx"] = ip(self._eig_idx)
        f["mat_idx"] = [ip(m) for m in self._mat_idx]
        self._lens = np.empty(len(self._eig_idx))
        self._lens1 = np.empty(len(self._branch))
        f["lens"] = self._lens.ctypes.data_as(DP)
        f["ops"] = {key: (ip(v[0]), len(v[0]) // 9, ip(v[1]), len(v[1]) // 7) for key, v in self._ops.items()}
        f["scale_idx"] = ip(self._scale_idx)
        f["roots"] = {pf: ip([self.T + 2 * (self.tree.root - T) + pf] * K) for pf in (0, 1)}
        f["range"] = ip(list(range(K)))
        f["cum"] = {True: ip([T - 1] * K), False: ip([NONE] * K)}
        self._by_part = np.zeros(K)
        self._total = np.zeros(1)
        f["by_part"] = self._by_part.ctypes.data_as(DP)
        f["total"] = self._total.ctypes.data_as(DP)
        f["keep"] = keep
        self._fast = f
        # ... and the record the native call sequence reads (fields that change per evaluation are set in calculate())
        f["native"] = None
        eng = self.b.lib
        if self.native_sequence and getattr(eng, "partition_api_table", None):
            from .treelikelihood import host_library
            host = host_library()
            DPP = C.POINTER(DP)
            MpEvaluation = _mp_evaluation_type()

            def dpp(ptrs):
                arr = (DP * K)(*ptrs); keep.append(arr); return C.cast(arr, DPP)
            e = MpEvaluation()
            e.K, e.always_rescale, e.n_matrices, e.n_scale = K, int(self.always_rescale), len(self._eig_idx), len(self._scale_idx)
            e.cum, e.n_branches = (T - 1) if self.always_rescale else NONE, len(self._branch)
            self._flag_subst = np.zeros(K, dtype=np.int32); self._flag_site = np.zeros(K, dtype=np.int32)
            e.update_substitution_model = self._flag_subst.ctypes.data_as(IP); e.update_site_rate_model = self._flag_site.ctypes.data_as(IP)
            e.U, e.Uinv, e.lambda_ = dpp([x[0] for x in f["eig"]]), dpp([x[1] for x in f["eig"]]), dpp([x[2] for x in f["eig"]])
            e.rates, e.weights, e.freqs = dpp(f["rates"]), dpp(f["weights"]), dpp(f["freqs"])
            e.eig_idx = f["eig_idx"]
            self._branch32 = i(self._branch); keep.append(self._branch32)
            e.branch = self._branch32.ctypes.data_as(IP)
            e.lens = f["lens"]
            e.scale_idx, e.range, e.cum_idx = f["scale_idx"], f["range"], f["cum"][self.always_rescale]
            e.by_part, e.total = f["by_part"], f["total"]
            host.mpEvaluate.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(MpEvaluation), DP, IP]
            host.mpEvaluate.restype = C.c_int
            self._failed = C.c_int(0)
            f["native"] = (host.mpEvaluate, e, eng.api_table, eng.partition_api_table)

    def calculate(self):
        """One full evaluation; returns (per-partition log-likelihoods, total)."""
        b, tree, K, T = self.b, self.tree, self.K, self.T
        if not hasattr(self, "_ops"):
            self._static_tables()
        if not hasattr(self, "_fast"):
            self._fast_tables()
        import ctypes as C
        f, fn, h, chk = self._fast, b._f, b.instance, b._check
        self.mflip ^= 1
        self.mf[:] = self.mflip                              # (a full evaluation rewrites every buffer: all offsets flip together)
        pf = int(self.flip[T]) ^ 1
        self.flip[T:] = pf
        self._saved = None
        if getattr(self, "_lens_stale", False):                # node heights moved since the tables were built
            self._lens0 = np.array([tree.branch_length(int(n)) for n in self._branch])
            self._lens_stale = False
        if f["native"] is not None:
            call, e, api, papi = f["native"]
            self._flag_subst[:] = self.update_substitution_models; self._flag_site[:] = self.update_site_rate_models
            ops9, n9, ops7, n7 = f["ops"][(pf, self.mflip)]
            e.mat_idx, e.ops9, e.n_ops9, e.ops7, e.n_ops7, e.roots = f["mat_idx"][self.mflip], ops9, n9, ops7, n7, f["roots"][pf]
            e.lens0 = self._lens0.ctypes.data_as(C.POINTER(C.c_double))
            rates = np.ascontiguousarray(self.branch_rates, dtype=np.float64)
            rc = call(api, papi, h, C.byref(e), rates.ctypes.data_as(C.POINTER(C.c_double)), C.byref(self._failed))
            if self._failed.value:
                chk(("", "setEigenDecomposition", "setCategoryRatesWithIndex", "updateTransitionMatricesWithMultipleModels", "updatePartials(ByPartition)",
                     "resetScaleFactors(ByPartition)", "accumulateScaleFactors(ByPartition)", "setCategoryWeights", "setStateFrequencies",
                     "calculateRootLogLikelihoods(ByPartition)")[self._failed.value], rc)
            self.evaluations += 1
            self.update_substitution_models = [False] * K        # (:1116-1117)
            self.update_site_rate_models = [False] * K
            if K > 1:
                return self._by_part.copy(), float(self._total[0])
            return np.array([float(self._total[0])]), float(self._total[0])
        for k in range(K):                                   # (:800-840: only the models flagged as changed go out)
            if self.update_substitution_models[k]:
                u, ui, lam = f["eig"][k]
                chk("setEigenDecomposition", fn["SetEigenDecomposition"](h, k, u, ui, lam))
            if self.update_site_rate_models[k]:
                chk("setCategoryRatesWithIndex", fn["SetCategoryRatesWithIndex"](h, k, f["rates"][k]))
        np.multiply(self._lens0, self.branch_rates[self._branch], out=self._lens1)
        self._lens.reshape(K, -1)[:] = self._lens1
        n = len(self._lens)
        chk("updateTransitionMatricesWithMultipleModels",
            fn["UpdateTransitionMatricesWithMultipleModels"](h, f["eig_idx"], f["eig_idx"], f["mat_idx"][self.mflip], None, None, f["lens"], n))
        ops9, n9, ops7, n7 = f["ops"][(pf, self.mflip)]
        if K > 1:
            chk("updatePartialsByPartition", fn["UpdatePartialsByPartition"](h, ops9, n9))
        else:
            chk("updatePartials", fn["UpdatePartials"](h, ops7, n7, NONE))
        cum = (T - 1) if self.always_rescale else NONE
        if self.always_rescale:
            ns = len(self._scale_idx)
            for k in range(K):
                if K > 1:
                    chk("resetScaleFactorsByPartition", fn["ResetScaleFactorsByPartition"](h, cum, k))
                    chk("accumulateScaleFactorsByPartition", fn["AccumulateScaleFactorsByPartition"](h, f["scale_idx"], ns, cum, k))
                else:
                    chk("resetScaleFactors", fn["ResetScaleFactors"](h, cum))
                    chk("accumulateScaleFactors", fn["AccumulateScaleFactors"](h, f["scale_idx"], ns, cum))
        for k in range(K):                                   # (:1027-1035: weights and root frequencies every time)
            chk("setCategoryWeights", fn["SetCategoryWeights"](h, k, f["weights"][k]))
            chk("setStateFrequencies", fn["SetStateFrequencies"](h, k, f["freqs"][k]))
        self.evaluations += 1
        self.update_substitution_models = [False] * K        # (:1116-1117)
        self.update_site_rate_models = [False] * K
        if K > 1:
            rc = fn["CalculateRootLogLikelihoodsByPartition"](h, f["roots"][pf], f["range"], f["range"], f["cum"][self.always_rescale], f["range"], K, 1,
                                                              f["by_part"], f["total"])
            if rc not in (0, -8):
                chk("calculateRootLogLikelihoodsByPartition", rc)
            return self._by_part.copy(), float(self._total[0])
        out = [0.0]
        b.calculateRootLogLikelihoods([self.pbuf(tree.root)], [0], [0], [cum], 1, out)
        return np.array(out), out[0]

    def move_node_height(self, node, height):
        """ONE node height changes (the move a chain makes most): the branch matrices of the node's two children and of the node
        itself are recomputed for every partition in one updateTransitionMatricesWithMultipleModels, then the operations of the
        node and of its ancestors up to the root (one 9-int tuple per node and partition), then the root integration — what
        MultiPartitionDataLikelihoodDelegate.calculateLikelihood issues when the tree flags one node (:835-1083; only the
        flagged buffers flip, BufferIndexHelper.flipOffset).  ``restore_move()`` takes the move back: offsets only, no engine call.
        Returns (per-partition log-likelihoods, total)."""
        import ctypes as C
        b, tree, K, T, nodes = self.b, self.tree, self.K, self.T, self.nodes
        if not hasattr(self, "_fast"):
            raise RuntimeError("move_node_height: a full calculate() comes first")
        if self.always_rescale:
            raise NotImplementedError("move_node_height: without per-node scale buffers only (always_rescale=False)")
        self._lens_stale = True
        f, fn, h, chk = self._fast, b._f, b.instance, b._check
        node = int(node)
        if node < T or node >= len(tree.left) or tree.left[node] < 0 or tree.right[node] < 0:
            raise ValueError("move_node_height: node %d is not an internal node (tips keep height 0)" % node)
        branches = [int(tree.left[node]), int(tree.right[node])] + ([node] if node != tree.root else [])
        path = []
        n = node
        while n >= 0:
            path.append(n)
            n = int(tree.parent[n])
        self._saved = (node, float(tree.height[node]), branches, path)
        tree.height[node] = height
        mf, flip = self.mf, self.flip
        for x in branches:
            mf[x] ^= 1
        for x in path:
            flip[x] ^= 1
        nb = len(branches)
        lens = [tree.branch_length(x) * self.branch_rates[x] for x in branches] * K
        eig = [k for k in range(K) for _ in range(nb)]
        mats = [(k * nodes + x) * 2 + int(mf[x]) for k in range(K) for x in branches]
        IA, DA = C.c_int * (K * nb), C.c_double * (K * nb)
        eig_c = IA(*eig)
        chk("updateTransitionMatricesWithMultipleModels",
            fn["UpdateTransitionMatricesWithMultipleModels"](h, eig_c, eig_c, IA(*mats), None, None, DA(*lens), K * nb))
        ops = []
        for x in path:
            l, r = int(tree.left[x]), int(tree.right[x])
            pl = l if l < T else T + 2 * (l - T) + int(flip[l])
            pr = r if r < T else T + 2 * (r - T) + int(flip[r])
            dest = T + 2 * (x - T) + int(flip[x])
            for k in range(K):
                ops += [dest, NONE, NONE, pl, (k * nodes + l) * 2 + int(mf[l]), pr, (k * nodes + r) * 2 + int(mf[r]), k, NONE]
        if K > 1:
            chk("updatePartialsByPartition", fn["UpdatePartialsByPartition"](h, (C.c_int * len(ops))(*ops), len(ops) // 9))
        else:
            ops7 = [v for i in range(0, len(ops), 9) for v in ops[i:i + 7]]
            chk("updatePartials", fn["UpdatePartials"](h, (C.c_int * len(ops7))(*ops7), len(ops7) // 7, NONE))
        self.evaluations += 1
        root = T + 2 * (tree.root - T) + int(flip[tree.root])
        if K > 1:
            rc = fn["CalculateRootLogLikelihoodsByPartition"](h, (C.c_int * K)(*([root] * K)), f["range"], f["range"], f["cum"][False], f["range"], K, 1,
                                                              f["by_part"], f["total"])
            if rc not in (0, -8):
                chk("calculateRootLogLikelihoodsByPartition", rc)
            return self._by_part.copy(), float(self._total[0])
        out = [0.0]
        b.calculateRootLogLikelihoods([root], [0], [0], [NONE], 1, out)
        return np.array(out), out[0]

    def restore_move(self):
        """restoreState after a rejected move_node_height: the offsets flip back and the height returns; the engine's buffers of
        the accepted state were never overwritten."""
        if self._saved is None:
            raise RuntimeError("restore_move: nothing to restore")
        node, height, branches, path = self._saved
        self.tree.height[node] = height
        for x in branches:
            self.mf[x] ^= 1
        for x in path:
            self.flip[x] ^= 1
        self._saved = None

    def getSiteLogLikelihoods(self):
        return self.b.getSiteLogLikelihoods()

    def close(self):
        if self.b is not None:
            self.b.finalize()
            self.b = None
