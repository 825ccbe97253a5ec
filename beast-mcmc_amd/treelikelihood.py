"""ctypes handle on the C++ host driver (tools/host/tree_likelihood.cpp) — the mirror of
``dr.evomodel.treelikelihood.BeagleTreeLikelihood`` (src/dr/evomodel/treelikelihood/BeagleTreeLikelihood.java)
plus the level-order traversal of ``dr.evomodel.treedatalikelihood.LikelihoodTreeTraversal``.

All per-evaluation host work (dirty-flag traversal, buffer-index flipping, op-list building, the
rescaling policy and the underflow retry) happens in C++; this class only forwards.
"""
import ctypes as C
import os

import numpy as np

from . import beagle as _b

_DP = C.POINTER(C.c_double)
_IP = C.POINTER(C.c_int)

RESCALE_NONE, RESCALE_ALWAYS, RESCALE_DYNAMIC, RESCALE_DELAYED = 0, 1, 2, 3
POST_ORDER, REVERSE_LEVEL_ORDER = 0, 1

_host = None


def host_library():
    global _host
    if _host is None:
        if not os.path.exists(_b.HOST_LIB):
            raise OSError("host driver %s is missing: run __graft_entry__.build()" % _b.HOST_LIB)
        lib = C.CDLL(_b.HOST_LIB, mode=C.RTLD_LOCAL)
        lib.btlCreate.argtypes = [C.c_void_p] + [C.c_int] * 7 + [_IP, C.c_int, C.c_long, C.c_long]
        lib.btlCreate.restype = C.c_void_p
        lib.btlDestroy.argtypes = [C.c_void_p]
        lib.btlDestroy.restype = None
        lib.btlInstance.argtypes = [C.c_void_p]
        lib.btlSetTree.argtypes = [C.c_void_p, _IP, _IP, _DP, C.c_int]
        lib.btlSetTipStates.argtypes = [C.c_void_p, C.c_int, _IP]
        lib.btlSetTipPartials.argtypes = [C.c_void_p, C.c_int, _DP]
        lib.btlSetPatternWeights.argtypes = [C.c_void_p, _DP]
        lib.btlSetSubstitutionModel.argtypes = [C.c_void_p, _DP, _DP, _DP, _DP]
        lib.btlSetSiteModel.argtypes = [C.c_void_p, _DP, _DP]
        lib.btlSetBranchRates.argtypes = [C.c_void_p, _DP]
        lib.btlSetNodeHeight.argtypes = [C.c_void_p, C.c_int, C.c_double]
        lib.btlRestoreNodeHeight.argtypes = [C.c_void_p, C.c_int, C.c_double]
        lib.btlMakeDirty.argtypes = [C.c_void_p]
        lib.btlSetRescalingFrequency.argtypes = [C.c_void_p, C.c_int]
        lib.btlGetLogLikelihood.argtypes = [C.c_void_p]
        lib.btlGetLogLikelihood.restype = C.c_double
        lib.btlPrepare.argtypes = [C.c_void_p]
        lib.btlAttemptDevice.argtypes = [C.c_void_p, C.c_void_p]
        lib.btlAttemptHost.argtypes = [C.c_void_p, _DP]
        lib.btlFinish.argtypes = [C.c_void_p, C.c_double]
        lib.btlStoreState.argtypes = [C.c_void_p]
        lib.btlRestoreState.argtypes = [C.c_void_p]
        lib.btlGetSiteLogLikelihoods.argtypes = [C.c_void_p, _DP]
        lib.btlLastError.argtypes = [C.c_void_p]
        lib.btlRootBufferIndex.argtypes = [C.c_void_p]
        lib.btlNodeBufferIndex.argtypes = [C.c_void_p, C.c_int]
        lib.btlNodeScaleIndex.argtypes = [C.c_void_p, C.c_int]
        lib.btlCumulativeScaleIndex.argtypes = [C.c_void_p]
        lib.btlCounters.argtypes = [C.c_void_p, C.POINTER(C.c_long)]
        lib.btlLastOperations.argtypes = [C.c_void_p, _IP, C.c_int]
        lib.btlTimings.argtypes = [C.c_void_p, _DP]
        lib.btlSetEngineCollective.argtypes = [C.c_void_p, C.c_int]
        _host = lib
    return _host


def _d(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i(a):
    return np.ascontiguousarray(a, dtype=np.int32)


class BeagleTreeLikelihood:
    """``BeagleTreeLikelihood(tree, patterns, model, site rates, ...)`` over any engine library.

    ``library``: a ``beagle.EngineLibrary``; default = the HIP engine.  Tests pass the CPU oracle's
    library to run the identical call sequence on the checker.
    """

    def __init__(self, workload=None, *, tree=None, tip_states=None, weights=None, eig=None, freqs=None,
                 cat_rates=None, cat_weights=None, state_count=None,
                 rescaling=RESCALE_DYNAMIC, delay_rescaling=True, traversal=REVERSE_LEVEL_ORDER,
                 resource_list=(1,), preference_flags=0, requirement_flags=0, library=None):
        if workload is not None:
            tree, tip_states, weights = workload.tree, workload.tip_states, workload.weights
            eig, freqs = workload.eig, workload.freqs
            cat_rates, cat_weights, state_count = workload.cat_rates, workload.cat_weights, workload.state_count
        self.engine = library or _b.engine()
        self.h = host_library()
        self.tree = tree
        self.tip_count = tree.tip_count
        self.node_count = tree.node_count
        self.state_count = int(state_count)
        tip_states = np.asarray(tip_states)
        self.pattern_count = int(tip_states.shape[1])
        self.category_count = len(cat_rates)
        rl = _i(list(resource_list))
        self.ptr = self.h.btlCreate(self.engine.api_table, self.tip_count, self.state_count, self.pattern_count,
                                    self.category_count, rescaling, int(delay_rescaling), traversal,
                                    rl.ctypes.data_as(_IP), len(rl), preference_flags, requirement_flags)
        if not self.ptr:
            raise _b.BeagleException("create", -6)
        self.instance = self.h.btlInstance(self.ptr)
        self._chk(self.h.btlSetTree(self.ptr, _i(tree.left).ctypes.data_as(_IP), _i(tree.right).ctypes.data_as(_IP),
                                    _d(tree.height).ctypes.data_as(_DP), tree.root), "setTree")
        for t in range(self.tip_count):
            row = _i(tip_states[t])
            self._chk(self.h.btlSetTipStates(self.ptr, t, row.ctypes.data_as(_IP)), "setTipStates")
        w = _d(weights)
        self._chk(self.h.btlSetPatternWeights(self.ptr, w.ctypes.data_as(_DP)), "setPatternWeights")
        self.set_substitution_model(eig, freqs)
        self.set_site_model(cat_rates, cat_weights)

    def _chk(self, rc, what):
        if rc != 0:
            raise _b.BeagleException(what, rc)

    def set_substitution_model(self, eig, freqs):
        u, ui, lam, f = _d(eig.evec), _d(eig.ievc), _d(eig.evals), _d(freqs)
        self._chk(self.h.btlSetSubstitutionModel(self.ptr, u.ctypes.data_as(_DP), ui.ctypes.data_as(_DP),
                                                 lam.ctypes.data_as(_DP), f.ctypes.data_as(_DP)), "setSubstitutionModel")

    def set_site_model(self, rates, weights):
        r, w = _d(rates), _d(weights)
        self._chk(self.h.btlSetSiteModel(self.ptr, r.ctypes.data_as(_DP), w.ctypes.data_as(_DP)), "setSiteModel")

    def set_branch_rates(self, rate_per_node):
        r = _d(rate_per_node)
        self._chk(self.h.btlSetBranchRates(self.ptr, r.ctypes.data_as(_DP)), "setBranchRates")

    def set_node_height(self, node, height):
        self._chk(self.h.btlSetNodeHeight(self.ptr, node, height), "setNodeHeight")

    def restore_node_height(self, node, height):
        """The tree model's restore after a rejected height move: the height goes back, nothing becomes dirty."""
        self._chk(self.h.btlRestoreNodeHeight(self.ptr, node, height), "restoreNodeHeight")

    def set_rescaling_frequency(self, f):
        self.h.btlSetRescalingFrequency(self.ptr, f)

    def makeDirty(self):
        self.h.btlMakeDirty(self.ptr)

    def set_engine_collective(self, on=True):
        """getLogLikelihood returns the sum over all ranks of the instance's communicator (sharding.py, collective="engine")."""
        self.h.btlSetEngineCollective(self.ptr, int(bool(on)))

    # Model parameters as ready-made C pointers: a chain proposes from a handful of parameter blocks, and converting six
    # numpy arrays per evaluation costs more host time (27 us) than the engine's own updatePartials call
    def model_handle(self, eig, freqs, rates, weights):
        arrs = [_d(eig.evec), _d(eig.ievc), _d(eig.evals), _d(freqs), _d(rates), _d(weights)]
        return (arrs, [a.ctypes.data_as(_DP) for a in arrs])

    def apply_model(self, handle):
        p = handle[1]
        self.h.btlSetSubstitutionModel(self.ptr, p[0], p[1], p[2], p[3])
        self.h.btlSetSiteModel(self.ptr, p[4], p[5])

    def getLogLikelihood(self):
        v = self.h.btlGetLogLikelihood(self.ptr)
        err = self.h.btlLastError(self.ptr)
        if err != 0:
            raise _b.BeagleException("calculateLogLikelihood", err)
        return v

    # phased evaluation (pattern-sharded multi-GPU): prepare; { attempt; <all-reduce>; } until finish(global)
    def prepare(self):
        self._chk(self.h.btlPrepare(self.ptr), "prepare")

    def attempt_device(self, device_ptr):
        self._chk(self.h.btlAttemptDevice(self.ptr, device_ptr), "attempt")

    def attempt_host(self):
        out = C.c_double(0.0)
        self._chk(self.h.btlAttemptHost(self.ptr, C.byref(out)), "attempt")
        return out.value

    def finish(self, global_log_likelihood):
        return bool(self.h.btlFinish(self.ptr, global_log_likelihood))

    def storeState(self):
        self.h.btlStoreState(self.ptr)

    def restoreState(self):
        self.h.btlRestoreState(self.ptr)

    def getSiteLogLikelihoods(self):
        out = np.empty(self.pattern_count)
        self._chk(self.h.btlGetSiteLogLikelihoods(self.ptr, out.ctypes.data_as(_DP)), "getSiteLogLikelihoods")
        return out

    def counters(self):
        out = (C.c_long * 8)()
        self.h.btlCounters(self.ptr, out)
        keys = ["operations", "matrix_updates", "evaluations", "rescale_retries", "last_op_count",
                "last_branch_count", "use_scale_factors", "ever_underflowed"]
        return dict(zip(keys, list(out)))

    def host_phase_times(self):
        """Development (BTL_TIMING=1 in the environment at creation): host microseconds per phase of calculateLogLikelihood summed
        since the last call — tools/step_profile.py."""
        out = (C.c_double * 8)()
        on = self.h.btlTimings(self.ptr, out)
        keys = ["traversal", "model_uploads", "updateTransitionMatrices", "updatePartials", "scale_factor_calls",
                "weights_frequencies", "root_enqueue_and_wait"]
        return dict(zip(keys, list(out))) if on else None

    def last_operations(self):
        buf = np.zeros((self.tip_count - 1) * 7, dtype=np.int32)
        n = self.h.btlLastOperations(self.ptr, buf.ctypes.data_as(_IP), self.tip_count - 1)
        return buf[:n * 7].reshape(n, 7)

    def root_buffer_index(self):
        return self.h.btlRootBufferIndex(self.ptr)

    def node_buffer_index(self, node):
        return self.h.btlNodeBufferIndex(self.ptr, node)

    def node_scale_index(self, node):
        return self.h.btlNodeScaleIndex(self.ptr, node)

    def cumulative_scale_index(self):
        return self.h.btlCumulativeScaleIndex(self.ptr)

    def close(self):
        if self.ptr:
            self.h.btlDestroy(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
