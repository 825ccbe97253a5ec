"""ctypes binding of the C ABI (include/beagle_mi355.h) — the Python-side stand-in for
``beagle.BeagleJNIImpl`` (lib/beagle.jar!beagle/BeagleJNIImpl.class), which wraps the JNI natives in
the ``beagle.Beagle`` interface and turns non-zero return codes into ``BeagleException``.

Method names and argument order are those of the ``beagle.Beagle`` Java interface, so test code reads
like the reference's callers (e.g. ``beagle.updatePartials(operations, operationCount, Beagle.NONE)``,
src/dr/evomodel/treelikelihood/BeagleTreeLikelihood.java:1003).

There is NO fallback: if the HIP library is missing this module raises at load time
(``BeagleFactory.loadBeagleInstance`` would fall back to a Java implementation; this engine does not).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ENGINE_LIB = os.path.join(_HERE, "lib", "libhmsbeagle-jni.so")
if os.environ.get("BEAGLE_MI355_ENGINE_LIB"):        # development: an A/B build of the same engine (tools/build_variant.sh)
    ENGINE_LIB = os.environ["BEAGLE_MI355_ENGINE_LIB"]
HOST_LIB = os.path.join(_HERE, "lib", "libbeast_host.so")

NONE = -1
# instance / requirement flag bits (lib/beagle.jar!beagle/BeagleFlag; include/beagle_mi355.h)
FLAG_EIGEN_REAL, FLAG_EIGEN_COMPLEX, FLAG_PROCESSOR_GPU, FLAG_FRAMEWORK_CPU = 1 << 4, 1 << 5, 1 << 16, 1 << 27
OPERATION_TUPLE_SIZE = 7

ERROR_NAMES = {0: "NO_ERROR", -1: "GENERAL_ERROR", -2: "OUT_OF_MEMORY_ERROR", -3: "UNIDENTIFIED_EXCEPTION_ERROR",
               -4: "UNINITIALIZED_INSTANCE_ERROR", -5: "OUT_OF_RANGE_ERROR", -6: "NO_RESOURCE_ERROR",
               -7: "NO_IMPLEMENTATION_ERROR", -8: "FLOATING_POINT_ERROR"}


class BeagleException(RuntimeError):
    """Mirror of beagle.BeagleException(functionName, errCode)."""

    def __init__(self, function_name, code):
        super().__init__("%s returned %d (%s)" % (function_name, code, ERROR_NAMES.get(code, "?")))
        self.function_name = function_name
        self.code = code


class InstanceDetails(C.Structure):
    _fields_ = [("resourceNumber", C.c_int), ("resourceName", C.c_char_p), ("implName", C.c_char_p),
                ("implDescription", C.c_char_p), ("flags", C.c_long)]


class _Resource(C.Structure):
    _fields_ = [("name", C.c_char_p), ("description", C.c_char_p), ("supportFlags", C.c_long),
                ("requiredFlags", C.c_long)]


class _ResourceList(C.Structure):
    _fields_ = [("list", C.POINTER(_Resource)), ("length", C.c_int)]


_DP = C.POINTER(C.c_double)
_IP = C.POINTER(C.c_int)


def _d(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float64)


def _i(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.int32)


def _dp(a):
    return None if a is None else a.ctypes.data_as(_DP)


def _ip(a):
    return None if a is None else a.ctypes.data_as(_IP)


_PROTOS = {
    "CreateInstance": ([C.c_int] * 9 + [_IP, C.c_int, C.c_long, C.c_long, C.POINTER(InstanceDetails)], C.c_int),
    "FinalizeInstance": ([C.c_int], C.c_int),
    "SetCPUThreadCount": ([C.c_int, C.c_int], C.c_int),
    "SetPatternWeights": ([C.c_int, _DP], C.c_int),
    "SetPatternPartitions": ([C.c_int, C.c_int, _IP], C.c_int),
    "SetTipStates": ([C.c_int, C.c_int, _IP], C.c_int),
    "GetTipStates": ([C.c_int, C.c_int, _IP], C.c_int),
    "SetTipPartials": ([C.c_int, C.c_int, _DP], C.c_int),
    "SetPartials": ([C.c_int, C.c_int, _DP], C.c_int),
    "GetPartials": ([C.c_int, C.c_int, C.c_int, _DP], C.c_int),
    "GetLogScaleFactors": ([C.c_int, C.c_int, _DP], C.c_int),
    "SetEigenDecomposition": ([C.c_int, C.c_int, _DP, _DP, _DP], C.c_int),
    "SetStateFrequencies": ([C.c_int, C.c_int, _DP], C.c_int),
    "SetCategoryWeights": ([C.c_int, C.c_int, _DP], C.c_int),
    "SetCategoryRates": ([C.c_int, _DP], C.c_int),
    "SetCategoryRatesWithIndex": ([C.c_int, C.c_int, _DP], C.c_int),
    "SetTransitionMatrix": ([C.c_int, C.c_int, _DP, C.c_double], C.c_int),
    "GetTransitionMatrix": ([C.c_int, C.c_int, _DP], C.c_int),
    "ConvolveTransitionMatrices": ([C.c_int, _IP, _IP, _IP, C.c_int], C.c_int),
    "UpdateTransitionMatrices": ([C.c_int, C.c_int, _IP, _IP, _IP, _DP, C.c_int], C.c_int),
    "UpdateTransitionMatricesWithMultipleModels": ([C.c_int, _IP, _IP, _IP, _IP, _IP, _DP, C.c_int], C.c_int),
    "UpdatePartials": ([C.c_int, _IP, C.c_int, C.c_int], C.c_int),
    "UpdatePartialsByPartition": ([C.c_int, _IP, C.c_int], C.c_int),
    "WaitForPartials": ([C.c_int, _IP, C.c_int], C.c_int),
    "AccumulateScaleFactors": ([C.c_int, _IP, C.c_int, C.c_int], C.c_int),
    "AccumulateScaleFactorsByPartition": ([C.c_int, _IP, C.c_int, C.c_int, C.c_int], C.c_int),
    "RemoveScaleFactors": ([C.c_int, _IP, C.c_int, C.c_int], C.c_int),
    "RemoveScaleFactorsByPartition": ([C.c_int, _IP, C.c_int, C.c_int, C.c_int], C.c_int),
    "ResetScaleFactors": ([C.c_int, C.c_int], C.c_int),
    "ResetScaleFactorsByPartition": ([C.c_int, C.c_int, C.c_int], C.c_int),
    "CopyScaleFactors": ([C.c_int, C.c_int, C.c_int], C.c_int),
    "CalculateRootLogLikelihoods": ([C.c_int, _IP, _IP, _IP, _IP, C.c_int, _DP], C.c_int),
    "CalculateRootLogLikelihoodsByPartition": ([C.c_int, _IP, _IP, _IP, _IP, _IP, C.c_int, C.c_int, _DP, _DP], C.c_int),
    "GetSiteLogLikelihoods": ([C.c_int, _DP], C.c_int),
    "SetRootPrePartials": ([C.c_int, _IP, _IP, C.c_int], C.c_int),
    "SetDifferentialMatrix": ([C.c_int, C.c_int, _DP], C.c_int),
    "AddTransitionMatrices": ([C.c_int, _IP, _IP, _IP, C.c_int], C.c_int),
    "TransposeTransitionMatrices": ([C.c_int, _IP, _IP, C.c_int], C.c_int),
    "UpdatePrePartials": ([C.c_int, _IP, C.c_int, C.c_int], C.c_int),
    "UpdatePrePartialsByPartition": ([C.c_int, _IP, C.c_int], C.c_int),
    "CalculateEdgeDifferentials": ([C.c_int, _IP, _IP, _IP, _IP, C.c_int, _DP, _DP, _DP], C.c_int),
    "CalculateCrossProductDifferentials": ([C.c_int, _IP, _IP, _IP, _IP, _DP, C.c_int, _DP, _DP], C.c_int),
}

# symbols every engine library must export (tests assert this list against include/beagle_mi355.h)
ABI_SYMBOLS = ["beagleGetVersion", "beagleGetCitation", "beagleGetResourceList", "beagleGetBenchmarkedResourceList", "beagleGetApiTable",
               "beagleGetPartitionApiTable"] + \
              ["beagle" + k for k in _PROTOS] + \
              ["beagleMi355SetStream", "beagleMi355CalculateRootLogLikelihoodsDevice", "beagleMi355Synchronize",
               "beagleMi355KernelTimer", "beagleMi355DeviceBytes", "beagleMi355WalkStats", "beagleMi355GradientStats", "beagleMi355GetPartialsBatch",
               "beagleMi355GetPartialsPinned", "beagleMi355GetSiteLogLikelihoodsPinned",
               "beagleMi355KernelTimerCalls", "beagleMi355WalkHealth", "beagleMi355WalkLaunchInfo", "beagleMi355RootFusedCount", "beagleMi355SitePrefetchCount", "beagleMi355KernelTimerRestart", "beagleMi355GetDimensions", "beagleMi355GetCommUniqueId", "beagleMi355CommInit", "beagleMi355CommInfo", "beagleMi355CalculateRootLogLikelihoodsAllReduce"]


class EngineLibrary:
    """A loaded engine shared object (the HIP engine, or — in tests only — the CPU oracle, whose
    symbols carry the ``oracle_`` prefix)."""

    def __init__(self, path=ENGINE_LIB, prefix=""):
        if not os.path.exists(path):
            raise OSError("engine library %s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(there is no CPU fallback)" % path)
        self.path = path
        self.prefix = prefix
        self.lib = C.CDLL(path, mode=C.RTLD_LOCAL)
        self.fn = {}
        for name, (argtypes, restype) in _PROTOS.items():
            f = getattr(self.lib, prefix + "beagle" + name, None)
            if f is None:
                continue
            f.argtypes = argtypes
            f.restype = restype
            self.fn[name] = f
        gv = getattr(self.lib, prefix + "beagleGetVersion")
        gv.restype = C.c_char_p
        self.version = gv().decode()
        tab = getattr(self.lib, prefix + "beagleGetApiTable")
        tab.restype = C.c_void_p
        self.api_table = tab()
        ptab = getattr(self.lib, prefix + "beagleGetPartitionApiTable", None)       # (the engine; the oracle restates no ...ByPartition call)
        self.partition_api_table = None
        if ptab is not None:
            ptab.restype = C.c_void_p
            self.partition_api_table = ptab()

    def has(self, name):
        return name in self.fn

    def resource_list(self):
        f = getattr(self.lib, self.prefix + "beagleGetResourceList")
        f.restype = C.POINTER(_ResourceList)
        rl = f().contents
        return [(rl.list[i].name.decode(), rl.list[i].description.decode(), rl.list[i].supportFlags)
                for i in range(rl.length)]


_engine = None


def engine():
    """The HIP engine library (loaded once)."""
    global _engine
    if _engine is None:
        _engine = EngineLibrary(ENGINE_LIB)
    return _engine


class Beagle:
    """One engine instance behind the ``beagle.Beagle`` method set."""

    NONE = NONE
    OPERATION_TUPLE_SIZE = OPERATION_TUPLE_SIZE

    def __init__(self, tipCount, partialsBufferCount, compactBufferCount, stateCount, patternCount,
                 eigenBufferCount, matrixBufferCount, categoryCount, scaleBufferCount,
                 resourceList=(1,), preferenceFlags=0, requirementFlags=0, library=None):
        self.lib = library or engine()
        self._f = self.lib.fn
        self.stateCount, self.patternCount, self.categoryCount = stateCount, patternCount, categoryCount
        self.details = InstanceDetails()
        rl = _i(list(resourceList)) if resourceList is not None else None
        h = self._f["CreateInstance"](tipCount, partialsBufferCount, compactBufferCount, stateCount, patternCount,
                                      eigenBufferCount, matrixBufferCount, categoryCount, scaleBufferCount,
                                      _ip(rl), 0 if rl is None else len(rl), preferenceFlags, requirementFlags,
                                      C.byref(self.details))
        if h < 0:
            raise BeagleException("create", h)
        self.instance = h

    def _check(self, name, rc):
        if rc != 0:
            raise BeagleException(name, rc)

    def finalize(self):
        if self.instance >= 0:
            self._check("finalize", self._f["FinalizeInstance"](self.instance))
            self.instance = -1

    def setCPUThreadCount(self, n):
        self._check("setCPUThreadCount", self._f["SetCPUThreadCount"](self.instance, n))

    def setPatternWeights(self, w):
        w = _d(w)
        assert w.size >= self.patternCount
        self._check("setPatternWeights", self._f["SetPatternWeights"](self.instance, _dp(w)))

    def setPatternPartitions(self, partitionCount, partitions):
        a = _i(partitions)
        self._check("setPatternPartitions", self._f["SetPatternPartitions"](self.instance, partitionCount, _ip(a)))

    def setTipStates(self, tipIndex, states):
        a = _i(states)
        assert a.size >= self.patternCount
        self._check("setTipStates", self._f["SetTipStates"](self.instance, tipIndex, _ip(a)))

    def getTipStates(self, tipIndex):
        out = np.empty(self.patternCount, dtype=np.int32)
        self._check("getTipStates", self._f["GetTipStates"](self.instance, tipIndex, _ip(out)))
        return out

    def setTipPartials(self, tipIndex, partials):
        a = _d(partials)
        assert a.size >= self.patternCount * self.stateCount
        self._check("setTipPartials", self._f["SetTipPartials"](self.instance, tipIndex, _dp(a)))

    def setPartials(self, bufferIndex, partials):
        a = _d(partials)
        assert a.size >= self.patternCount * self.stateCount * self.categoryCount
        self._check("setPartials", self._f["SetPartials"](self.instance, bufferIndex, _dp(a)))

    @classmethod
    def attach(cls, tl):
        """This binding on the instance a host driver (treelikelihood.BeagleTreeLikelihood) already owns."""
        raw = cls.__new__(cls)
        raw.lib, raw._f, raw.instance = tl.engine, tl.engine.fn, tl.instance
        raw.stateCount, raw.patternCount, raw.categoryCount = tl.state_count, tl.pattern_count, tl.category_count
        return raw

    def getPartials(self, bufferIndex, scaleIndex=NONE):
        out = np.empty(self.categoryCount * self.patternCount * self.stateCount)
        self._check("getPartials", self._f["GetPartials"](self.instance, bufferIndex, scaleIndex, _dp(out)))
        return out.reshape(self.categoryCount, self.patternCount, self.stateCount)

    def getLogScaleFactors(self, scaleIndex):
        out = np.empty(self.patternCount)
        self._check("getLogScaleFactors", self._f["GetLogScaleFactors"](self.instance, scaleIndex, _dp(out)))
        return out

    def setEigenDecomposition(self, eigenIndex, eigenVectors, inverseEigenValues, eigenValues):
        u, ui, lam = _d(eigenVectors), _d(inverseEigenValues), _d(eigenValues)
        self._check("setEigenDecomposition",
                    self._f["SetEigenDecomposition"](self.instance, eigenIndex, _dp(u), _dp(ui), _dp(lam)))

    def setStateFrequencies(self, index, freqs):
        a = _d(freqs)
        self._check("setStateFrequencies", self._f["SetStateFrequencies"](self.instance, index, _dp(a)))

    def setCategoryWeights(self, index, weights):
        a = _d(weights)
        self._check("setCategoryWeights", self._f["SetCategoryWeights"](self.instance, index, _dp(a)))

    def setCategoryRates(self, rates):
        a = _d(rates)
        self._check("setCategoryRates", self._f["SetCategoryRates"](self.instance, _dp(a)))

    def setCategoryRatesWithIndex(self, index, rates):
        a = _d(rates)
        self._check("setCategoryRatesWithIndex", self._f["SetCategoryRatesWithIndex"](self.instance, index, _dp(a)))

    def setTransitionMatrix(self, matrixIndex, matrix, paddedValue=1.0):
        a = _d(matrix)
        self._check("setTransitionMatrix", self._f["SetTransitionMatrix"](self.instance, matrixIndex, _dp(a), paddedValue))

    def getTransitionMatrix(self, matrixIndex):
        out = np.empty(self.categoryCount * self.stateCount * self.stateCount)
        self._check("getTransitionMatrix", self._f["GetTransitionMatrix"](self.instance, matrixIndex, _dp(out)))
        return out.reshape(self.categoryCount, self.stateCount, self.stateCount)

    def convolveTransitionMatrices(self, first, second, result, count):
        a, b, c = _i(first), _i(second), _i(result)
        self._check("convolveTransitionMatrices",
                    self._f["ConvolveTransitionMatrices"](self.instance, _ip(a), _ip(b), _ip(c), count))

    def updateTransitionMatrices(self, eigenIndex, probabilityIndices, firstDerivativeIndices,
                                 secondDerivativeIndices, edgeLengths, count):
        p, d1, d2, t = _i(probabilityIndices), _i(firstDerivativeIndices), _i(secondDerivativeIndices), _d(edgeLengths)
        self._check("updateTransitionMatrices",
                    self._f["UpdateTransitionMatrices"](self.instance, eigenIndex, _ip(p), _ip(d1), _ip(d2), _dp(t), count))

    def updateTransitionMatricesWithMultipleModels(self, eigenIndices, categoryRateIndices, probabilityIndices,
                                                   firstDerivativeIndices, secondDerivativeIndices, edgeLengths, count):
        e, r, p = _i(eigenIndices), _i(categoryRateIndices), _i(probabilityIndices)
        d1, d2, t = _i(firstDerivativeIndices), _i(secondDerivativeIndices), _d(edgeLengths)
        self._check("updateTransitionMatricesWithMultipleModels",
                    self._f["UpdateTransitionMatricesWithMultipleModels"](self.instance, _ip(e), _ip(r), _ip(p),
                                                                           _ip(d1), _ip(d2), _dp(t), count))

    def updatePartials(self, operations, operationCount, cumulativeScaleIndex=NONE):
        a = _i(operations)
        assert a.size >= operationCount * OPERATION_TUPLE_SIZE
        self._check("updatePartials", self._f["UpdatePartials"](self.instance, _ip(a), operationCount, cumulativeScaleIndex))

    def updatePartialsByPartition(self, operations, operationCount):
        a = _i(operations)
        assert a.size >= operationCount * 9
        self._check("updatePartialsByPartition", self._f["UpdatePartialsByPartition"](self.instance, _ip(a), operationCount))

    def accumulateScaleFactors(self, scaleIndices, count, cumulativeScaleIndex):
        a = _i(scaleIndices)
        self._check("accumulateScaleFactors",
                    self._f["AccumulateScaleFactors"](self.instance, _ip(a), count, cumulativeScaleIndex))

    def removeScaleFactors(self, scaleIndices, count, cumulativeScaleIndex):
        a = _i(scaleIndices)
        self._check("removeScaleFactors", self._f["RemoveScaleFactors"](self.instance, _ip(a), count, cumulativeScaleIndex))

    def resetScaleFactors(self, cumulativeScaleIndex):
        self._check("resetScaleFactors", self._f["ResetScaleFactors"](self.instance, cumulativeScaleIndex))

    def copyScaleFactors(self, dest, src):
        self._check("copyScaleFactors", self._f["CopyScaleFactors"](self.instance, dest, src))

    def accumulateScaleFactorsByPartition(self, scaleIndices, count, cumulativeScaleIndex, partitionIndex):
        a = _i(scaleIndices)
        self._check("accumulateScaleFactorsByPartition",
                    self._f["AccumulateScaleFactorsByPartition"](self.instance, _ip(a), count, cumulativeScaleIndex, partitionIndex))

    def resetScaleFactorsByPartition(self, cumulativeScaleIndex, partitionIndex):
        self._check("resetScaleFactorsByPartition",
                    self._f["ResetScaleFactorsByPartition"](self.instance, cumulativeScaleIndex, partitionIndex))

    def calculateRootLogLikelihoods(self, bufferIndices, categoryWeightsIndices, stateFrequenciesIndices,
                                    cumulativeScaleIndices, count, outSumLogLikelihood):
        """``outSumLogLikelihood``: writable sequence of length >= 1 (as the Java double[1]).
        Error -8 (FLOATING_POINT) is tolerated exactly as BeagleJNIImpl#calculateRootLogLikelihoods does."""
        b, w, f, s = _i(bufferIndices), _i(categoryWeightsIndices), _i(stateFrequenciesIndices), _i(cumulativeScaleIndices)
        out = np.zeros(1)
        rc = self._f["CalculateRootLogLikelihoods"](self.instance, _ip(b), _ip(w), _ip(f), _ip(s), count, _dp(out))
        outSumLogLikelihood[0] = out[0]
        if rc != 0 and rc != -8:
            raise BeagleException("calculateRootLogLikelihoods", rc)

    def calculateRootLogLikelihoodsByPartition(self, bufferIndices, categoryWeightsIndices, stateFrequenciesIndices,
                                               cumulativeScaleIndices, partitionIndices, partitionCount, count,
                                               outSumLogLikelihoodByPartition, outSumLogLikelihood):
        b, w, f = _i(bufferIndices), _i(categoryWeightsIndices), _i(stateFrequenciesIndices)
        s, p = _i(cumulativeScaleIndices), _i(partitionIndices)
        byp = np.zeros(partitionCount * count)
        tot = np.zeros(1)
        rc = self._f["CalculateRootLogLikelihoodsByPartition"](self.instance, _ip(b), _ip(w), _ip(f), _ip(s), _ip(p),
                                                               partitionCount, count, _dp(byp), _dp(tot))
        outSumLogLikelihoodByPartition[:len(byp)] = byp
        outSumLogLikelihood[0] = tot[0]
        if rc != 0 and rc != -8:
            raise BeagleException("calculateRootLogLikelihoodsByPartition", rc)

    # ---- pre-order partials and branch gradients (AbstractBeagleGradientDelegate / ...BranchGradientDelegate) ----
    def setRootPrePartials(self, bufferIndices, stateFrequenciesIndices, count):
        b, f = _i(bufferIndices), _i(stateFrequenciesIndices)
        self._check("setRootPrePartials", self._f["SetRootPrePartials"](self.instance, _ip(b), _ip(f), count))

    def setDifferentialMatrix(self, matrixIndex, matrix):
        m = _d(matrix)
        assert m.size == self.stateCount * self.stateCount * self.categoryCount
        self._check("setDifferentialMatrix", self._f["SetDifferentialMatrix"](self.instance, matrixIndex, _dp(m)))

    def transposeTransitionMatrices(self, inputIndices, resultIndices, count):
        a, b = _i(inputIndices), _i(resultIndices)
        self._check("transposeTransitionMatrices",
                    self._f["TransposeTransitionMatrices"](self.instance, _ip(a), _ip(b), count))

    def updatePrePartials(self, operations, operationCount, cumulativeScaleIndex=NONE):
        ops = _i(operations)
        self._check("updatePrePartials", self._f["UpdatePrePartials"](self.instance, _ip(ops), operationCount,
                                                                     cumulativeScaleIndex))

    def updatePrePartialsByPartition(self, operations, operationCount):
        """9-int tuples {pre(child), writeScale, readScale, pre(parent), matrix(child), post(sibling), matrix(sibling), partition,
        cumulativeScale}"""
        ops = _i(operations)
        self._check("updatePrePartialsByPartition", self._f["UpdatePrePartialsByPartition"](self.instance, _ip(ops), operationCount))

    def addTransitionMatrices(self, first, second, result, count):
        a, b, c = _i(first), _i(second), _i(result)
        self._check("addTransitionMatrices", self._f["AddTransitionMatrices"](self.instance, _ip(a), _ip(b), _ip(c), count))

    def calculateEdgeDifferentials(self, postBufferIndices, preBufferIndices, derivativeMatrixIndices,
                                   categoryWeightsIndices, count, want_per_pattern=False, want_squared=True):
        """-> (outSumDerivatives[count], outSumSquaredDerivatives[count] or None, outDerivatives[count, P] or None); the two
        optional outputs are passed as NULL when not wanted, as the gradient delegates do
        (AbstractBeagleBranchGradientDelegate.java:82-92)."""
        po, pr, dm, cw = _i(postBufferIndices), _i(preBufferIndices), _i(derivativeMatrixIndices), _i(categoryWeightsIndices)
        s1 = np.zeros(count)
        s2 = np.zeros(count) if want_squared else None
        per = np.zeros((count, self.patternCount)) if want_per_pattern else None
        self._check("calculateEdgeDifferentials",
                    self._f["CalculateEdgeDifferentials"](self.instance, _ip(po), _ip(pr), _ip(dm), _ip(cw), count,
                                                          _dp(per) if per is not None else None, _dp(s1),
                                                          _dp(s2) if s2 is not None else None))
        return s1, s2, per

    def calculateCrossProductDifferentials(self, postBufferIndices, preBufferIndices, categoryRateIndices,
                                           categoryWeightsIndices, edgeLengths, count, out=None):
        """-> outSumDerivatives[S*S] (accumulated into ``out`` when given, as BEAST's zero-filled array)."""
        po, pr = _i(postBufferIndices), _i(preBufferIndices)
        cr, cw, t = _i(categoryRateIndices), _i(categoryWeightsIndices), _d(edgeLengths)
        acc = np.zeros(self.stateCount * self.stateCount) if out is None else out
        self._check("calculateCrossProductDifferentials",
                    self._f["CalculateCrossProductDifferentials"](self.instance, _ip(po), _ip(pr), _ip(cr), _ip(cw), _dp(t),
                                                                  count, _dp(acc), None))
        return acc

    def getSiteLogLikelihoods(self, out=None):
        if out is None:
            out = np.empty(self.patternCount)
        self._check("getSiteLogLikelihoods", self._f["GetSiteLogLikelihoods"](self.instance, _dp(out)))
        return out

    # --- MI355X extensions -----------------------------------------------------------------
    def _ext(self, name, argtypes, restype=C.c_int):
        f = getattr(self.lib.lib, name)
        f.argtypes = argtypes
        f.restype = restype
        return f

    def setStream(self, hip_stream):
        self._check("setStream", self._ext("beagleMi355SetStream", [C.c_int, C.c_void_p])(self.instance, hip_stream))

    def calculateRootLogLikelihoodsDevice(self, bufferIndex, categoryWeightsIndex, stateFrequenciesIndex,
                                          cumulativeScaleIndex, device_ptr):
        f = self._ext("beagleMi355CalculateRootLogLikelihoodsDevice", [C.c_int] * 5 + [C.c_void_p])
        self._check("calculateRootLogLikelihoodsDevice",
                    f(self.instance, bufferIndex, categoryWeightsIndex, stateFrequenciesIndex, cumulativeScaleIndex, device_ptr))

    def synchronize(self):
        self._check("synchronize", self._ext("beagleMi355Synchronize", [C.c_int])(self.instance))

    # one process per GPU: the collective inside the engine (include/beagle_mi355.h)
    def commUniqueId(self):
        """128 bytes that identify a new communicator; produced on ONE rank and handed to all (any channel)."""
        buf = C.create_string_buffer(128)
        self._check("getCommUniqueId", self._ext("beagleMi355GetCommUniqueId", [C.c_void_p])(buf))
        return buf.raw

    def commInit(self, unique_id, rank, rank_count):
        buf = C.create_string_buffer(bytes(unique_id), 128)
        self._check("commInit", self._ext("beagleMi355CommInit", [C.c_int, C.c_void_p, C.c_int, C.c_int])(self.instance, buf, rank, rank_count))

    def commRanks(self):
        """Ranks of the instance's communicator as RCCL counts them (0: none) — include/beagle_mi355.h beagleMi355CommInfo."""
        n = C.c_int(0)
        self._check("commInfo", self._ext("beagleMi355CommInfo", [C.c_int, C.POINTER(C.c_int)])(self.instance, C.byref(n)))
        return n.value

    def calculateRootLogLikelihoodsAllReduce(self, bufferIndex, categoryWeightsIndex, stateFrequenciesIndex, cumulativeScaleIndex):
        out = C.c_double(0.0)
        f = self._ext("beagleMi355CalculateRootLogLikelihoodsAllReduce", [C.c_int] * 5 + [C.POINTER(C.c_double)])
        rc = f(self.instance, bufferIndex, categoryWeightsIndex, stateFrequenciesIndex, cumulativeScaleIndex, C.byref(out))
        if rc not in (0, -8):
            self._check("calculateRootLogLikelihoodsAllReduce", rc)
        return out.value

    def kernelTimer(self, enable):
        ms = C.c_double(0.0)
        n = C.c_long(0)
        f = self._ext("beagleMi355KernelTimer", [C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_long)])
        self._check("kernelTimer", f(self.instance, int(enable), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def kernelTimerRestart(self):
        """Forget what the enabled kernel timer and the walk counters hold (no synchronisation: the caller has just done one)."""
        self._check("kernelTimerRestart", self._ext("beagleMi355KernelTimerRestart", [C.c_int])(self.instance))

    def kernelTimerCalls(self):
        """updatePartials calls the kernel timer bracketed since this was last asked (kernelTimer(N > 1) samples every N-th)."""
        n = C.c_long(0)
        self._check("kernelTimerCalls", self._ext("beagleMi355KernelTimerCalls", [C.c_int, C.POINTER(C.c_long)])(self.instance, C.byref(n)))
        return n.value

    def rootFusedCount(self):
        """calculateRootLogLikelihoods calls answered inside the walk's launch (include/beagle_mi355.h beagleMi355RootFusedCount)."""
        n = C.c_long(0)
        self._check("rootFusedCount", self._ext("beagleMi355RootFusedCount", [C.c_int, C.POINTER(C.c_long)])(self.instance, C.byref(n)))
        return n.value

    def sitePrefetchCount(self):
        """getSiteLogLikelihoods calls that found the values already on the host (include/beagle_mi355.h beagleMi355SitePrefetchCount)."""
        n = C.c_long(0)
        self._check("sitePrefetchCount", self._ext("beagleMi355SitePrefetchCount", [C.c_int, C.POINTER(C.c_long)])(self.instance, C.byref(n)))
        return n.value

    def getPartialsBatch(self, bufferIndices, scaleIndices=None):
        """-> [count][C][P][S]: several buffers in one call (include/beagle_mi355.h beagleMi355GetPartialsBatch)."""
        b = _i(list(bufferIndices))
        sc = None if scaleIndices is None else _i(list(scaleIndices))
        out = np.empty(len(b) * self.categoryCount * self.patternCount * self.stateCount)
        f = self._ext("beagleMi355GetPartialsBatch", [C.c_int, _IP, _IP, C.c_int, _DP])
        self._check("getPartialsBatch", f(self.instance, _ip(b), _ip(sc), len(b), _dp(out)))
        return out.reshape(len(b), self.categoryCount, self.patternCount, self.stateCount)

    def walkStats(self):
        """Counters of the 4-state pattern walk since the last kernelTimer call (include/beagle_mi355.h)."""
        out = (C.c_long * 8)()
        self._check("walkStats", self._ext("beagleMi355WalkStats", [C.c_int, C.POINTER(C.c_long)])(self.instance, out))
        keys = ("micro_ops", "stored", "mem_reads", "tip_reads", "scale_reads", "walks", "scale_writes", "fast_walks")
        return {k: int(out[i]) for i, k in enumerate(keys)}

    def walkHealth(self):
        """The one-launch walk since instance creation (include/beagle_mi355.h beagleMi355WalkHealth)."""
        out = (C.c_long * 4)()
        self._check("walkHealth", self._ext("beagleMi355WalkHealth", [C.c_int, C.POINTER(C.c_long)])(self.instance, out))
        return {"self_served": int(out[0]), "spin_limit_us": int(out[1]), "folded_vectors": int(out[2]), "fold_builds": int(out[3])}

    def walkLaunchInfo(self):
        """How the one-launch walks were run (include/beagle_mi355.h beagleMi355WalkLaunchInfo)."""
        out = (C.c_long * 8)()
        self._check("walkLaunchInfo", self._ext("beagleMi355WalkLaunchInfo", [C.c_int, C.POINTER(C.c_long)])(self.instance, out))
        return {"ticket_walks": int(out[0]), "flag_walks": int(out[1]), "rows": int(out[2]), "slices": int(out[3]),
                "fused_cherries": int(out[4]), "micro_ops": int(out[5]), "slice_accumulations": int(out[6]), "partition_roots_in_walk": int(out[7])}

    def gradientStats(self):
        """How the pre-order lists of this instance were run (include/beagle_mi355.h beagleMi355GradientStats)."""
        out = (C.c_long * 4)()
        self._check("gradientStats", self._ext("beagleMi355GradientStats", [C.c_int, C.POINTER(C.c_long)])(self.instance, out))
        return {"fused": int(out[0]), "by_operation": int(out[1]), "walked": int(out[2]), "late": int(out[3])}

    def deviceBytes(self):
        return self._ext("beagleMi355DeviceBytes", [C.c_int], C.c_long)(self.instance)
