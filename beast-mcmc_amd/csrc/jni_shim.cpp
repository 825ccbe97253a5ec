// jni_shim.cpp — the 47 `native` methods of lib/beagle.jar!beagle/BeagleJNIWrapper.class as
// Java_beagle_BeagleJNIWrapper_<name> symbols, each forwarding one-to-one to the C ABI of
// include/beagle_mi355.h.  With this file the engine's shared object IS `libhmsbeagle-jni.so`:
// BEAST loads it with System.loadLibrary("hmsbeagle-jni") (BeagleJNIWrapper#loadBeagleLibrary) and
// BeagleTreeLikelihood / TreeDataLikelihood call it unchanged.
//
// Parameter order of every function = the method descriptor in the class file (after JNIEnv*, jobject).
// Java arrays are COPIED with Get<Type>ArrayRegion into library-owned buffers and, when they are outputs, copied back with
// Set<Type>ArrayRegion (no pinning semantics to get wrong).  What is copied is what the call uses, not the array:
//   * inputs: the `count`-derived number of entries — BEAST's arrays are longer than that (operations[] is sized
//     internalNodeCount * 7 whatever the count, BeagleDataLikelihoodDelegate.java:183; edgeLengths is the whole
//     branchLengths[nodeCount], :179, 838-843);
//   * outputs: nothing on the way in, exactly the entries written on the way out; getPartials and getSiteLogLikelihoods
//     (12.8 MB and 0.8 MB per call in the metric's configuration, the latter once per evaluation of BeagleTreeLikelihood,
//     BeagleTreeLikelihood.java:1050) go from the engine's pinned bounce buffer straight into the Java array.
// null arrays are passed through as NULL (HomogenousSubstitutionModelDelegate.java:260-261 passes null derivative indices).
//
// Compiled against a self-authored minimal JNI header (jni_min.h).  The image has no JVM; tests/native/fake_jvm.cpp is a
// JVM-less JNIEnv (229-slot function table over plain C++ objects) that drives these symbols end to end on the GPU box and
// counts the bytes every call moves (tests/test_gpu_jni_shim.py); INTEGRATION.md §4 lists the on-JVM validation steps.
#include <string.h>

#include <algorithm>
#include <vector>

#include "../../include/beagle_mi355.h"
#include "jni_min.h"

namespace {

enum Mode { IN, OUT, INOUT };
// n < 0: the whole array.  n >= 0: the number of entries the call defines (inputs: reads; outputs: writes) — a Java array
// shorter than that is an error the wrapper reports (tooShort) instead of letting the library run past the copy.
// Outputs go back to the Java array only when the call succeeded (commit), and only the n entries it defines.
struct IntArr {
    JNIEnv* env; jintArray arr; std::vector<jint> buf; Mode mode; bool tooShort = false, write = true;
    IntArr(JNIEnv* e, jintArray a, long n = -1, Mode m = IN) : env(e), arr(a), mode(m) {
        if (!a) return;
        const long len = (long)jni::GetArrayLength(e, a);
        tooShort = n > len;
        buf.resize((size_t)(n < 0 ? len : std::min(n, len)));
        if (mode != OUT && !buf.empty()) jni::GetIntArrayRegion(e, a, 0, (jsize)buf.size(), buf.data());
    }
    ~IntArr() { if (arr && mode != IN && write && !tooShort && !buf.empty()) jni::SetIntArrayRegion(env, arr, 0, (jsize)buf.size(), buf.data()); }
    operator int*() { return arr ? buf.data() : nullptr; }
};
struct DblArr {
    JNIEnv* env; jdoubleArray arr; std::vector<jdouble> buf; Mode mode; bool tooShort = false, write = true;
    DblArr(JNIEnv* e, jdoubleArray a, long n = -1, Mode m = IN) : env(e), arr(a), mode(m) {
        if (!a) return;
        const long len = (long)jni::GetArrayLength(e, a);
        tooShort = n > len;
        buf.resize((size_t)(n < 0 ? len : std::min(n, len)));
        if (mode != OUT && !buf.empty()) jni::GetDoubleArrayRegion(e, a, 0, (jsize)buf.size(), buf.data());
    }
    ~DblArr() { if (arr && mode != IN && write && !tooShort && !buf.empty()) jni::SetDoubleArrayRegion(env, arr, 0, (jsize)buf.size(), buf.data()); }
    operator double*() { return arr ? buf.data() : nullptr; }
};
inline bool anyShort() { return false; }
template <class A, class... R> inline bool anyShort(const A& a, const R&... r) { return a.tooShort || anyShort(r...); }
// the result code decides whether output arrays are written back: yes on success and on BEAGLE_ERROR_FLOATING_POINT (the value IS
// the result: NaN — BeagleJNIImpl tolerates -8), no otherwise (the Java array keeps what it held)
inline void commitTo(int) {}
template <class A, class... R> inline void commitTo(int rc, A& a, R&... r) { a.write = rc == BEAGLE_SUCCESS || rc == BEAGLE_ERROR_FLOATING_POINT; commitTo(rc, r...); }
#define SHORT_CHECK(...) do { if (anyShort(__VA_ARGS__)) return BEAGLE_ERROR_OUT_OF_RANGE; } while (0)
// what an instance's whole-array outputs define: patterns, states, categories (beagleMi355GetDimensions)
struct Dims { long P = -1, S = -1, C = -1; bool ok = false; };
inline Dims dimsOf(int instance) {
    int d[8];
    Dims r;
    if (beagleMi355GetDimensions(instance, d) == BEAGLE_SUCCESS) { r.S = d[2]; r.P = d[3]; r.C = d[4]; r.ok = true; }
    return r;
}

// a failed class / method lookup leaves a pending NoSuchMethodError / NoClassDefFoundError: clear it, the caller sees null / skips
bool pendingCleared(JNIEnv* env) {
    if (!jni::ExceptionCheck(env)) return false;
    jni::ExceptionClear(env);
    return true;
}
jmethodID method(JNIEnv* env, jclass cls, const char* name, const char* sig) {
    jmethodID m = jni::GetMethodID(env, cls, name, sig);
    if (pendingCleared(env)) return nullptr;
    return m;
}

void callSetString(JNIEnv* env, jobject obj, jclass cls, const char* name, const char* value) {
    jmethodID m = method(env, cls, name, "(Ljava/lang/String;)V");
    if (!m) return;
    jstring s = jni::NewStringUTF(env, value ? value : "");
    jni::CallVoidMethod(env)(env, obj, m, s);
    jni::DeleteLocalRef(env, s);
}
void callSetInt(JNIEnv* env, jobject obj, jclass cls, const char* name, jint v) {
    jmethodID m = method(env, cls, name, "(I)V");
    if (m) jni::CallVoidMethod(env)(env, obj, m, v);
}
void callSetLong(JNIEnv* env, jobject obj, jclass cls, const char* name, jlong v) {
    jmethodID m = method(env, cls, name, "(J)V");
    if (m) jni::CallVoidMethod(env)(env, obj, m, v);
}
void callSetDouble(JNIEnv* env, jobject obj, jclass cls, const char* name, jdouble v) {
    jmethodID m = method(env, cls, name, "(D)V");
    if (m) jni::CallVoidMethod(env)(env, obj, m, v);
}

}  // namespace

#define JNI_FN(ret, name) extern "C" JNIEXPORT ret JNICALL Java_beagle_BeagleJNIWrapper_##name

JNI_FN(jstring, getVersion)(JNIEnv* env, jobject) { return jni::NewStringUTF(env, beagleGetVersion()); }
JNI_FN(jstring, getCitation)(JNIEnv* env, jobject) { return jni::NewStringUTF(env, beagleGetCitation()); }

// getResourceList ()[Lbeagle/ResourceDetails;  — ResourceDetails(int number) + setName/setDescription/setFlags
JNI_FN(jobjectArray, getResourceList)(JNIEnv* env, jobject) {
    BeagleResourceList* rl = beagleGetResourceList();
    jclass cls = jni::FindClass(env, "beagle/ResourceDetails");
    if (pendingCleared(env) || !cls) return nullptr;
    jmethodID ctor = method(env, cls, "<init>", "(I)V");
    jmethodID setFlags = method(env, cls, "setFlags", "(J)V");
    if (!ctor) return nullptr;
    jobjectArray out = jni::NewObjectArray(env, rl->length, cls, nullptr);
    for (int i = 0; i < rl->length; i++) {
        jobject r = jni::NewObject(env)(env, cls, ctor, (jint)i);
        callSetString(env, r, cls, "setName", rl->list[i].name);
        callSetString(env, r, cls, "setDescription", rl->list[i].description);
        if (setFlags) jni::CallVoidMethod(env)(env, r, setFlags, (jlong)rl->list[i].supportFlags);
        jni::SetObjectArrayElement(env, out, i, r);
        jni::DeleteLocalRef(env, r);
    }
    return out;
}

// getBenchmarkedResourceList (IIIII[IIJJIIIJ)[Lbeagle/BenchmarkedResourceDetails;  — BEAST's -beagle_auto
// (BeagleTreeLikelihood.java:392-414): the entries come back fastest first, element 0's resource number is what BEAST uses.
JNI_FN(jobjectArray, getBenchmarkedResourceList)(JNIEnv* env, jobject, jint tipCount, jint compactBufferCount, jint stateCount,
                                                 jint patternCount, jint categoryCount, jintArray resourceList, jint resourceCount,
                                                 jlong preferenceFlags, jlong requirementFlags, jint eigenModelCount, jint partitionCount,
                                                 jint calculateDerivatives, jlong benchmarkFlags) {
    IntArr res(env, resourceList, resourceCount);
    BeagleBenchmarkedResourceList* bl = beagleGetBenchmarkedResourceList(tipCount, compactBufferCount, stateCount, patternCount, categoryCount,
                                                                         res, resourceCount, (long)preferenceFlags, (long)requirementFlags,
                                                                         eigenModelCount, partitionCount, calculateDerivatives, (long)benchmarkFlags);
    jclass cls = jni::FindClass(env, "beagle/BenchmarkedResourceDetails");
    if (pendingCleared(env) || !cls || !bl) return nullptr;
    jmethodID ctor = method(env, cls, "<init>", "(I)V");
    if (!ctor) return nullptr;
    jobjectArray out = jni::NewObjectArray(env, bl->length, cls, nullptr);
    for (int i = 0; i < bl->length; i++) {
        const BeagleBenchmarkedResource& b = bl->list[i];
        jobject r = jni::NewObject(env)(env, cls, ctor, (jint)i);
        callSetInt(env, r, cls, "setResourceNumber", b.number);
        callSetString(env, r, cls, "setName", b.name);
        callSetString(env, r, cls, "setDescription", b.description);
        callSetLong(env, r, cls, "setSupportFlags", (jlong)b.supportFlags);
        callSetLong(env, r, cls, "setRequiredFlags", (jlong)b.requiredFlags);
        callSetInt(env, r, cls, "setReturnCode", b.returnCode);
        callSetString(env, r, cls, "setImplName", b.implName);
        callSetLong(env, r, cls, "setBenchedFlags", (jlong)b.benchedFlags);
        callSetDouble(env, r, cls, "setBenchmarkResult", b.benchmarkResult);
        callSetDouble(env, r, cls, "setPerformanceRatio", b.performanceRatio);
        jni::SetObjectArrayElement(env, out, i, r);
        jni::DeleteLocalRef(env, r);
    }
    return out;
}

// createInstance (IIIIIIIII[IIJJLbeagle/InstanceDetails;)I
JNI_FN(jint, createInstance)(JNIEnv* env, jobject, jint tipCount, jint partialsBufferCount, jint compactBufferCount,
                             jint stateCount, jint patternCount, jint eigenBufferCount, jint matrixBufferCount,
                             jint categoryCount, jint scaleBufferCount, jintArray resourceList, jint resourceCount,
                             jlong preferenceFlags, jlong requirementFlags, jobject outDetails) {
    BeagleInstanceDetails d = {0, nullptr, nullptr, nullptr, 0};
    IntArr res(env, resourceList, resourceCount);
    const int h = beagleCreateInstance(tipCount, partialsBufferCount, compactBufferCount, stateCount, patternCount,
                                       eigenBufferCount, matrixBufferCount, categoryCount, scaleBufferCount, res, resourceCount,
                                       (long)preferenceFlags, (long)requirementFlags, &d);
    if (h >= 0 && outDetails) {
        jclass cls = jni::GetObjectClass(env, outDetails);
        callSetInt(env, outDetails, cls, "setResourceNumber", (jint)d.resourceNumber);
        callSetLong(env, outDetails, cls, "setFlags", (jlong)d.flags);
        callSetString(env, outDetails, cls, "setResourceName", d.resourceName);
        callSetString(env, outDetails, cls, "setImplementationName", d.implName);
    }
    return h;
}

JNI_FN(jint, finalize)(JNIEnv*, jobject, jint instance) { return beagleFinalizeInstance(instance); }
JNI_FN(jint, setCPUThreadCount)(JNIEnv*, jobject, jint instance, jint n) { return beagleSetCPUThreadCount(instance, n); }

JNI_FN(jint, setPatternWeights)(JNIEnv* env, jobject, jint instance, jdoubleArray w) {
    DblArr a(env, w); return beagleSetPatternWeights(instance, a);
}
JNI_FN(jint, setPatternPartitions)(JNIEnv* env, jobject, jint instance, jint partitionCount, jintArray parts) {
    IntArr a(env, parts); return beagleSetPatternPartitions(instance, partitionCount, a);
}
JNI_FN(jint, setTipStates)(JNIEnv* env, jobject, jint instance, jint tip, jintArray states) {
    IntArr a(env, states); return beagleSetTipStates(instance, tip, a);
}
JNI_FN(jint, getTipStates)(JNIEnv* env, jobject, jint instance, jint tip, jintArray states) {
    const Dims d = dimsOf(instance);
    IntArr a(env, states, d.ok ? d.P : -1, OUT); SHORT_CHECK(a);
    const int rc = beagleGetTipStates(instance, tip, a); commitTo(rc, a); return rc;
}
JNI_FN(jint, setTipPartials)(JNIEnv* env, jobject, jint instance, jint tip, jdoubleArray partials) {
    DblArr a(env, partials); return beagleSetTipPartials(instance, tip, a);
}
JNI_FN(jint, setRootPrePartials)(JNIEnv* env, jobject, jint instance, jintArray bufs, jintArray freqs, jint count) {
    IntArr a(env, bufs, count), b(env, freqs, count); SHORT_CHECK(a, b);
    return beagleSetRootPrePartials(instance, a, b, count);
}
JNI_FN(jint, setPartials)(JNIEnv* env, jobject, jint instance, jint buf, jdoubleArray partials) {
    DblArr a(env, partials); return beagleSetPartials(instance, buf, a);
}
JNI_FN(jint, getPartials)(JNIEnv* env, jobject, jint instance, jint buf, jint scaleIndex, jdoubleArray out) {
    // straight from the engine's pinned bounce buffer into the Java array (no copy in, one copy out)
    const double* pinned = nullptr; long n = 0;
    const int rc = beagleMi355GetPartialsPinned(instance, buf, scaleIndex, &pinned, &n);
    if (rc == BEAGLE_ERROR_NO_IMPLEMENTATION) {
        const Dims d = dimsOf(instance);
        DblArr a(env, out, d.ok ? d.C * d.P * d.S : -1, OUT); SHORT_CHECK(a);
        const int rc2 = beagleGetPartials(instance, buf, scaleIndex, a); commitTo(rc2, a); return rc2;
    }
    if (rc == BEAGLE_SUCCESS && out) {
        if ((long)jni::GetArrayLength(env, out) < n) return BEAGLE_ERROR_OUT_OF_RANGE;
        jni::SetDoubleArrayRegion(env, out, 0, (jsize)n, pinned);
    }
    return rc;
}
JNI_FN(jint, getLogScaleFactors)(JNIEnv* env, jobject, jint instance, jint scaleIndex, jdoubleArray out) {
    const Dims d = dimsOf(instance);
    DblArr a(env, out, d.ok ? d.P : -1, OUT); SHORT_CHECK(a);
    const int rc = beagleGetLogScaleFactors(instance, scaleIndex, a); commitTo(rc, a); return rc;
}
JNI_FN(jint, setEigenDecomposition)(JNIEnv* env, jobject, jint instance, jint eigenIndex, jdoubleArray u, jdoubleArray ui, jdoubleArray lam) {
    DblArr a(env, u), b(env, ui), c(env, lam); return beagleSetEigenDecomposition(instance, eigenIndex, a, b, c);
}
JNI_FN(jint, setStateFrequencies)(JNIEnv* env, jobject, jint instance, jint idx, jdoubleArray f) {
    DblArr a(env, f); return beagleSetStateFrequencies(instance, idx, a);
}
JNI_FN(jint, setCategoryWeights)(JNIEnv* env, jobject, jint instance, jint idx, jdoubleArray w) {
    DblArr a(env, w); return beagleSetCategoryWeights(instance, idx, a);
}
JNI_FN(jint, setCategoryRates)(JNIEnv* env, jobject, jint instance, jdoubleArray r) {
    DblArr a(env, r); return beagleSetCategoryRates(instance, a);
}
JNI_FN(jint, setCategoryRatesWithIndex)(JNIEnv* env, jobject, jint instance, jint idx, jdoubleArray r) {
    DblArr a(env, r); return beagleSetCategoryRatesWithIndex(instance, idx, a);
}
JNI_FN(jint, setTransitionMatrix)(JNIEnv* env, jobject, jint instance, jint idx, jdoubleArray m, jdouble padded) {
    DblArr a(env, m); return beagleSetTransitionMatrix(instance, idx, a, padded);
}
JNI_FN(jint, setDifferentialMatrix)(JNIEnv* env, jobject, jint instance, jint idx, jdoubleArray m) {
    DblArr a(env, m); return beagleSetDifferentialMatrix(instance, idx, a);
}
JNI_FN(jint, getTransitionMatrix)(JNIEnv* env, jobject, jint instance, jint idx, jdoubleArray out) {
    const Dims d = dimsOf(instance);
    DblArr a(env, out, d.ok ? d.C * d.S * d.S : -1, OUT); SHORT_CHECK(a);
    const int rc = beagleGetTransitionMatrix(instance, idx, a); commitTo(rc, a); return rc;
}
JNI_FN(jint, convolveTransitionMatrices)(JNIEnv* env, jobject, jint instance, jintArray f, jintArray s, jintArray r, jint count) {
    IntArr a(env, f, count), b(env, s, count), c(env, r, count); SHORT_CHECK(a, b, c);
    return beagleConvolveTransitionMatrices(instance, a, b, c, count);
}
JNI_FN(jint, addTransitionMatrices)(JNIEnv* env, jobject, jint instance, jintArray f, jintArray s, jintArray r, jint count) {
    IntArr a(env, f, count), b(env, s, count), c(env, r, count); SHORT_CHECK(a, b, c);
    return beagleAddTransitionMatrices(instance, a, b, c, count);
}
JNI_FN(jint, transposeTransitionMatrices)(JNIEnv* env, jobject, jint instance, jintArray in, jintArray out, jint count) {
    IntArr a(env, in, count), b(env, out, count); SHORT_CHECK(a, b);
    return beagleTransposeTransitionMatrices(instance, a, b, count);
}
JNI_FN(jint, updateTransitionMatrices)(JNIEnv* env, jobject, jint instance, jint eigenIndex, jintArray prob, jintArray d1,
                                       jintArray d2, jdoubleArray lengths, jint count) {
    IntArr a(env, prob, count), b(env, d1, count), c(env, d2, count); DblArr t(env, lengths, count);
    SHORT_CHECK(a, b, c, t);
    return beagleUpdateTransitionMatrices(instance, eigenIndex, a, b, c, t, count);
}
JNI_FN(jint, updateTransitionMatricesWithMultipleModels)(JNIEnv* env, jobject, jint instance, jintArray eigen, jintArray rates,
                                                         jintArray prob, jintArray d1, jintArray d2, jdoubleArray lengths, jint count) {
    IntArr e(env, eigen, count), r(env, rates, count), a(env, prob, count), b(env, d1, count), c(env, d2, count); DblArr t(env, lengths, count);
    SHORT_CHECK(e, r, a, b, c, t);
    return beagleUpdateTransitionMatricesWithMultipleModels(instance, e, r, a, b, c, t, count);
}
JNI_FN(jint, updatePrePartials)(JNIEnv* env, jobject, jint instance, jintArray ops, jint count, jint cum) {
    IntArr a(env, ops, 7L * count); SHORT_CHECK(a);
    return beagleUpdatePrePartials(instance, a, count, cum);
}
JNI_FN(jint, updatePrePartialsByPartition)(JNIEnv* env, jobject, jint instance, jintArray ops, jint count) {
    IntArr a(env, ops, 9L * count); SHORT_CHECK(a);
    return beagleUpdatePrePartialsByPartition(instance, a, count);
}
JNI_FN(jint, updatePartials)(JNIEnv* env, jobject, jint instance, jintArray ops, jint count, jint cum) {
    IntArr a(env, ops, 7L * count); SHORT_CHECK(a);
    return beagleUpdatePartials(instance, a, count, cum);
}
JNI_FN(jint, updatePartialsByPartition)(JNIEnv* env, jobject, jint instance, jintArray ops, jint count) {
    IntArr a(env, ops, 9L * count); SHORT_CHECK(a);
    return beagleUpdatePartialsByPartition(instance, a, count);
}
JNI_FN(jint, waitForPartials)(JNIEnv* env, jobject, jint instance, jintArray dest, jint count) {
    IntArr a(env, dest, count); SHORT_CHECK(a);
    return beagleWaitForPartials(instance, a, count);
}
JNI_FN(jint, accumulateScaleFactors)(JNIEnv* env, jobject, jint instance, jintArray idx, jint count, jint cum) {
    IntArr a(env, idx, count); SHORT_CHECK(a);
    return beagleAccumulateScaleFactors(instance, a, count, cum);
}
JNI_FN(jint, accumulateScaleFactorsByPartition)(JNIEnv* env, jobject, jint instance, jintArray idx, jint count, jint cum, jint part) {
    IntArr a(env, idx, count); SHORT_CHECK(a);
    return beagleAccumulateScaleFactorsByPartition(instance, a, count, cum, part);
}
JNI_FN(jint, removeScaleFactors)(JNIEnv* env, jobject, jint instance, jintArray idx, jint count, jint cum) {
    IntArr a(env, idx, count); SHORT_CHECK(a);
    return beagleRemoveScaleFactors(instance, a, count, cum);
}
JNI_FN(jint, removeScaleFactorsByPartition)(JNIEnv* env, jobject, jint instance, jintArray idx, jint count, jint cum, jint part) {
    IntArr a(env, idx, count); SHORT_CHECK(a);
    return beagleRemoveScaleFactorsByPartition(instance, a, count, cum, part);
}
JNI_FN(jint, resetScaleFactors)(JNIEnv*, jobject, jint instance, jint cum) { return beagleResetScaleFactors(instance, cum); }
JNI_FN(jint, resetScaleFactorsByPartition)(JNIEnv*, jobject, jint instance, jint cum, jint part) {
    return beagleResetScaleFactorsByPartition(instance, cum, part);
}
JNI_FN(jint, copyScaleFactors)(JNIEnv*, jobject, jint instance, jint dst, jint src) { return beagleCopyScaleFactors(instance, dst, src); }

JNI_FN(jint, calculateRootLogLikelihoods)(JNIEnv* env, jobject, jint instance, jintArray bufs, jintArray weights, jintArray freqs,
                                          jintArray cums, jint count, jdoubleArray outSum) {
    IntArr a(env, bufs, count), b(env, weights, count), c(env, freqs, count), d(env, cums, count); DblArr o(env, outSum, count, OUT);
    SHORT_CHECK(a, b, c, d, o);
    const int rc = beagleCalculateRootLogLikelihoods(instance, a, b, c, d, count, o); commitTo(rc, o); return rc;
}
JNI_FN(jint, calculateRootLogLikelihoodsByPartition)(JNIEnv* env, jobject, jint instance, jintArray bufs, jintArray weights,
                                                     jintArray freqs, jintArray cums, jintArray parts, jint partitionCount,
                                                     jint count, jdoubleArray outByPartition, jdoubleArray outSum) {
    const long n = (long)partitionCount * count;
    IntArr a(env, bufs, n), b(env, weights, n), c(env, freqs, n), d(env, cums, n), p(env, parts, partitionCount);
    DblArr o1(env, outByPartition, n, OUT), o2(env, outSum, count, OUT);
    SHORT_CHECK(a, b, c, d, p, o1, o2);
    const int rc = beagleCalculateRootLogLikelihoodsByPartition(instance, a, b, c, d, p, partitionCount, count, o1, o2);
    commitTo(rc, o1, o2); return rc;
}
JNI_FN(jint, getSiteLogLikelihoods)(JNIEnv* env, jobject, jint instance, jdoubleArray out) {
    const double* pinned = nullptr; long n = 0;
    const int rc = beagleMi355GetSiteLogLikelihoodsPinned(instance, &pinned, &n);
    if (rc == BEAGLE_ERROR_NO_IMPLEMENTATION) {
        const Dims d = dimsOf(instance);
        DblArr o(env, out, d.ok ? d.P : -1, OUT); SHORT_CHECK(o);
        const int rc2 = beagleGetSiteLogLikelihoods(instance, o); commitTo(rc2, o); return rc2;
    }
    if (rc == BEAGLE_SUCCESS && out) {
        if ((long)jni::GetArrayLength(env, out) < n) return BEAGLE_ERROR_OUT_OF_RANGE;
        jni::SetDoubleArrayRegion(env, out, 0, (jsize)n, pinned);
    }
    return rc;
}

// gradient entry points.  BEAST passes null for outDerivatives and, on the second-derivative call, for
// outSumSquaredDerivatives (AbstractBeagleBranchGradientDelegate.java:82-92): IntArr/DblArr turn a null array into nullptr.
JNI_FN(jint, calculateEdgeDifferentials)(JNIEnv* env, jobject, jint instance, jintArray post, jintArray pre, jintArray dmat,
                                         jintArray weights, jint count, jdoubleArray outDeriv, jdoubleArray outSum,
                                         jdoubleArray outSumSquared) {
    IntArr a(env, post, count), b(env, pre, count), c(env, dmat, count), w(env, weights);
    const Dims dm = dimsOf(instance);
    DblArr o0(env, outDeriv, dm.ok ? (long)count * dm.P : -1, OUT), o1(env, outSum, count, OUT), o2(env, outSumSquared, count, OUT);
    SHORT_CHECK(a, b, c, o0, o1, o2);
    const int rc = beagleCalculateEdgeDifferentials(instance, a, b, c, w, count, o0, o1, o2); commitTo(rc, o0, o1, o2); return rc;
}
JNI_FN(jint, calculateCrossProductDifferentials)(JNIEnv* env, jobject, jint instance, jintArray post, jintArray pre, jintArray rates,
                                                 jintArray weights, jdoubleArray lengths, jint count, jdoubleArray outSum,
                                                 jdoubleArray outSumSquared) {
    IntArr a(env, post, count), b(env, pre, count), r(env, rates), w(env, weights);
    DblArr t(env, lengths, count), o1(env, outSum, -1, INOUT), o2(env, outSumSquared, -1, INOUT);      // the sums are ADDED to what the arrays hold
    SHORT_CHECK(a, b, t);
    const int rc = beagleCalculateCrossProductDifferentials(instance, a, b, r, w, t, count, o1, o2); commitTo(rc, o1, o2); return rc;
}
JNI_FN(jint, calculateEdgeDerivative)(JNIEnv*, jobject, jint, jintArray, jintArray, jint, jintArray, jintArray, jint, jint, jint,
                                      jintArray, jint, jdoubleArray, jdoubleArray) { return BEAGLE_ERROR_NO_IMPLEMENTATION; }
