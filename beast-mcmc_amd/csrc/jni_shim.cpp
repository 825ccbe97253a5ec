// jni_shim.cpp — the 47 `native` methods of lib/beagle.jar!beagle/BeagleJNIWrapper.class as
// Java_beagle_BeagleJNIWrapper_<name> symbols, each forwarding one-to-one to the C ABI of
// include/beagle_mi355.h.  With this file the engine's shared object IS `libhmsbeagle-jni.so`:
// BEAST loads it with System.loadLibrary("hmsbeagle-jni") (BeagleJNIWrapper#loadBeagleLibrary) and
// BeagleTreeLikelihood / TreeDataLikelihood call it unchanged.
//
// Parameter order of every function = the method descriptor in the class file (after JNIEnv*, jobject).
// Java arrays are COPIED with Get<Type>ArrayRegion into library-owned buffers and, when they are outputs, copied back
// with Set<Type>ArrayRegion (no pinning semantics to get wrong, and the JVM never has to hand out — or copy —
// its heap array for the 7 (T-1) ints of an operation list).  null arrays are passed through as NULL
// (HomogenousSubstitutionModelDelegate.java:260-261 passes null derivative indices).
//
// Compiled against a self-authored minimal JNI header (jni_min.h).  The image has no JVM; tests/native/fake_jvm.cpp is a
// JVM-less JNIEnv (229-slot function table over plain C++ objects) that drives these symbols end to end on the GPU box
// (tests/test_gpu_jni_shim.py); INTEGRATION.md §4 lists the on-JVM validation steps.
#include <string.h>

#include <vector>

#include "../../include/beagle_mi355.h"
#include "jni_min.h"

namespace {

struct IntArr {
    JNIEnv* env; jintArray arr; std::vector<jint> buf; bool output;
    IntArr(JNIEnv* e, jintArray a, bool out = false) : env(e), arr(a), output(out) {
        if (a) { buf.resize((size_t)jni::GetArrayLength(e, a)); if (!buf.empty()) jni::GetIntArrayRegion(e, a, 0, (jsize)buf.size(), buf.data()); }
    }
    ~IntArr() { if (arr && output && !buf.empty()) jni::SetIntArrayRegion(env, arr, 0, (jsize)buf.size(), buf.data()); }
    operator int*() { return arr ? buf.data() : nullptr; }
};
struct DblArr {
    JNIEnv* env; jdoubleArray arr; std::vector<jdouble> buf; bool output;
    DblArr(JNIEnv* e, jdoubleArray a, bool out = false) : env(e), arr(a), output(out) {
        if (a) { buf.resize((size_t)jni::GetArrayLength(e, a)); if (!buf.empty()) jni::GetDoubleArrayRegion(e, a, 0, (jsize)buf.size(), buf.data()); }
    }
    ~DblArr() { if (arr && output && !buf.empty()) jni::SetDoubleArrayRegion(env, arr, 0, (jsize)buf.size(), buf.data()); }
    operator double*() { return arr ? buf.data() : nullptr; }
};

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
    IntArr res(env, resourceList);
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
    IntArr res(env, resourceList);
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
    IntArr a(env, states, true); return beagleGetTipStates(instance, tip, a);
}
JNI_FN(jint, setTipPartials)(JNIEnv* env, jobject, jint instance, jint tip, jdoubleArray partials) {
    DblArr a(env, partials); return beagleSetTipPartials(instance, tip, a);
}
JNI_FN(jint, setRootPrePartials)(JNIEnv* env, jobject, jint instance, jintArray bufs, jintArray freqs, jint count) {
    IntArr a(env, bufs), b(env, freqs); return beagleSetRootPrePartials(instance, a, b, count);
}
JNI_FN(jint, setPartials)(JNIEnv* env, jobject, jint instance, jint buf, jdoubleArray partials) {
    DblArr a(env, partials); return beagleSetPartials(instance, buf, a);
}
JNI_FN(jint, getPartials)(JNIEnv* env, jobject, jint instance, jint buf, jint scaleIndex, jdoubleArray out) {
    DblArr a(env, out, true); return beagleGetPartials(instance, buf, scaleIndex, a);
}
JNI_FN(jint, getLogScaleFactors)(JNIEnv* env, jobject, jint instance, jint scaleIndex, jdoubleArray out) {
    DblArr a(env, out, true); return beagleGetLogScaleFactors(instance, scaleIndex, a);
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
    DblArr a(env, out, true); return beagleGetTransitionMatrix(instance, idx, a);
}
JNI_FN(jint, convolveTransitionMatrices)(JNIEnv* env, jobject, jint instance, jintArray f, jintArray s, jintArray r, jint count) {
    IntArr a(env, f), b(env, s), c(env, r); return beagleConvolveTransitionMatrices(instance, a, b, c, count);
}
JNI_FN(jint, addTransitionMatrices)(JNIEnv* env, jobject, jint instance, jintArray f, jintArray s, jintArray r, jint count) {
    IntArr a(env, f), b(env, s), c(env, r); return beagleAddTransitionMatrices(instance, a, b, c, count);
}
JNI_FN(jint, transposeTransitionMatrices)(JNIEnv* env, jobject, jint instance, jintArray in, jintArray out, jint count) {
    IntArr a(env, in), b(env, out); return beagleTransposeTransitionMatrices(instance, a, b, count);
}
JNI_FN(jint, updateTransitionMatrices)(JNIEnv* env, jobject, jint instance, jint eigenIndex, jintArray prob, jintArray d1,
                                       jintArray d2, jdoubleArray lengths, jint count) {
    IntArr a(env, prob), b(env, d1), c(env, d2); DblArr t(env, lengths);
    return beagleUpdateTransitionMatrices(instance, eigenIndex, a, b, c, t, count);
}
JNI_FN(jint, updateTransitionMatricesWithMultipleModels)(JNIEnv* env, jobject, jint instance, jintArray eigen, jintArray rates,
                                                         jintArray prob, jintArray d1, jintArray d2, jdoubleArray lengths, jint count) {
    IntArr e(env, eigen), r(env, rates), a(env, prob), b(env, d1), c(env, d2); DblArr t(env, lengths);
    return beagleUpdateTransitionMatricesWithMultipleModels(instance, e, r, a, b, c, t, count);
}
JNI_FN(jint, updatePrePartials)(JNIEnv* env, jobject, jint instance, jintArray ops, jint count, jint cum) {
    IntArr a(env, ops); return beagleUpdatePrePartials(instance, a, count, cum);
}
JNI_FN(jint, updatePrePartialsByPartition)(JNIEnv* env, jobject, jint instance, jintArray ops, jint count) {
    IntArr a(env, ops); return beagleUpdatePrePartialsByPartition(instance, a, count);
}
JNI_FN(jint, updatePartials)(JNIEnv* env, jobject, jint instance, jintArray ops, jint count, jint cum) {
    IntArr a(env, ops); return beagleUpdatePartials(instance, a, count, cum);
}
JNI_FN(jint, updatePartialsByPartition)(JNIEnv* env, jobject, jint instance, jintArray ops, jint count) {
    IntArr a(env, ops); return beagleUpdatePartialsByPartition(instance, a, count);
}
JNI_FN(jint, waitForPartials)(JNIEnv* env, jobject, jint instance, jintArray dest, jint count) {
    IntArr a(env, dest); return beagleWaitForPartials(instance, a, count);
}
JNI_FN(jint, accumulateScaleFactors)(JNIEnv* env, jobject, jint instance, jintArray idx, jint count, jint cum) {
    IntArr a(env, idx); return beagleAccumulateScaleFactors(instance, a, count, cum);
}
JNI_FN(jint, accumulateScaleFactorsByPartition)(JNIEnv* env, jobject, jint instance, jintArray idx, jint count, jint cum, jint part) {
    IntArr a(env, idx); return beagleAccumulateScaleFactorsByPartition(instance, a, count, cum, part);
}
JNI_FN(jint, removeScaleFactors)(JNIEnv* env, jobject, jint instance, jintArray idx, jint count, jint cum) {
    IntArr a(env, idx); return beagleRemoveScaleFactors(instance, a, count, cum);
}
JNI_FN(jint, removeScaleFactorsByPartition)(JNIEnv* env, jobject, jint instance, jintArray idx, jint count, jint cum, jint part) {
    IntArr a(env, idx); return beagleRemoveScaleFactorsByPartition(instance, a, count, cum, part);
}
JNI_FN(jint, resetScaleFactors)(JNIEnv*, jobject, jint instance, jint cum) { return beagleResetScaleFactors(instance, cum); }
JNI_FN(jint, resetScaleFactorsByPartition)(JNIEnv*, jobject, jint instance, jint cum, jint part) {
    return beagleResetScaleFactorsByPartition(instance, cum, part);
}
JNI_FN(jint, copyScaleFactors)(JNIEnv*, jobject, jint instance, jint dst, jint src) { return beagleCopyScaleFactors(instance, dst, src); }

JNI_FN(jint, calculateRootLogLikelihoods)(JNIEnv* env, jobject, jint instance, jintArray bufs, jintArray weights, jintArray freqs,
                                          jintArray cums, jint count, jdoubleArray outSum) {
    IntArr a(env, bufs), b(env, weights), c(env, freqs), d(env, cums); DblArr o(env, outSum, true);
    return beagleCalculateRootLogLikelihoods(instance, a, b, c, d, count, o);
}
JNI_FN(jint, calculateRootLogLikelihoodsByPartition)(JNIEnv* env, jobject, jint instance, jintArray bufs, jintArray weights,
                                                     jintArray freqs, jintArray cums, jintArray parts, jint partitionCount,
                                                     jint count, jdoubleArray outByPartition, jdoubleArray outSum) {
    IntArr a(env, bufs), b(env, weights), c(env, freqs), d(env, cums), p(env, parts);
    DblArr o1(env, outByPartition, true), o2(env, outSum, true);
    return beagleCalculateRootLogLikelihoodsByPartition(instance, a, b, c, d, p, partitionCount, count, o1, o2);
}
JNI_FN(jint, getSiteLogLikelihoods)(JNIEnv* env, jobject, jint instance, jdoubleArray out) {
    DblArr o(env, out, true); return beagleGetSiteLogLikelihoods(instance, o);
}

// gradient entry points.  BEAST passes null for outDerivatives and, on the second-derivative call, for
// outSumSquaredDerivatives (AbstractBeagleBranchGradientDelegate.java:82-92): IntArr/DblArr turn a null array into nullptr.
JNI_FN(jint, calculateEdgeDifferentials)(JNIEnv* env, jobject, jint instance, jintArray post, jintArray pre, jintArray dmat,
                                         jintArray weights, jint count, jdoubleArray outDeriv, jdoubleArray outSum,
                                         jdoubleArray outSumSquared) {
    IntArr a(env, post), b(env, pre), c(env, dmat), w(env, weights);
    DblArr o0(env, outDeriv, true), o1(env, outSum, true), o2(env, outSumSquared, true);
    return beagleCalculateEdgeDifferentials(instance, a, b, c, w, count, o0, o1, o2);
}
JNI_FN(jint, calculateCrossProductDifferentials)(JNIEnv* env, jobject, jint instance, jintArray post, jintArray pre, jintArray rates,
                                                 jintArray weights, jdoubleArray lengths, jint count, jdoubleArray outSum,
                                                 jdoubleArray outSumSquared) {
    IntArr a(env, post), b(env, pre), r(env, rates), w(env, weights);
    DblArr t(env, lengths), o1(env, outSum, true), o2(env, outSumSquared, true);
    return beagleCalculateCrossProductDifferentials(instance, a, b, r, w, t, count, o1, o2);
}
JNI_FN(jint, calculateEdgeDerivative)(JNIEnv*, jobject, jint, jintArray, jintArray, jint, jintArray, jintArray, jint, jint, jint,
                                      jintArray, jint, jdoubleArray, jdoubleArray) { return BEAGLE_ERROR_NO_IMPLEMENTATION; }
