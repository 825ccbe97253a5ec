// fake_jvm.cpp — a JVM-less JNIEnv for driving libhmsbeagle-jni.so's Java_beagle_BeagleJNIWrapper_* symbols.
// TEST INFRASTRUCTURE (tests/test_gpu_jni_shim.py); the build image and the GPU box have no JDK.
//
// A JNIEnv* is a pointer to a pointer to the JVM's function table.  This program builds such a table — 229 slots, the
// JNI 1.6 "Interface Function Table" — with the entries the shim uses implemented over plain C++ objects (every other
// slot aborts with its number, so a call through a wrong slot cannot pass silently), plays the role of
// beagle.BeagleJNIWrapper's Java callers and prints what it observed for the pytest side to assert:
//   getVersion / getResourceList / getBenchmarkedResourceList -> the Java objects the shim constructed (class, ctor argument,
//       every setter it invoked — checked against the method tables of lib/beagle.jar's classes, so a misspelt setter or a
//       wrong descriptor raises the fake NoSuchMethodError)
//   the beagle.jar smoke test (BeagleFactory#main: 3 taxa, JC69, "PAUP logL = -1574.63623") through createInstance ->
//       setTipStates ... updateTransitionMatrices (null derivative arrays) -> updatePartials -> calculateRootLogLikelihoods.
// Slot numbers here are written down from the JNI specification independently of csrc/jni_min.h.
#include <dlfcn.h>
#include <math.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <set>
#include <string>
#include <vector>

typedef int32_t jint; typedef int64_t jlong; typedef double jdouble; typedef uint8_t jboolean; typedef jint jsize;

struct FObj {
    std::string cls;                       // "java/lang/String", "[I", "[D", "[L", or a beagle class
    std::string str;
    std::vector<jint> ints; std::vector<jdouble> dbls; std::vector<FObj*> elems;
    jint ctorArg = 0;
    std::map<std::string, std::string> calls;     // setter name -> value as text
};
struct FMethod { std::string cls, name, sig; };
typedef const void* const* FEnv;

static bool g_pending = false;
static int g_exceptions = 0;
static std::map<std::string, std::set<std::string>> g_classes = {     // method tables of lib/beagle.jar (name + descriptor)
    {"beagle/ResourceDetails", {"<init>(I)V", "setName(Ljava/lang/String;)V", "setDescription(Ljava/lang/String;)V", "setFlags(J)V"}},
    {"beagle/InstanceDetails", {"<init>()V", "setResourceNumber(I)V", "setFlags(J)V", "setResourceName(Ljava/lang/String;)V",
                                "setImplementationName(Ljava/lang/String;)V"}},
    {"beagle/BenchmarkedResourceDetails", {"<init>(I)V", "setResourceNumber(I)V", "setName(Ljava/lang/String;)V",
                                           "setDescription(Ljava/lang/String;)V", "setSupportFlags(J)V", "setRequiredFlags(J)V",
                                           "setReturnCode(I)V", "setImplName(Ljava/lang/String;)V", "setBenchedFlags(J)V",
                                           "setBenchmarkResult(D)V", "setPerformanceRatio(D)V"}},
};

static void trap(int slot) { fprintf(stderr, "fake JVM: unimplemented JNI slot %d was called\n", slot); abort(); }
#define TRAP(n) static void trap##n() { trap(n); }

static jint f_GetVersion(FEnv*) { return 0x00010006; }
static FObj* f_FindClass(FEnv*, const char* name) {
    if (!g_classes.count(name)) { g_pending = true; g_exceptions++; return nullptr; }
    FObj* c = new FObj(); c->cls = "java/lang/Class"; c->str = name; return c;
}
static void f_ExceptionClear(FEnv*) { g_pending = false; }
static jboolean f_ExceptionCheck(FEnv*) { return g_pending ? 1 : 0; }
static void f_DeleteLocalRef(FEnv*, FObj*) {}
static FObj* f_GetObjectClass(FEnv*, FObj* o) { FObj* c = new FObj(); c->cls = "java/lang/Class"; c->str = o->cls; return c; }
static FMethod* f_GetMethodID(FEnv*, FObj* cls, const char* name, const char* sig) {
    auto it = g_classes.find(cls->str);
    if (it == g_classes.end() || !it->second.count(std::string(name) + sig)) { g_pending = true; g_exceptions++; return nullptr; }
    return new FMethod{cls->str, name, sig};
}
static FObj* f_NewObject(FEnv*, FObj* cls, FMethod* ctor, ...) {
    FObj* o = new FObj(); o->cls = cls->str;
    va_list ap; va_start(ap, ctor);
    if (ctor->sig == "(I)V") o->ctorArg = va_arg(ap, jint);
    va_end(ap);
    return o;
}
static void f_CallVoidMethod(FEnv*, FObj* obj, FMethod* m, ...) {
    va_list ap; va_start(ap, m);
    char buf[64];
    if (m->sig == "(I)V") { snprintf(buf, sizeof buf, "%d", va_arg(ap, jint)); obj->calls[m->name] = buf; }
    else if (m->sig == "(J)V") { snprintf(buf, sizeof buf, "%lld", (long long)va_arg(ap, jlong)); obj->calls[m->name] = buf; }
    else if (m->sig == "(D)V") { snprintf(buf, sizeof buf, "%.17g", va_arg(ap, jdouble)); obj->calls[m->name] = buf; }
    else { FObj* s = va_arg(ap, FObj*); obj->calls[m->name] = s ? s->str : "(null)"; }
    va_end(ap);
}
static FObj* f_NewStringUTF(FEnv*, const char* s) { FObj* o = new FObj(); o->cls = "java/lang/String"; o->str = s ? s : ""; return o; }
static jsize f_GetArrayLength(FEnv*, FObj* a) { return (jsize)(a->cls == "[I" ? a->ints.size() : a->cls == "[D" ? a->dbls.size() : a->elems.size()); }
static FObj* f_NewObjectArray(FEnv*, jsize n, FObj* cls, FObj*) { FObj* a = new FObj(); a->cls = "[L"; a->str = cls->str; a->elems.assign(n, nullptr); return a; }
static void f_SetObjectArrayElement(FEnv*, FObj* a, jsize i, FObj* v) { a->elems[i] = v; }
// bytes the shim moves between the "Java heap" and its own buffers (printed per call by the driver below: the shim must copy
// what a call uses — count-derived lengths in, outputs out only — not whole arrays)
static long g_bytesIn = 0, g_bytesOut = 0;
static void f_GetIntArrayRegion(FEnv*, FObj* a, jsize s, jsize n, jint* buf) { memcpy(buf, a->ints.data() + s, (size_t)n * sizeof(jint)); g_bytesIn += (long)n * 4; }
static void f_GetDoubleArrayRegion(FEnv*, FObj* a, jsize s, jsize n, jdouble* buf) { memcpy(buf, a->dbls.data() + s, (size_t)n * sizeof(jdouble)); g_bytesIn += (long)n * 8; }
static void f_SetIntArrayRegion(FEnv*, FObj* a, jsize s, jsize n, const jint* buf) { memcpy(a->ints.data() + s, buf, (size_t)n * sizeof(jint)); g_bytesOut += (long)n * 4; }
static void f_SetDoubleArrayRegion(FEnv*, FObj* a, jsize s, jsize n, const jdouble* buf) { memcpy(a->dbls.data() + s, buf, (size_t)n * sizeof(jdouble)); g_bytesOut += (long)n * 8; }
static void traffic(const char* call) { printf("traffic %s in=%ld out=%ld\n", call, g_bytesIn, g_bytesOut); g_bytesIn = g_bytesOut = 0; }

// JNI 1.6 Interface Function Table: 0-3 reserved, 4 GetVersion, 5 DefineClass, 6 FindClass, ... 17 ExceptionClear, ...
// 23 DeleteLocalRef, ... 28 NewObject, ... 31 GetObjectClass, 32 IsInstanceOf, 33 GetMethodID, ... 61 CallVoidMethod, ...
// 167 NewStringUTF, ... 171 GetArrayLength, 172 NewObjectArray, 173 GetObjectArrayElement, 174 SetObjectArrayElement, ...
// 199-206 Get<T>ArrayRegion (Boolean, Byte, Char, Short, Int = 203, Long, Float, Double = 206),
// 207-214 Set<T>ArrayRegion (Int = 211, Double = 214), ... 228 ExceptionCheck.
static const void* g_table[229];
template <int N> static void trapN() { trap(N); }
template <int N> struct Fill { static void go() { g_table[N] = (const void*)&trapN<N>; Fill<N - 1>::go(); } };
template <> struct Fill<-1> { static void go() {} };

static FObj* ints(std::vector<jint> v) { FObj* a = new FObj(); a->cls = "[I"; a->ints = std::move(v); return a; }
static FObj* dbls(std::vector<jdouble> v) { FObj* a = new FObj(); a->cls = "[D"; a->dbls = std::move(v); return a; }

static void* g_lib;
template <typename Fn> static Fn sym(const char* name) {
    std::string s = std::string("Java_beagle_BeagleJNIWrapper_") + name;
    void* p = dlsym(g_lib, s.c_str());
    if (!p) { fprintf(stderr, "missing symbol %s\n", s.c_str()); exit(2); }
    return (Fn)p;
}
static void dumpObject(const char* what, int i, const FObj* o) {
    printf("%s[%d] class=%s ctor=%d", what, i, o->cls.c_str(), o->ctorArg);
    for (auto& kv : o->calls) printf(" %s=<%s>", kv.first.c_str(), kv.second.c_str());
    printf("\n");
}

int main(int argc, char** argv) {
    if (argc < 3) { fprintf(stderr, "usage: fake_jvm <libhmsbeagle-jni.so> <smoke.txt>\n"); return 2; }
    Fill<228>::go();
    g_table[4] = (const void*)f_GetVersion; g_table[6] = (const void*)f_FindClass; g_table[17] = (const void*)f_ExceptionClear;
    g_table[23] = (const void*)f_DeleteLocalRef; g_table[28] = (const void*)f_NewObject; g_table[31] = (const void*)f_GetObjectClass;
    g_table[33] = (const void*)f_GetMethodID; g_table[61] = (const void*)f_CallVoidMethod; g_table[167] = (const void*)f_NewStringUTF;
    g_table[171] = (const void*)f_GetArrayLength; g_table[172] = (const void*)f_NewObjectArray; g_table[174] = (const void*)f_SetObjectArrayElement;
    g_table[203] = (const void*)f_GetIntArrayRegion; g_table[206] = (const void*)f_GetDoubleArrayRegion;
    g_table[211] = (const void*)f_SetIntArrayRegion; g_table[214] = (const void*)f_SetDoubleArrayRegion; g_table[228] = (const void*)f_ExceptionCheck;
    FEnv envTable = g_table;
    FEnv* env = &envTable;
    g_lib = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
    if (!g_lib) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 2; }

    // ---- the smoke fixture: nSites, then 3 rows of states, 16 U, 16 Uinv, 4 lambda, 4 edge lengths
    FILE* fp = fopen(argv[2], "r");
    int n = 0;
    if (!fp || fscanf(fp, "%d", &n) != 1) return 2;
    std::vector<std::vector<jint>> rows(3, std::vector<jint>(n));
    for (auto& r : rows) for (auto& v : r) if (fscanf(fp, "%d", &v) != 1) return 2;
    auto readD = [&](int k) { std::vector<jdouble> v(k); for (auto& x : v) if (fscanf(fp, "%lf", &x) != 1) exit(2); return v; };
    std::vector<jdouble> U = readD(16), Ui = readD(16), lam = readD(4), edges = readD(4);
    fclose(fp);

    FObj* self = new FObj();
    FObj* ver = sym<FObj* (*)(FEnv*, FObj*)>("getVersion")(env, self);
    printf("version=%s\n", ver->str.c_str());
    FObj* rl = sym<FObj* (*)(FEnv*, FObj*)>("getResourceList")(env, self);
    printf("resources=%zu elementClass=%s\n", rl->elems.size(), rl->str.c_str());
    for (size_t i = 0; i < rl->elems.size(); i++) dumpObject("resource", (int)i, rl->elems[i]);

    FObj* details = new FObj(); details->cls = "beagle/InstanceDetails";
    typedef jint (*CreateFn)(FEnv*, FObj*, jint, jint, jint, jint, jint, jint, jint, jint, jint, FObj*, jint, jlong, jlong, FObj*);
    const jint h = sym<CreateFn>("createInstance")(env, self, 3, 10, 3, 4, n, 1, 4, 1, 3, ints({1}), 1, 0, 0, details);
    printf("instance=%d\n", h);
    dumpObject("details", 0, details);
    if (h < 0) return 1;
    typedef jint (*IIA)(FEnv*, FObj*, jint, jint, FObj*);
    typedef jint (*IA)(FEnv*, FObj*, jint, FObj*);
    jint rc = 0;
    for (int t = 0; t < 3; t++) rc |= sym<IIA>("setTipStates")(env, self, h, t, ints(rows[t]));
    rc |= sym<IA>("setPatternWeights")(env, self, h, dbls(std::vector<jdouble>(n, 1.0)));
    rc |= sym<IIA>("setStateFrequencies")(env, self, h, 0, dbls({0.25, 0.25, 0.25, 0.25}));
    rc |= sym<IIA>("setCategoryWeights")(env, self, h, 0, dbls({1.0}));
    rc |= sym<IA>("setCategoryRates")(env, self, h, dbls({1.0}));
    typedef jint (*EigFn)(FEnv*, FObj*, jint, jint, FObj*, FObj*, FObj*);
    rc |= sym<EigFn>("setEigenDecomposition")(env, self, h, 0, dbls(U), dbls(Ui), dbls(lam));
    typedef jint (*UtmFn)(FEnv*, FObj*, jint, jint, FObj*, FObj*, FObj*, FObj*, jint);
    // the edge-length array is longer than `count`, the derivative index arrays are null (HomogenousSubstitutionModelDelegate.java:260-261)
    std::vector<jdouble> longEdges = edges; longEdges.resize(9, 123.0);
    traffic("setup");
    rc |= sym<UtmFn>("updateTransitionMatrices")(env, self, h, 0, ints({0, 1, 2, 3}), nullptr, nullptr, dbls(longEdges), 4);
    traffic("updateTransitionMatrices");
    typedef jint (*UpFn)(FEnv*, FObj*, jint, FObj*, jint, jint);
    std::vector<jint> ops = {3, -1, -1, 0, 0, 1, 1, 4, -1, -1, 2, 2, 3, 3};
    ops.resize(21, 0);                                                       // operations[] is sized internalNodeCount * 7 by BEAST
    rc |= sym<UpFn>("updatePartials")(env, self, h, ints(ops), 2, -1);
    traffic("updatePartials");
    typedef jint (*RootFn)(FEnv*, FObj*, jint, FObj*, FObj*, FObj*, FObj*, jint, FObj*);
    FObj* out = dbls({0.0});
    const jint rcRoot = sym<RootFn>("calculateRootLogLikelihoods")(env, self, h, ints({4}), ints({0}), ints({0}), ints({-1}), 1, out);
    traffic("calculateRootLogLikelihoods");
    printf("rc=%d rootRc=%d lnL=%.10f\n", rc, rcRoot, out->dbls[0]);
    FObj* site = dbls(std::vector<jdouble>(n, 0.0));
    rc = sym<IA>("getSiteLogLikelihoods")(env, self, h, site);
    traffic("getSiteLogLikelihoods");
    {   // getPartials (III[D)I of the root buffer: an output array, nothing goes in
        typedef jint (*GpFn)(FEnv*, FObj*, jint, jint, jint, FObj*);
        FObj* part = dbls(std::vector<jdouble>((size_t)n * 4, -1.0));
        const jint rcp = sym<GpFn>("getPartials")(env, self, h, 4, -1, part);
        traffic("getPartials");
        double lnl = 0.0;
        for (int p = 0; p < n; p++) lnl += log(0.25 * (part->dbls[4 * p] + part->dbls[4 * p + 1] + part->dbls[4 * p + 2] + part->dbls[4 * p + 3]));
        printf("getPartialsRc=%d lnLfromPartials=%.10f\n", rcp, lnl);
    }
    double s = 0.0; for (double v : site->dbls) s += v;
    printf("siteRc=%d siteSum=%.10f\n", rc, s);
    FObj* tips = ints(std::vector<jint>(n, -7));
    rc = sym<IIA>("getTipStates")(env, self, h, 1, tips);
    traffic("getTipStates");
    int same = 0; for (int i = 0; i < n; i++) same += tips->ints[i] == rows[1][i];
    printf("getTipStatesRc=%d matching=%d of %d\n", rc, same, n);
    {   // error behaviour of output arrays: a failing call leaves the Java array as it was; an array shorter than what the call
        // defines is refused; a longer one keeps its tail
        FObj* keep = dbls(std::vector<jdouble>(n + 3, 123.0));
        const jint rcBad = sym<IIA>("getLogScaleFactors")(env, self, h, 99, keep);                // scale index out of range
        int untouched = 0; for (double v : keep->dbls) untouched += v == 123.0;
        FObj* shortTips = ints(std::vector<jint>(n - 1, -7));
        const jint rcShort = sym<IIA>("getTipStates")(env, self, h, 1, shortTips);
        int shortUntouched = 0; for (int v : shortTips->ints) shortUntouched += v == -7;
        FObj* longTips = ints(std::vector<jint>(n + 2, -7));
        const jint rcLong = sym<IIA>("getTipStates")(env, self, h, 1, longTips);
        printf("outputOnError rc=%d untouched=%d of %d | shortArray rc=%d untouched=%d of %d | longArray rc=%d tail=%d,%d first=%d\n", rcBad, untouched, n + 3,
               rcShort, shortUntouched, n - 1, rcLong, longTips->ints[n], longTips->ints[n + 1], longTips->ints[0] == rows[1][0]);
        g_bytesIn = g_bytesOut = 0;
    }
    rc = sym<jint (*)(FEnv*, FObj*, jint)>("finalize")(env, self, h);
    printf("finalizeRc=%d\n", rc);

    typedef FObj* (*BenchFn)(FEnv*, FObj*, jint, jint, jint, jint, jint, FObj*, jint, jlong, jlong, jint, jint, jint, jlong);
    FObj* bl = sym<BenchFn>("getBenchmarkedResourceList")(env, self, 16, 16, 4, 1000, 4, nullptr, 0, 0, 0, 1, 1, 0, 4);
    printf("benchmarked=%zu elementClass=%s\n", bl ? bl->elems.size() : (size_t)0, bl ? bl->str.c_str() : "");
    for (size_t i = 0; bl && i < bl->elems.size(); i++) dumpObject("benchmarked", (int)i, bl->elems[i]);
    printf("exceptionsRaised=%d pending=%d\n", g_exceptions, (int)g_pending);
    return 0;
}
