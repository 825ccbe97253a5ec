// jni_min.h — minimal, self-authored subset of the Java Native Interface, written from the public JNI
// specification (the build image has no JDK and therefore no <jni.h>; SURVEY 8b "JNI header").
//
// A JNIEnv* points to a pointer to the JVM's function table; native code calls entry k as
// ((Fn)(*env)[k])(env, ...).  The slot numbers below are the 0-based positions of the functions in the
// specification's "Interface Function Table" (slots 0-3 are reserved).  Only what the shim uses is named.
// RE-VERIFY against a real <jni.h> the first time a JDK is available — the shim checks the
// table version at load (JNI_OnLoad) but cannot check slot positions without a JVM.
#pragma once
#include <stdint.h>

typedef int32_t  jint;
typedef int64_t  jlong;
typedef double   jdouble;
typedef uint8_t  jboolean;
typedef jint     jsize;
typedef void*    jobject;
typedef jobject  jclass;
typedef jobject  jstring;
typedef jobject  jarray;
typedef jarray   jintArray;
typedef jarray   jdoubleArray;
typedef jarray   jobjectArray;
typedef void*    jmethodID;

typedef const void* const* JNIFunctionTable;   // the table: an array of function pointers
typedef JNIFunctionTable JNIEnv;                // C view of JNIEnv: `JNIEnv* env`, table = *env

#define JNIEXPORT __attribute__((visibility("default")))
#define JNICALL
#define JNI_ABORT 2
#define JNI_VERSION_1_6 0x00010006

enum JniSlot {
    JNI_GetVersion = 4,
    JNI_FindClass = 6,
    JNI_ExceptionClear = 17,
    JNI_DeleteLocalRef = 23,
    JNI_NewObject = 28,
    JNI_GetObjectClass = 31,
    JNI_GetMethodID = 33,
    JNI_CallVoidMethod = 61,
    JNI_NewStringUTF = 167,
    JNI_GetArrayLength = 171,
    JNI_NewObjectArray = 172,
    JNI_SetObjectArrayElement = 174,
    JNI_GetIntArrayElements = 187,
    JNI_GetDoubleArrayElements = 190,
    JNI_ReleaseIntArrayElements = 195,
    JNI_ReleaseDoubleArrayElements = 198,
    JNI_GetIntArrayRegion = 203,
    JNI_GetDoubleArrayRegion = 206,
    JNI_SetIntArrayRegion = 211,
    JNI_SetDoubleArrayRegion = 214,
    JNI_ExceptionCheck = 228,
};

namespace jni {
template <typename Fn> inline Fn fn(JNIEnv* env, int slot) { return reinterpret_cast<Fn>(const_cast<void*>((*env)[slot])); }

inline jclass FindClass(JNIEnv* e, const char* name) { return fn<jclass (*)(JNIEnv*, const char*)>(e, JNI_FindClass)(e, name); }
inline jclass GetObjectClass(JNIEnv* e, jobject o) { return fn<jclass (*)(JNIEnv*, jobject)>(e, JNI_GetObjectClass)(e, o); }
inline jmethodID GetMethodID(JNIEnv* e, jclass c, const char* n, const char* sig) {
    return fn<jmethodID (*)(JNIEnv*, jclass, const char*, const char*)>(e, JNI_GetMethodID)(e, c, n, sig);
}
inline jstring NewStringUTF(JNIEnv* e, const char* s) { return fn<jstring (*)(JNIEnv*, const char*)>(e, JNI_NewStringUTF)(e, s); }
inline jsize GetArrayLength(JNIEnv* e, jarray a) { return fn<jsize (*)(JNIEnv*, jarray)>(e, JNI_GetArrayLength)(e, a); }
inline jobjectArray NewObjectArray(JNIEnv* e, jsize n, jclass c, jobject init) {
    return fn<jobjectArray (*)(JNIEnv*, jsize, jclass, jobject)>(e, JNI_NewObjectArray)(e, n, c, init);
}
inline void SetObjectArrayElement(JNIEnv* e, jobjectArray a, jsize i, jobject v) {
    fn<void (*)(JNIEnv*, jobjectArray, jsize, jobject)>(e, JNI_SetObjectArrayElement)(e, a, i, v);
}
inline jint* GetIntArrayElements(JNIEnv* e, jintArray a) {
    return fn<jint* (*)(JNIEnv*, jintArray, jboolean*)>(e, JNI_GetIntArrayElements)(e, a, nullptr);
}
inline jdouble* GetDoubleArrayElements(JNIEnv* e, jdoubleArray a) {
    return fn<jdouble* (*)(JNIEnv*, jdoubleArray, jboolean*)>(e, JNI_GetDoubleArrayElements)(e, a, nullptr);
}
inline void ReleaseIntArrayElements(JNIEnv* e, jintArray a, jint* p, jint mode) {
    fn<void (*)(JNIEnv*, jintArray, jint*, jint)>(e, JNI_ReleaseIntArrayElements)(e, a, p, mode);
}
inline void ReleaseDoubleArrayElements(JNIEnv* e, jdoubleArray a, jdouble* p, jint mode) {
    fn<void (*)(JNIEnv*, jdoubleArray, jdouble*, jint)>(e, JNI_ReleaseDoubleArrayElements)(e, a, p, mode);
}
inline void GetIntArrayRegion(JNIEnv* e, jintArray a, jsize start, jsize len, jint* buf) {
    fn<void (*)(JNIEnv*, jintArray, jsize, jsize, jint*)>(e, JNI_GetIntArrayRegion)(e, a, start, len, buf);
}
inline void GetDoubleArrayRegion(JNIEnv* e, jdoubleArray a, jsize start, jsize len, jdouble* buf) {
    fn<void (*)(JNIEnv*, jdoubleArray, jsize, jsize, jdouble*)>(e, JNI_GetDoubleArrayRegion)(e, a, start, len, buf);
}
inline void SetIntArrayRegion(JNIEnv* e, jintArray a, jsize start, jsize len, const jint* buf) {
    fn<void (*)(JNIEnv*, jintArray, jsize, jsize, const jint*)>(e, JNI_SetIntArrayRegion)(e, a, start, len, buf);
}
inline void SetDoubleArrayRegion(JNIEnv* e, jdoubleArray a, jsize start, jsize len, const jdouble* buf) {
    fn<void (*)(JNIEnv*, jdoubleArray, jsize, jsize, const jdouble*)>(e, JNI_SetDoubleArrayRegion)(e, a, start, len, buf);
}
inline jboolean ExceptionCheck(JNIEnv* e) { return fn<jboolean (*)(JNIEnv*)>(e, JNI_ExceptionCheck)(e); }
inline void ExceptionClear(JNIEnv* e) { fn<void (*)(JNIEnv*)>(e, JNI_ExceptionClear)(e); }
inline void DeleteLocalRef(JNIEnv* e, jobject o) { fn<void (*)(JNIEnv*, jobject)>(e, JNI_DeleteLocalRef)(e, o); }
// variadic entries
typedef jobject (*NewObjectFn)(JNIEnv*, jclass, jmethodID, ...);
typedef void (*CallVoidMethodFn)(JNIEnv*, jobject, jmethodID, ...);
inline NewObjectFn NewObject(JNIEnv* e) { return fn<NewObjectFn>(e, JNI_NewObject); }
inline CallVoidMethodFn CallVoidMethod(JNIEnv* e) { return fn<CallVoidMethodFn>(e, JNI_CallVoidMethod); }
}  // namespace jni
