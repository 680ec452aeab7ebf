/* Syntax-check stand-in for <jni.h>: TEST INFRASTRUCTURE ONLY (tests/test_jni_shims.py compiles the files of integration/jni with -fsyntax-only).
 * The image has no JDK.  Declares, with the signatures of the Java Native Interface specification, exactly the types and the
 * JNIEnv members the shims use, so a typo, a wrong argument count or a wrong pointer type in a shim fails a CPU test.  Nothing links
 * against it and nothing ships with it; a real build uses $JAVA_HOME/include/jni.h. */
#ifndef BBTOOLS_AMD_JNI_STUB_H
#define BBTOOLS_AMD_JNI_STUB_H
#include <stdint.h>
#define JNIEXPORT __attribute__((visibility("default")))
#define JNICALL
#define JNI_ABORT 2
typedef int32_t jint; typedef int64_t jlong; typedef int8_t jbyte; typedef uint8_t jboolean; typedef float jfloat; typedef double jdouble;
typedef jint jsize;
struct _jobject; typedef struct _jobject* jobject;
typedef jobject jclass; typedef jobject jstring; typedef jobject jarray;
typedef jarray jbyteArray; typedef jarray jintArray; typedef jarray jlongArray; typedef jarray jfloatArray;
struct JNINativeInterface_; typedef const struct JNINativeInterface_* JNIEnv;
struct JNINativeInterface_ {
    jsize (*GetArrayLength)(JNIEnv*, jarray);
    void (*GetByteArrayRegion)(JNIEnv*, jbyteArray, jsize, jsize, jbyte*);
    void (*GetIntArrayRegion)(JNIEnv*, jintArray, jsize, jsize, jint*);
    void (*GetLongArrayRegion)(JNIEnv*, jlongArray, jsize, jsize, jlong*);
    void (*GetFloatArrayRegion)(JNIEnv*, jfloatArray, jsize, jsize, jfloat*);
    void (*SetIntArrayRegion)(JNIEnv*, jintArray, jsize, jsize, const jint*);
    void (*SetLongArrayRegion)(JNIEnv*, jlongArray, jsize, jsize, const jlong*);
    void* (*GetPrimitiveArrayCritical)(JNIEnv*, jarray, jboolean*);
    void (*ReleasePrimitiveArrayCritical)(JNIEnv*, jarray, void*, jint);
    void* (*GetDirectBufferAddress)(JNIEnv*, jobject);
    jlong (*GetDirectBufferCapacity)(JNIEnv*, jobject);
    jobject (*NewDirectByteBuffer)(JNIEnv*, void*, jlong);
    jstring (*NewStringUTF)(JNIEnv*, const char*);
};
#endif
