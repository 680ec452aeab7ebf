/*
 * JNI shim: bbduk.BBDukGpu natives -> the C ABI of include/bbduk_gpu.h.  C99, same conventions as the reference's
 * jni/BBMergeOverlapper.c:505-519 (GetPrimitiveArrayCritical; inputs released JNI_ABORT, outputs released 0; the
 * result is the jint return value plus caller-allocated arrays; no exceptions, no callbacks, no retained references).
 * NOT compiled in this repository (no jni.h in the build image).  Build where a JDK exists:
 *   gcc -O3 -std=c99 -fPIC -shared -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -I../../include \
 *       BBDukGpu.c -L../../bbtools_amd -lbbduk_hip -o libbbduk_jni.so
 */
#include <jni.h>
#include <stdint.h>
#include <string.h>
#include "bbduk_gpu.h"

JNIEXPORT jlong JNICALL Java_bbduk_BBDukGpu_createJNI(JNIEnv* env, jclass cls, jintArray ip, jlong middleMask, jfloat minLenFraction) {
    if ((*env)->GetArrayLength(env, ip) < 20) return BBDUK_ERR_ARG;
    jint* v = (jint*)(*env)->GetPrimitiveArrayCritical(env, ip, NULL);
    bbduk_params p;
    memset(&p, 0, sizeof p);
    p.abi_version = BBDUK_ABI_VERSION;
    p.mode = v[0]; p.k = v[1]; p.mink = v[2]; p.rcomp = v[3]; p.forbidNs = v[4]; p.minlen = v[5]; p.minlen2 = v[6];
    p.middleMask = (int64_t)middleMask;
    p.qhdist = v[7]; p.qhdist2 = v[8]; p.maxBadKmers = v[9]; p.minReadLength = v[10]; p.minLenFraction = minLenFraction;
    p.removePairsIfEitherBad = v[11]; p.trimPad = v[12]; p.ktrimExclusive = v[13];
    p.restrictLeft = v[14]; p.restrictRight = v[15]; p.skipR1 = v[16]; p.skipR2 = v[17]; p.numScaffolds = v[18]; p.device = v[19];
    (*env)->ReleasePrimitiveArrayCritical(env, ip, v, JNI_ABORT);
    bbduk_handle* h = NULL;
    const int rc = bbduk_create(&p, &h);
    return rc == BBDUK_OK ? (jlong)(intptr_t)h : (jlong)rc;
}

JNIEXPORT jint JNICALL Java_bbduk_BBDukGpu_uploadWayJNI(JNIEnv* env, jclass cls, jlong h, jint way, jint prime,
        jlongArray keys, jintArray values, jlongArray vkeys, jintArray vvals) {
    const jint nc = (*env)->GetArrayLength(env, keys);
    const jint nv = vkeys ? (*env)->GetArrayLength(env, vkeys) : 0;
    jlong* k = (jlong*)(*env)->GetPrimitiveArrayCritical(env, keys, NULL);
    jint*  v = (jint*)(*env)->GetPrimitiveArrayCritical(env, values, NULL);
    jlong* vk = nv ? (jlong*)(*env)->GetPrimitiveArrayCritical(env, vkeys, NULL) : NULL;
    jint*  vv = nv ? (jint*)(*env)->GetPrimitiveArrayCritical(env, vvals, NULL) : NULL;
    const jint rc = bbduk_upload_table_way((bbduk_handle*)(intptr_t)h, way, prime, (const int64_t*)k, (const int32_t*)v, nc,
                                           (const int64_t*)vk, (const int32_t*)vv, nv);
    if (nv) { (*env)->ReleasePrimitiveArrayCritical(env, vvals, vv, JNI_ABORT); (*env)->ReleasePrimitiveArrayCritical(env, vkeys, vk, JNI_ABORT); }
    (*env)->ReleasePrimitiveArrayCritical(env, values, v, JNI_ABORT);
    (*env)->ReleasePrimitiveArrayCritical(env, keys, k, JNI_ABORT);
    return rc;
}

JNIEXPORT jint JNICALL Java_bbduk_BBDukGpu_uploadPairsJNI(JNIEnv* env, jclass cls, jlong h, jlongArray keys, jintArray values) {
    const jint n = (*env)->GetArrayLength(env, keys);
    jlong* k = (jlong*)(*env)->GetPrimitiveArrayCritical(env, keys, NULL);
    jint*  v = (jint*)(*env)->GetPrimitiveArrayCritical(env, values, NULL);
    const jint rc = bbduk_upload_pairs((bbduk_handle*)(intptr_t)h, (const int64_t*)k, (const int32_t*)v, n);
    (*env)->ReleasePrimitiveArrayCritical(env, values, v, JNI_ABORT);
    (*env)->ReleasePrimitiveArrayCritical(env, keys, k, JNI_ABORT);
    return rc;
}

JNIEXPORT jint JNICALL Java_bbduk_BBDukGpu_finalizeJNI(JNIEnv* env, jclass cls, jlong h) {
    return bbduk_finalize_table((bbduk_handle*)(intptr_t)h);
}

static jint batch(JNIEnv* env, jlong h, int kfilter, jbyteArray bases, jlongArray offsets, jint n, jboolean paired,
                  jintArray outA, jintArray outId, jbyteArray outFlags) {
    jbyte* jb = (jbyte*)(*env)->GetPrimitiveArrayCritical(env, bases, NULL);
    jlong* jo = (jlong*)(*env)->GetPrimitiveArrayCritical(env, offsets, NULL);
    jint*  ja = (jint*)(*env)->GetPrimitiveArrayCritical(env, outA, NULL);
    jint*  ji = (jint*)(*env)->GetPrimitiveArrayCritical(env, outId, NULL);
    jbyte* jf = (jbyte*)(*env)->GetPrimitiveArrayCritical(env, outFlags, NULL);
    bbduk_handle* hh = (bbduk_handle*)(intptr_t)h;
    const jint rc = kfilter
        ? bbduk_kfilter_batch(hh, (const uint8_t*)jb, (const int64_t*)jo, n, paired, (int32_t*)ja, (int32_t*)ji, (uint8_t*)jf)
        : bbduk_ktrim_batch(hh, (const uint8_t*)jb, (const int64_t*)jo, n, paired, (int32_t*)ja, (int32_t*)ji, (uint8_t*)jf);
    (*env)->ReleasePrimitiveArrayCritical(env, outFlags, jf, 0);          /* outputs: copy back   */
    (*env)->ReleasePrimitiveArrayCritical(env, outId, ji, 0);
    (*env)->ReleasePrimitiveArrayCritical(env, outA, ja, 0);
    (*env)->ReleasePrimitiveArrayCritical(env, offsets, jo, JNI_ABORT);   /* inputs: no copy-back */
    (*env)->ReleasePrimitiveArrayCritical(env, bases, jb, JNI_ABORT);
    return rc;
}

JNIEXPORT jint JNICALL Java_bbduk_BBDukGpu_ktrimBatchJNI(JNIEnv* env, jclass cls, jlong h, jbyteArray bases, jlongArray offsets,
        jint n, jboolean paired, jintArray outTrimmed, jintArray outId0, jbyteArray outFlags) {
    return batch(env, h, 0, bases, offsets, n, paired, outTrimmed, outId0, outFlags);
}
JNIEXPORT jint JNICALL Java_bbduk_BBDukGpu_kfilterBatchJNI(JNIEnv* env, jclass cls, jlong h, jbyteArray bases, jlongArray offsets,
        jint n, jboolean paired, jintArray outFound, jintArray outId, jbyteArray outFlags) {
    return batch(env, h, 1, bases, offsets, n, paired, outFound, outId, outFlags);
}

JNIEXPORT jint JNICALL Java_bbduk_BBDukGpu_countersJNI(JNIEnv* env, jclass cls, jlong h, jlongArray out) {
    const jint n = (*env)->GetArrayLength(env, out);
    jlong* o = (jlong*)(*env)->GetPrimitiveArrayCritical(env, out, NULL);
    const jint rc = bbduk_get_counters((bbduk_handle*)(intptr_t)h, (int64_t*)o, n);
    (*env)->ReleasePrimitiveArrayCritical(env, out, o, 0);
    return rc;
}

JNIEXPORT void JNICALL Java_bbduk_BBDukGpu_destroyJNI(JNIEnv* env, jclass cls, jlong h) {
    bbduk_destroy((bbduk_handle*)(intptr_t)h);
}
