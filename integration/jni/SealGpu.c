/*
 * JNI shim: jgi.SealGpu natives -> the C ABI of include/seal_gpu.h.  Same conventions as BBDukGpu.c beside it (static natives, jint
 * status, direct buffers for the batch so that no critical region is held while the GPU works, array regions for the one-off uploads).
 * NOT compiled in this repository (no jni.h in the build image); build it into libbbduk_jni.so together with BBDukGpu.c.
 */
#include <jni.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "seal_gpu.h"

#define H(h) ((seal_handle*)(intptr_t)(h))
static void* dbuf(JNIEnv* env, jobject buf, jlong need, int* bad) {
    if (!buf) { if (need > 0) *bad = 1; return NULL; }
    void* p = (*env)->GetDirectBufferAddress(env, buf);
    if (!p || (*env)->GetDirectBufferCapacity(env, buf) < need) { *bad = 1; return NULL; }
    return p;
}

JNIEXPORT jlong JNICALL Java_jgi_SealGpu_createJNI(JNIEnv* env, jclass cls, jintArray ip, jfloatArray fp) {
    jint v[21]; jfloat f[3];
    if ((*env)->GetArrayLength(env, ip) < 21 || (*env)->GetArrayLength(env, fp) < 3) return -1;
    (*env)->GetIntArrayRegion(env, ip, 0, 21, v); (*env)->GetFloatArrayRegion(env, fp, 0, 3, f);
    seal_params p; seal_default_params(&p);
    p.k = v[0]; p.maskMiddle = v[1]; p.midMaskLen = v[2]; p.rcomp = v[3]; p.forbidNs = v[4]; p.hdist = v[5]; p.refSkip = v[6];
    p.restrictLeft = v[7]; p.restrictRight = v[8]; p.qSkip = v[9]; p.speed = v[10]; p.matchMode = v[11]; p.ambigMode = v[12];
    p.keepPairsTogether = v[13]; p.minKmerHits = v[14]; p.clearzone = v[15]; p.minReadLength = v[16]; p.maxReadLength = v[17];
    p.requireBothBad = v[18]; p.maxScaffolds = v[19]; p.device = v[20];
    p.minKmerFraction = f[0]; p.minLenFraction = f[1]; p.clearzoneFraction = f[2];
    seal_handle* h = NULL;
    const int rc = seal_create(&p, &h);
    return rc == 0 ? (jlong)(intptr_t)h : (jlong)rc;
}
JNIEXPORT jint JNICALL Java_jgi_SealGpu_addRefSequenceJNI(JNIEnv* env, jclass cls, jlong h, jbyteArray bases) {
    const jsize n = bases ? (*env)->GetArrayLength(env, bases) : 0;
    jbyte* b = (jbyte*)malloc((size_t)(n > 0 ? n : 1));
    if (!b) return -3;
    if (n > 0) (*env)->GetByteArrayRegion(env, bases, 0, n, b);
    const int rc = seal_add_ref_sequence(H(h), (const uint8_t*)b, n, NULL);
    free(b);
    return rc;
}
#define CHUNK (1 << 24)
JNIEXPORT jint JNICALL Java_jgi_SealGpu_uploadPairsJNI(JNIEnv* env, jclass cls, jlong h, jlongArray keys, jintArray ids) {
    const jsize n = keys ? (*env)->GetArrayLength(env, keys) : 0;
    if (n == 0) return 0;
    const jsize cap = n < CHUNK ? n : CHUNK;
    jlong* k = (jlong*)malloc((size_t)cap * sizeof(jlong)); jint* v = (jint*)malloc((size_t)cap * sizeof(jint));
    int rc = (k && v) ? 0 : -3;
    for (jsize o = 0; rc == 0 && o < n; o += cap) {
        const jsize m = n - o < cap ? n - o : cap;
        (*env)->GetLongArrayRegion(env, keys, o, m, k); (*env)->GetIntArrayRegion(env, ids, o, m, v);
        rc = seal_upload_pairs(H(h), (const int64_t*)k, (const int32_t*)v, m);
    }
    free(k); free(v);
    return rc;
}
JNIEXPORT jint JNICALL Java_jgi_SealGpu_finalizeJNI(JNIEnv* env, jclass cls, jlong h) { return seal_finalize(H(h)); }
JNIEXPORT jint JNICALL Java_jgi_SealGpu_batchJNI(JNIEnv* env, jclass cls, jlong h, jobject bases, jobject offsets, jint n, jboolean paired,
        jlong firstNumericID, jint maxIds, jobject outSites, jobject outAssigned, jobject outMax, jobject outIds, jobject outFlags) {
    int bad = 0;
    const int64_t* off = (const int64_t*)dbuf(env, offsets, 8 * ((jlong)n + 1), &bad);
    if (bad || n < 0) return -1;
    const int64_t total = n > 0 ? off[n] : 0;
    const uint8_t* b = (const uint8_t*)dbuf(env, bases, total, &bad);
    int32_t* s = (int32_t*)dbuf(env, outSites, 4 * (jlong)n, &bad); int32_t* a = (int32_t*)dbuf(env, outAssigned, 4 * (jlong)n, &bad);
    int32_t* m = (int32_t*)dbuf(env, outMax, 4 * (jlong)n, &bad); int32_t* i = (int32_t*)dbuf(env, outIds, 4 * (jlong)n * maxIds, &bad);
    uint8_t* f = (uint8_t*)dbuf(env, outFlags, n, &bad);
    if (bad) return -1;
    return seal_batch(H(h), b, off, n, paired ? 1 : 0, firstNumericID, maxIds, s, a, m, i, f);
}
JNIEXPORT jlong JNICALL Java_jgi_SealGpu_countersLenJNI(JNIEnv* env, jclass cls, jlong h) { return seal_counters_len(H(h)); }
JNIEXPORT jint JNICALL Java_jgi_SealGpu_readCountersJNI(JNIEnv* env, jclass cls, jlong h, jlongArray out) {
    const jlong n = seal_counters_len(H(h));
    if (n < 0 || (*env)->GetArrayLength(env, out) < n) return -1;
    int64_t* tmp = (int64_t*)malloc((size_t)n * 8);
    if (!tmp) return -3;
    const int rc = seal_read_counters(H(h), tmp);
    if (rc == 0) (*env)->SetLongArrayRegion(env, out, 0, (jsize)n, (const jlong*)tmp);
    free(tmp);
    return rc;
}
JNIEXPORT jstring JNICALL Java_jgi_SealGpu_lastErrorJNI(JNIEnv* env, jclass cls, jlong h) { return (*env)->NewStringUTF(env, seal_last_error(H(h))); }
JNIEXPORT void JNICALL Java_jgi_SealGpu_destroyJNI(JNIEnv* env, jclass cls, jlong h) { seal_destroy(H(h)); }
