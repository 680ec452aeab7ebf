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

#define H(h) ((bbduk_handle*)(intptr_t)(h))
#define PIN(a) ((a) ? (*env)->GetPrimitiveArrayCritical(env, (a), NULL) : NULL)
#define UNPIN_IN(a, p)  do { if (a) (*env)->ReleasePrimitiveArrayCritical(env, (a), (p), JNI_ABORT); } while (0)   /* inputs: no copy-back */
#define UNPIN_OUT(a, p) do { if (a) (*env)->ReleasePrimitiveArrayCritical(env, (a), (p), 0); } while (0)           /* outputs: copy back   */

JNIEXPORT jlong JNICALL Java_bbduk_BBDukGpu_createJNI(JNIEnv* env, jclass cls, jintArray ip, jlong middleMask, jfloatArray fp) {
    if ((*env)->GetArrayLength(env, ip) < 26 || (*env)->GetArrayLength(env, fp) < 3) return BBDUK_ERR_ARG;
    jint* v = (jint*)PIN(ip);
    jfloat* f = (jfloat*)PIN(fp);
    bbduk_params p;
    memset(&p, 0, sizeof p);
    p.abi_version = BBDUK_ABI_VERSION;
    p.mode = v[0]; p.k = v[1]; p.mink = v[2]; p.rcomp = v[3]; p.forbidNs = v[4]; p.minlen = v[5]; p.minlen2 = v[6];
    p.middleMask = (int64_t)middleMask;
    p.qhdist = v[7]; p.qhdist2 = v[8]; p.maxBadKmers = v[9]; p.minReadLength = v[10];
    p.removePairsIfEitherBad = v[11]; p.trimPad = v[12]; p.ktrimExclusive = v[13];
    p.restrictLeft = v[14]; p.restrictRight = v[15]; p.skipR1 = v[16]; p.skipR2 = v[17]; p.numScaffolds = v[18]; p.device = v[19];
    p.trimPairsEvenly = v[20]; p.qSkip = v[21]; p.speed = v[22]; p.kbig = v[23]; p.findBestMatch = v[24]; p.kmaskFullyCovered = v[25];
    p.minLenFraction = f[0]; p.minKmerFraction = f[1]; p.minCoveredFraction = f[2];
    UNPIN_IN(fp, f);
    UNPIN_IN(ip, v);
    bbduk_handle* h = NULL;
    const int rc = bbduk_create(&p, &h);
    return rc == BBDUK_OK ? (jlong)(intptr_t)h : (jlong)rc;
}

JNIEXPORT jint JNICALL Java_bbduk_BBDukGpu_uploadWayJNI(JNIEnv* env, jclass cls, jlong h, jint way, jint prime,
        jlongArray keys, jintArray values, jlongArray vkeys, jintArray vvals) {
    const jint nc = (*env)->GetArrayLength(env, keys);
    const jint nv = vkeys ? (*env)->GetArrayLength(env, vkeys) : 0;
    jlong* k = (jlong*)PIN(keys);
    jint*  v = (jint*)PIN(values);
    jlong* vk = nv ? (jlong*)PIN(vkeys) : NULL;
    jint*  vv = nv ? (jint*)PIN(vvals) : NULL;
    const jint rc = bbduk_upload_table_way(H(h), way, prime, (const int64_t*)k, (const int32_t*)v, nc, (const int64_t*)vk, (const int32_t*)vv, nv);
    if (nv) { UNPIN_IN(vvals, vv); UNPIN_IN(vkeys, vk); }
    UNPIN_IN(values, v);
    UNPIN_IN(keys, k);
    return rc;
}

JNIEXPORT jint JNICALL Java_bbduk_BBDukGpu_uploadPairsJNI(JNIEnv* env, jclass cls, jlong h, jlongArray keys, jintArray values) {
    const jint n = (*env)->GetArrayLength(env, keys);
    jlong* k = (jlong*)PIN(keys);
    jint*  v = (jint*)PIN(values);
    const jint rc = bbduk_upload_pairs(H(h), (const int64_t*)k, (const int32_t*)v, n);
    UNPIN_IN(values, v);
    UNPIN_IN(keys, k);
    return rc;
}

JNIEXPORT jint JNICALL Java_bbduk_BBDukGpu_finalizeJNI(JNIEnv* env, jclass cls, jlong h) { return bbduk_finalize_table(H(h)); }

JNIEXPORT jint JNICALL Java_bbduk_BBDukGpu_buildTableJNI(JNIEnv* env, jclass cls, jlong h, jbyteArray refs, jlongArray refOffsets,
        jint nRefs, jint hdist, jint hdist2) {
    jbyte* r = (jbyte*)PIN(refs);
    jlong* o = (jlong*)PIN(refOffsets);
    const jint rc = bbduk_build_table_device(H(h), (const uint8_t*)r, (const int64_t*)o, nRefs, hdist, hdist2);
    UNPIN_IN(refOffsets, o);
    UNPIN_IN(refs, r);
    return rc;
}

/* every batch operator: two or three input arrays, three to five output arrays */
typedef struct { jarray a; void* p; } pinned;
static void pin_all(JNIEnv* env, pinned* x, int n) { for (int i = 0; i < n; i++) x[i].p = PIN(x[i].a); }
static void unpin_all(JNIEnv* env, pinned* x, int n, int first_out) {
    for (int i = n - 1; i >= 0; i--) { if (i >= first_out) UNPIN_OUT(x[i].a, x[i].p); else UNPIN_IN(x[i].a, x[i].p); }
}

static jint batch(JNIEnv* env, jlong h, int kfilter, jbyteArray bases, jlongArray offsets, jint n, jboolean paired,
                  jintArray outA, jintArray outId, jbyteArray outFlags) {
    pinned x[5] = {{bases, 0}, {offsets, 0}, {outA, 0}, {outId, 0}, {outFlags, 0}};
    pin_all(env, x, 5);
    const jint rc = kfilter
        ? bbduk_kfilter_batch(H(h), (const uint8_t*)x[0].p, (const int64_t*)x[1].p, n, paired, (int32_t*)x[2].p, (int32_t*)x[3].p, (uint8_t*)x[4].p)
        : bbduk_ktrim_batch(H(h), (const uint8_t*)x[0].p, (const int64_t*)x[1].p, n, paired, (int32_t*)x[2].p, (int32_t*)x[3].p, (uint8_t*)x[4].p);
    unpin_all(env, x, 5, 2);
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

static jint batch_packed(JNIEnv* env, jlong h, int kfilter, jintArray codes, jintArray undef, jlongArray offsets, jint n, jboolean paired,
                         jintArray outA, jintArray outId, jbyteArray outFlags) {
    pinned x[6] = {{codes, 0}, {undef, 0}, {offsets, 0}, {outA, 0}, {outId, 0}, {outFlags, 0}};
    pin_all(env, x, 6);
    const jint rc = kfilter
        ? bbduk_kfilter_batch_packed(H(h), (const uint32_t*)x[0].p, (const uint32_t*)x[1].p, (const int64_t*)x[2].p, n, paired, (int32_t*)x[3].p, (int32_t*)x[4].p, (uint8_t*)x[5].p)
        : bbduk_ktrim_batch_packed(H(h), (const uint32_t*)x[0].p, (const uint32_t*)x[1].p, (const int64_t*)x[2].p, n, paired, (int32_t*)x[3].p, (int32_t*)x[4].p, (uint8_t*)x[5].p);
    unpin_all(env, x, 6, 3);
    return rc;
}
JNIEXPORT jint JNICALL Java_bbduk_BBDukGpu_ktrimBatchPackedJNI(JNIEnv* env, jclass cls, jlong h, jintArray codes, jintArray undef, jlongArray offsets,
        jint n, jboolean paired, jintArray outTrimmed, jintArray outId0, jbyteArray outFlags) {
    return batch_packed(env, h, 0, codes, undef, offsets, n, paired, outTrimmed, outId0, outFlags);
}
JNIEXPORT jint JNICALL Java_bbduk_BBDukGpu_kfilterBatchPackedJNI(JNIEnv* env, jclass cls, jlong h, jintArray codes, jintArray undef, jlongArray offsets,
        jint n, jboolean paired, jintArray outFound, jintArray outId, jbyteArray outFlags) {
    return batch_packed(env, h, 1, codes, undef, offsets, n, paired, outFound, outId, outFlags);
}

JNIEXPORT jint JNICALL Java_bbduk_BBDukGpu_kmaskBatchJNI(JNIEnv* env, jclass cls, jlong h, jbyteArray bases, jlongArray offsets, jint n, jboolean paired,
        jintArray outMasked, jintArray outId0, jbyteArray outFlags, jintArray outMask) {
    pinned x[6] = {{bases, 0}, {offsets, 0}, {outMasked, 0}, {outId0, 0}, {outFlags, 0}, {outMask, 0}};
    pin_all(env, x, 6);
    const jint rc = bbduk_kmask_batch(H(h), (const uint8_t*)x[0].p, (const int64_t*)x[1].p, n, paired, (int32_t*)x[2].p, (int32_t*)x[3].p, (uint8_t*)x[4].p, (uint32_t*)x[5].p);
    unpin_all(env, x, 6, 2);
    return rc;
}

JNIEXPORT jint JNICALL Java_bbduk_BBDukGpu_ktrimTipsBatchJNI(JNIEnv* env, jclass cls, jlong h, jbyteArray bases, jlongArray offsets, jint n, jboolean paired,
        jintArray outRight, jintArray outLeft, jintArray outId0, jbyteArray outFlags) {
    pinned x[6] = {{bases, 0}, {offsets, 0}, {outRight, 0}, {outLeft, 0}, {outId0, 0}, {outFlags, 0}};
    pin_all(env, x, 6);
    const jint rc = bbduk_ktrimtips_batch(H(h), (const uint8_t*)x[0].p, (const int64_t*)x[1].p, n, paired, (int32_t*)x[2].p, (int32_t*)x[3].p, (int32_t*)x[4].p, (uint8_t*)x[5].p);
    unpin_all(env, x, 6, 2);
    return rc;
}

JNIEXPORT jint JNICALL Java_bbduk_BBDukGpu_ksplitBatchJNI(JNIEnv* env, jclass cls, jlong h, jbyteArray bases, jlongArray offsets, jint n,
        jintArray outTrimmed, jintArray outLeftmost, jintArray outRightmost, jintArray outId0, jbyteArray outFlags) {
    pinned x[7] = {{bases, 0}, {offsets, 0}, {outTrimmed, 0}, {outLeftmost, 0}, {outRightmost, 0}, {outId0, 0}, {outFlags, 0}};
    pin_all(env, x, 7);
    const jint rc = bbduk_ksplit_batch(H(h), (const uint8_t*)x[0].p, (const int64_t*)x[1].p, n, (int32_t*)x[2].p, (int32_t*)x[3].p, (int32_t*)x[4].p, (int32_t*)x[5].p, (uint8_t*)x[6].p);
    unpin_all(env, x, 7, 2);
    return rc;
}

JNIEXPORT jint JNICALL Java_bbduk_BBDukGpu_kfilterBatchMatchesJNI(JNIEnv* env, jclass cls, jlong h, jbyteArray bases, jlongArray offsets, jint n,
        jboolean paired, jintArray outFound, jintArray outId, jbyteArray outFlags, jint maxIds, jintArray outNids, jintArray outMatchIds, jintArray outMatchCounts) {
    pinned x[8] = {{bases, 0}, {offsets, 0}, {outFound, 0}, {outId, 0}, {outFlags, 0}, {outNids, 0}, {outMatchIds, 0}, {outMatchCounts, 0}};
    pin_all(env, x, 8);
    const jint rc = bbduk_kfilter_batch_matches(H(h), (const uint8_t*)x[0].p, (const int64_t*)x[1].p, n, paired ? 1 : 0, (int32_t*)x[2].p, (int32_t*)x[3].p,
                                                (uint8_t*)x[4].p, maxIds, (int32_t*)x[5].p, (int32_t*)x[6].p, (int32_t*)x[7].p);
    unpin_all(env, x, 8, 2);
    return rc;
}

JNIEXPORT jint JNICALL Java_bbduk_BBDukGpu_countersJNI(JNIEnv* env, jclass cls, jlong h, jlongArray out) {
    const jint n = (*env)->GetArrayLength(env, out);
    jlong* o = (jlong*)PIN(out);
    const jint rc = bbduk_get_counters(H(h), (int64_t*)o, n);
    UNPIN_OUT(out, o);
    return rc;
}

JNIEXPORT void JNICALL Java_bbduk_BBDukGpu_destroyJNI(JNIEnv* env, jclass cls, jlong h) { bbduk_destroy(H(h)); }
