/*
 * JNI shim: bbduk.BBDukGpu natives -> the C ABI of include/bbduk_gpu.h.  C99; the conventions of the reference's
 * jni/BBMergeOverlapper.c (static natives, jint status, caller-allocated outputs, no exceptions, no callbacks, no retained
 * references) with ONE deliberate difference: the batch operators take DIRECT java.nio buffers, not arrays.
 *
 * Why: BBMergeOverlapper.c:505-519 pins its arrays with GetPrimitiveArrayCritical around a few microseconds of CPU work.  A batch
 * operator here takes a staging slot (condition variable), copies over PCIe and waits for the GPU: the JNI specification forbids
 * blocking inside a critical region, and with every BBDuk worker thread parked in one the collector would be locked out JVM-wide
 * (GCLocker stalls, possible deadlock).  Direct buffers have a stable address outside the Java heap, so nothing is pinned and
 * nothing is copied: the Java side concatenates a batch's Read.bases straight into a buffer from BBDukGpu.allocPinned (page-locked
 * host memory: the DMA engines read it without an intermediate copy), and reads the results from direct IntBuffers.
 * The table uploads (once per run) copy array regions instead (Get<Type>ArrayRegion, chunked): no critical region either.
 *
 * NOT compiled in this repository (no jni.h in the build image).  Build where a JDK exists:
 *   gcc -O3 -std=c99 -fPIC -shared -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -I../../include \
 *       BBDukGpu.c -L../../bbtools_amd -lbbduk_hip -o libbbduk_jni.so
 */
#include <jni.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "bbduk_gpu.h"

#define H(h) ((bbduk_handle*)(intptr_t)(h))

/* address of a direct buffer that must hold at least `need` bytes (NULL buffer allowed when need == 0) */
static void* dbuf(JNIEnv* env, jobject buf, jlong need, int* bad) {
    if (!buf) { if (need > 0) *bad = 1; return NULL; }
    void* p = (*env)->GetDirectBufferAddress(env, buf);
    if (!p || (*env)->GetDirectBufferCapacity(env, buf) < need) { *bad = 1; return NULL; }
    return p;
}

JNIEXPORT jlong JNICALL Java_bbduk_BBDukGpu_createJNI(JNIEnv* env, jclass cls, jintArray ip, jlong middleMask, jfloatArray fp) {
    jint v[27]; jfloat f[3];
    if ((*env)->GetArrayLength(env, ip) < 27 || (*env)->GetArrayLength(env, fp) < 3) return BBDUK_ERR_ARG;
    (*env)->GetIntArrayRegion(env, ip, 0, 27, v);
    (*env)->GetFloatArrayRegion(env, fp, 0, 3, f);
    bbduk_params p;
    memset(&p, 0, sizeof p);
    p.abi_version = BBDUK_ABI_VERSION;
    p.mode = v[0]; p.k = v[1]; p.mink = v[2]; p.rcomp = v[3]; p.forbidNs = v[4]; p.minlen = v[5]; p.minlen2 = v[6];
    p.middleMask = (int64_t)middleMask;
    p.qhdist = v[7]; p.qhdist2 = v[8]; p.maxBadKmers = v[9]; p.minReadLength = v[10];
    p.removePairsIfEitherBad = v[11]; p.trimPad = v[12]; p.ktrimExclusive = v[13];
    p.restrictLeft = v[14]; p.restrictRight = v[15]; p.skipR1 = v[16]; p.skipR2 = v[17]; p.numScaffolds = v[18]; p.device = v[19];
    p.trimPairsEvenly = v[20]; p.qSkip = v[21]; p.speed = v[22]; p.kbig = v[23]; p.findBestMatch = v[24]; p.kmaskFullyCovered = v[25];
    p.trimFailuresTo1bp = v[26];
    p.minLenFraction = f[0]; p.minKmerFraction = f[1]; p.minCoveredFraction = f[2];
    bbduk_handle* h = NULL;
    const int rc = bbduk_create(&p, &h);
    return rc == BBDUK_OK ? (jlong)(intptr_t)h : (jlong)rc;
}

/* page-locked host memory as a direct ByteBuffer (order it LITTLE_ENDIAN on the Java side before taking Int/Long views) */
JNIEXPORT jobject JNICALL Java_bbduk_BBDukGpu_allocPinnedJNI(JNIEnv* env, jclass cls, jlong bytes) {
    void* p = NULL;
    if (bytes <= 0 || bbduk_pinned_malloc(bytes, &p) != BBDUK_OK) return NULL;
    return (*env)->NewDirectByteBuffer(env, p, bytes);
}
JNIEXPORT void JNICALL Java_bbduk_BBDukGpu_freePinnedJNI(JNIEnv* env, jclass cls, jobject buf) {
    if (buf) bbduk_pinned_free((*env)->GetDirectBufferAddress(env, buf));
}

/* table uploads: array regions through a bounded bounce buffer (16 M cells per step) */
#define CHUNK (1 << 24)
static jint upload_cells(JNIEnv* env, jlong h, jlongArray keys, jintArray values) {
    const jsize n = keys ? (*env)->GetArrayLength(env, keys) : 0;
    if (n == 0) return BBDUK_OK;
    const jsize cap = n < CHUNK ? n : CHUNK;
    jlong* k = (jlong*)malloc((size_t)cap * sizeof(jlong)); jint* v = (jint*)malloc((size_t)cap * sizeof(jint));
    jint rc = (k && v) ? BBDUK_OK : BBDUK_ERR_NOMEM;
    for (jsize q = 0; q < n && rc == BBDUK_OK; q += cap) {
        const jsize m = n - q < cap ? n - q : cap;
        (*env)->GetLongArrayRegion(env, keys, q, m, k);
        (*env)->GetIntArrayRegion(env, values, q, m, v);
        /* a HashArray1D image: cells with key -1 are empty (kmer/AbstractKmerTable.java:807); bbduk_upload_table_way skips them */
        rc = bbduk_upload_table_way(H(h), 0, 0, (const int64_t*)k, (const int32_t*)v, m, NULL, NULL, 0);
    }
    free(k); free(v);
    return rc;
}
JNIEXPORT jint JNICALL Java_bbduk_BBDukGpu_uploadWayJNI(JNIEnv* env, jclass cls, jlong h, jint way, jint prime,
        jlongArray keys, jintArray values, jlongArray vkeys, jintArray vvals) {
    (void)way; (void)prime;                /* the device re-hashes: the geometry of the Java image is not needed */
    jint rc = upload_cells(env, h, keys, values);
    if (rc == BBDUK_OK) rc = upload_cells(env, h, vkeys, vvals);
    return rc;
}
JNIEXPORT jint JNICALL Java_bbduk_BBDukGpu_uploadPairsJNI(JNIEnv* env, jclass cls, jlong h, jlongArray keys, jintArray values) {
    return upload_cells(env, h, keys, values);
}
JNIEXPORT jint JNICALL Java_bbduk_BBDukGpu_finalizeJNI(JNIEnv* env, jclass cls, jlong h) { return bbduk_finalize_table(H(h)); }

/* refs: direct buffer with the scaffolds' bases concatenated; refOffsets: direct buffer of nRefs+1 little-endian longs */
JNIEXPORT jint JNICALL Java_bbduk_BBDukGpu_buildTableJNI(JNIEnv* env, jclass cls, jlong h, jobject refs, jobject refOffsets,
        jint nRefs, jint hdist, jint hdist2) {
    int bad = 0;
    const int64_t* o = (const int64_t*)dbuf(env, refOffsets, (jlong)(nRefs + 1) * 8, &bad);
    if (bad || nRefs < 0) return BBDUK_ERR_ARG;
    const uint8_t* r = (const uint8_t*)dbuf(env, refs, o[nRefs], &bad);
    if (bad) return BBDUK_ERR_ARG;
    return bbduk_build_table_device(H(h), r, o, nRefs, hdist, hdist2);
}

/* ---- batch operators.  bases: n reads' bases concatenated; offsets: n+1 longs; outputs: n ints / n bytes each (direct buffers,
 * native byte order).  The call blocks until the results are in the output buffers; two threads may submit to one handle at once
 * (its two staging slots overlap one call's copies with the other's kernel). */
static jint batch(JNIEnv* env, jlong h, int kfilter, jobject bases, jobject offsets, jint n, jboolean paired, jobject outA, jobject outId, jobject outFlags) {
    int bad = 0;
    if (n < 0) return BBDUK_ERR_ARG;
    const int64_t* off = (const int64_t*)dbuf(env, offsets, ((jlong)n + 1) * 8, &bad);
    if (bad) return BBDUK_ERR_ARG;
    const uint8_t* b = (const uint8_t*)dbuf(env, bases, off[n], &bad);
    int32_t* a = (int32_t*)dbuf(env, outA, (jlong)n * 4, &bad); int32_t* id = (int32_t*)dbuf(env, outId, (jlong)n * 4, &bad);
    uint8_t* fl = (uint8_t*)dbuf(env, outFlags, n, &bad);
    if (bad) return BBDUK_ERR_ARG;
    return kfilter ? bbduk_kfilter_batch(H(h), b, off, n, paired, a, id, fl) : bbduk_ktrim_batch(H(h), b, off, n, paired, a, id, fl);
}
JNIEXPORT jint JNICALL Java_bbduk_BBDukGpu_ktrimBatchJNI(JNIEnv* env, jclass cls, jlong h, jobject bases, jobject offsets,
        jint n, jboolean paired, jobject outTrimmed, jobject outId0, jobject outFlags) {
    return batch(env, h, 0, bases, offsets, n, paired, outTrimmed, outId0, outFlags);
}
JNIEXPORT jint JNICALL Java_bbduk_BBDukGpu_kfilterBatchJNI(JNIEnv* env, jclass cls, jlong h, jobject bases, jobject offsets,
        jint n, jboolean paired, jobject outFound, jobject outId, jobject outFlags) {
    return batch(env, h, 1, bases, offsets, n, paired, outFound, outId, outFlags);
}

static jint batch_packed(JNIEnv* env, jlong h, int kfilter, jobject codes, jobject undef, jobject offsets, jint n, jboolean paired,
                         jobject outA, jobject outId, jobject outFlags) {
    int bad = 0;
    if (n < 0) return BBDUK_ERR_ARG;
    const int64_t* off = (const int64_t*)dbuf(env, offsets, ((jlong)n + 1) * 8, &bad);
    if (bad) return BBDUK_ERR_ARG;
    const uint32_t* c = (const uint32_t*)dbuf(env, codes, ((off[n] + 15) / 16) * 4, &bad);
    const uint32_t* u = (const uint32_t*)dbuf(env, undef, ((off[n] + 31) / 32) * 4, &bad);
    int32_t* a = (int32_t*)dbuf(env, outA, (jlong)n * 4, &bad); int32_t* id = (int32_t*)dbuf(env, outId, (jlong)n * 4, &bad);
    uint8_t* fl = (uint8_t*)dbuf(env, outFlags, n, &bad);
    if (bad) return BBDUK_ERR_ARG;
    return kfilter ? bbduk_kfilter_batch_packed(H(h), c, u, off, n, paired, a, id, fl) : bbduk_ktrim_batch_packed(H(h), c, u, off, n, paired, a, id, fl);
}
JNIEXPORT jint JNICALL Java_bbduk_BBDukGpu_ktrimBatchPackedJNI(JNIEnv* env, jclass cls, jlong h, jobject codes, jobject undef, jobject offsets,
        jint n, jboolean paired, jobject outTrimmed, jobject outId0, jobject outFlags) {
    return batch_packed(env, h, 0, codes, undef, offsets, n, paired, outTrimmed, outId0, outFlags);
}
JNIEXPORT jint JNICALL Java_bbduk_BBDukGpu_kfilterBatchPackedJNI(JNIEnv* env, jclass cls, jlong h, jobject codes, jobject undef, jobject offsets,
        jint n, jboolean paired, jobject outFound, jobject outId, jobject outFlags) {
    return batch_packed(env, h, 1, codes, undef, offsets, n, paired, outFound, outId, outFlags);
}

JNIEXPORT jint JNICALL Java_bbduk_BBDukGpu_kmaskBatchJNI(JNIEnv* env, jclass cls, jlong h, jobject bases, jobject offsets, jint n, jboolean paired,
        jobject outMasked, jobject outId0, jobject outFlags, jobject outMask) {
    int bad = 0;
    if (n < 0) return BBDUK_ERR_ARG;
    const int64_t* off = (const int64_t*)dbuf(env, offsets, ((jlong)n + 1) * 8, &bad);
    if (bad) return BBDUK_ERR_ARG;
    const uint8_t* b = (const uint8_t*)dbuf(env, bases, off[n], &bad);
    int32_t* a = (int32_t*)dbuf(env, outMasked, (jlong)n * 4, &bad); int32_t* id = (int32_t*)dbuf(env, outId0, (jlong)n * 4, &bad);
    uint8_t* fl = (uint8_t*)dbuf(env, outFlags, n, &bad);
    uint32_t* m = (uint32_t*)dbuf(env, outMask, ((off[n] + 31) / 32 + 1) * 4, &bad);
    if (bad) return BBDUK_ERR_ARG;
    return bbduk_kmask_batch(H(h), b, off, n, paired, a, id, fl, m);
}

JNIEXPORT jint JNICALL Java_bbduk_BBDukGpu_ktrimTipsBatchJNI(JNIEnv* env, jclass cls, jlong h, jobject bases, jobject offsets, jint n, jboolean paired,
        jobject outRight, jobject outLeft, jobject outId0, jobject outFlags) {
    int bad = 0;
    if (n < 0) return BBDUK_ERR_ARG;
    const int64_t* off = (const int64_t*)dbuf(env, offsets, ((jlong)n + 1) * 8, &bad);
    if (bad) return BBDUK_ERR_ARG;
    const uint8_t* b = (const uint8_t*)dbuf(env, bases, off[n], &bad);
    int32_t* r = (int32_t*)dbuf(env, outRight, (jlong)n * 4, &bad); int32_t* l = (int32_t*)dbuf(env, outLeft, (jlong)n * 4, &bad);
    int32_t* id = (int32_t*)dbuf(env, outId0, (jlong)n * 4, &bad); uint8_t* fl = (uint8_t*)dbuf(env, outFlags, n, &bad);
    if (bad) return BBDUK_ERR_ARG;
    return bbduk_ktrimtips_batch(H(h), b, off, n, paired, r, l, id, fl);
}

JNIEXPORT jint JNICALL Java_bbduk_BBDukGpu_ksplitBatchJNI(JNIEnv* env, jclass cls, jlong h, jobject bases, jobject offsets, jint n,
        jobject outTrimmed, jobject outLeftmost, jobject outRightmost, jobject outId0, jobject outFlags) {
    int bad = 0;
    if (n < 0) return BBDUK_ERR_ARG;
    const int64_t* off = (const int64_t*)dbuf(env, offsets, ((jlong)n + 1) * 8, &bad);
    if (bad) return BBDUK_ERR_ARG;
    const uint8_t* b = (const uint8_t*)dbuf(env, bases, off[n], &bad);
    int32_t* x = (int32_t*)dbuf(env, outTrimmed, (jlong)n * 4, &bad); int32_t* lm = (int32_t*)dbuf(env, outLeftmost, (jlong)n * 4, &bad);
    int32_t* rm = (int32_t*)dbuf(env, outRightmost, (jlong)n * 4, &bad); int32_t* id = (int32_t*)dbuf(env, outId0, (jlong)n * 4, &bad);
    uint8_t* fl = (uint8_t*)dbuf(env, outFlags, n, &bad);
    if (bad) return BBDUK_ERR_ARG;
    return bbduk_ksplit_batch(H(h), b, off, n, x, lm, rm, id, fl);
}

JNIEXPORT jint JNICALL Java_bbduk_BBDukGpu_kfilterBatchMatchesJNI(JNIEnv* env, jclass cls, jlong h, jobject bases, jobject offsets, jint n,
        jboolean paired, jobject outFound, jobject outId, jobject outFlags, jint maxIds, jobject outNids, jobject outMatchIds, jobject outMatchCounts) {
    int bad = 0;
    if (n < 0 || maxIds < 1) return BBDUK_ERR_ARG;
    const int64_t* off = (const int64_t*)dbuf(env, offsets, ((jlong)n + 1) * 8, &bad);
    if (bad) return BBDUK_ERR_ARG;
    const uint8_t* b = (const uint8_t*)dbuf(env, bases, off[n], &bad);
    int32_t* a = (int32_t*)dbuf(env, outFound, (jlong)n * 4, &bad); int32_t* id = (int32_t*)dbuf(env, outId, (jlong)n * 4, &bad);
    uint8_t* fl = (uint8_t*)dbuf(env, outFlags, n, &bad);
    int32_t* ni = (int32_t*)dbuf(env, outNids, (jlong)n * 4, &bad);
    int32_t* mi = (int32_t*)dbuf(env, outMatchIds, (jlong)n * maxIds * 4, &bad); int32_t* mc = (int32_t*)dbuf(env, outMatchCounts, (jlong)n * maxIds * 4, &bad);
    if (bad) return BBDUK_ERR_ARG;
    return bbduk_kfilter_batch_matches(H(h), b, off, n, paired ? 1 : 0, a, id, fl, maxIds, ni, mi, mc);
}

JNIEXPORT jint JNICALL Java_bbduk_BBDukGpu_countersJNI(JNIEnv* env, jclass cls, jlong h, jlongArray out) {
    const jsize n = (*env)->GetArrayLength(env, out);
    int64_t* o = (int64_t*)malloc((size_t)(n > 0 ? n : 1) * 8);
    if (!o) return BBDUK_ERR_NOMEM;
    const jint rc = bbduk_get_counters(H(h), o, n);
    if (rc == BBDUK_OK) (*env)->SetLongArrayRegion(env, out, 0, n, (const jlong*)o);
    free(o);
    return rc;
}

/* one JVM, several GPUs: the handles' counter vectors become their sum (RCCL all-reduce over the distinct devices) */
JNIEXPORT jint JNICALL Java_bbduk_BBDukGpu_allreduceCountersLocalJNI(JNIEnv* env, jclass cls, jlongArray handles) {
    const jsize n = (*env)->GetArrayLength(env, handles);
    if (n < 1 || n > 64) return BBDUK_ERR_ARG;
    jlong hv[64]; bbduk_handle* hs[64];
    (*env)->GetLongArrayRegion(env, handles, 0, n, hv);
    for (jsize i = 0; i < n; i++) hs[i] = H(hv[i]);
    if (bbduk_comm_size(hs[0]) == 0) { const jint rc = bbduk_comm_create_local(hs, n); if (rc != BBDUK_OK) return rc; }
    return bbduk_allreduce_counters_local(hs, n);
}

JNIEXPORT void JNICALL Java_bbduk_BBDukGpu_destroyJNI(JNIEnv* env, jclass cls, jlong h) { bbduk_destroy(H(h)); }
