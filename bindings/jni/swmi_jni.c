/*
 * swmi_jni.c -- thin JNI shim over include/swmi.h for the Java class sw.GpuSmithWaterman
 * (bindings/java/sw/GpuSmithWaterman.java).  It only unwraps JNI types and forwards to swmi_shim.c, which holds the
 * argument checks and the C-ABI call sequence in JNI-free C99 (that part IS compiled and run by this repository's
 * tests: tests/c/shim_kat.c).  This file is NOT compiled in this repository's build: the build image has no JDK (no
 * jni.h).  On a machine with a JDK:
 *     gcc -shared -fPIC -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -Iinclude -Ibindings/jni \
 *         bindings/jni/swmi_jni.c bindings/jni/swmi_shim.c -Lsparksmithwaterman_amd/lib -lswmi -o libswmi_jni.so
 *
 * Replaces the per-pair call  new SmithWaterman.OptAlignments().call(seqs, alignScores, alignTypes)
 * at src/sw/Distribution.java:421-422 by ONE native call per partition (per-pair JNI calls would
 * drown in call overhead).
 */
#include <jni.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "swmi.h"
#include "swmi_shim.h"

static void throw_msg(JNIEnv *env, const char *msg) {
    jclass cls = (*env)->FindClass(env, "java/lang/RuntimeException");
    if (cls) (*env)->ThrowNew(env, cls, msg);
}

static void throw_rt(JNIEnv *env, const char *where) {
    char buf[640];
    snprintf(buf, sizeof buf, "%s: %s", where, swmi_last_error());
    throw_msg(env, buf);
}

JNIEXPORT jlong JNICALL Java_sw_GpuSmithWaterman_nativeCreate(JNIEnv *env, jclass cls, jint device) {
    swmi_ctx *ctx = NULL;
    (void)cls;
    if (swmi_create(device, &ctx) != SWMI_OK) { throw_rt(env, "swmi_create"); return 0; }
    return (jlong)(intptr_t)ctx;
}

JNIEXPORT void JNICALL Java_sw_GpuSmithWaterman_nativeDestroy(JNIEnv *env, jclass cls, jlong ctx) {
    (void)env; (void)cls;
    swmi_destroy((swmi_ctx *)(intptr_t)ctx);
}

/* refBytes/readBytes: direct ByteBuffers of ISO-8859-1 bytes; refOff/readOff: long[n+1] */
JNIEXPORT jlong JNICALL Java_sw_GpuSmithWaterman_nativeAlignBatch(
        JNIEnv *env, jclass cls, jlong ctx, jint match, jint mismatch, jint gap, jint tieMode, jbyteArray types,
        jobject refBytes, jlongArray refOff, jint nRefs, jobject readBytes, jlongArray readOff, jint nReads) {
    char err[640];
    jbyte ty[4] = {0, 0, 0, 0};
    jsize ty_len;
    jlong *ro, *qo;
    swmi_batch *b = NULL;
    int rc;
    (void)cls;
    if (!types || !refOff || !readOff) { throw_msg(env, "nativeAlignBatch: null array argument"); return 0; }
    ty_len = (*env)->GetArrayLength(env, types);
    if (ty_len == 4) (*env)->GetByteArrayRegion(env, types, 0, 4, ty);
    if ((*env)->GetArrayLength(env, refOff) != (jsize)nRefs + 1 || (*env)->GetArrayLength(env, readOff) != (jsize)nReads + 1) {
        throw_msg(env, "nativeAlignBatch: an offset array does not have n + 1 entries");
        return 0;
    }
    ro = (*env)->GetLongArrayElements(env, refOff, NULL);
    qo = (*env)->GetLongArrayElements(env, readOff, NULL);
    if (!ro || !qo) {
        if (ro) (*env)->ReleaseLongArrayElements(env, refOff, ro, JNI_ABORT);
        if (qo) (*env)->ReleaseLongArrayElements(env, readOff, qo, JNI_ABORT);
        throw_msg(env, "nativeAlignBatch: out of memory pinning the offset arrays");
        return 0;
    }
    rc = swmi_shim_align_batch((swmi_ctx *)(intptr_t)ctx, match, mismatch, gap, tieMode, (const signed char *)ty, (size_t)ty_len,
                               refBytes ? (*env)->GetDirectBufferAddress(env, refBytes) : NULL,
                               refBytes ? (int64_t)(*env)->GetDirectBufferCapacity(env, refBytes) : 0, (const int64_t *)ro, nRefs,
                               readBytes ? (*env)->GetDirectBufferAddress(env, readBytes) : NULL,
                               readBytes ? (int64_t)(*env)->GetDirectBufferCapacity(env, readBytes) : 0, (const int64_t *)qo, nReads,
                               &b, err, sizeof err);
    (*env)->ReleaseLongArrayElements(env, refOff, ro, JNI_ABORT);
    (*env)->ReleaseLongArrayElements(env, readOff, qo, JNI_ABORT);
    if (rc != SWMI_OK) { throw_msg(env, err); return 0; }
    return (jlong)(intptr_t)b;
}

JNIEXPORT void JNICALL Java_sw_GpuSmithWaterman_nativeFreeBatch(JNIEnv *env, jclass cls, jlong ctx, jlong batch) {
    (void)env; (void)cls;
    swmi_batch_free((swmi_ctx *)(intptr_t)ctx, (swmi_batch *)(intptr_t)batch);
}

JNIEXPORT jint JNICALL Java_sw_GpuSmithWaterman_nativeRefTotal(JNIEnv *env, jclass cls, jlong batch, jint ref) {
    char err[640];
    int32_t t = 0;
    (void)cls;
    if (swmi_shim_ref_total((swmi_batch *)(intptr_t)batch, ref, &t, err, sizeof err) != SWMI_OK) throw_msg(env, err);
    return t;
}

JNIEXPORT jlong JNICALL Java_sw_GpuSmithWaterman_nativeRefSiteCount(JNIEnv *env, jclass cls, jlong batch, jint ref) {
    char err[640];
    int64_t n = 0;
    (void)cls;
    if (swmi_shim_ref_site_count((swmi_batch *)(intptr_t)batch, ref, &n, err, sizeof err) != SWMI_OK) throw_msg(env, err);
    return (jlong)n;
}

/* fills begin[0] and returns {refAligned, readAligned} as ISO-8859-1 byte arrays */
JNIEXPORT jobjectArray JNICALL Java_sw_GpuSmithWaterman_nativeRefSite(JNIEnv *env, jclass cls, jlong batch, jint ref, jlong k, jintArray begin) {
    char err[640];
    int32_t b = 0; const char *r = NULL, *q = NULL; uint32_t len = 0;
    jint jb;
    jobjectArray out;
    jbyteArray ra, qa;
    jclass bytes_cls;
    (void)cls;
    if (swmi_shim_ref_site((swmi_batch *)(intptr_t)batch, ref, k, &b, &r, &q, &len, err, sizeof err) != SWMI_OK) {
        throw_msg(env, err);
        return NULL;
    }
    jb = b;
    if (begin && (*env)->GetArrayLength(env, begin) >= 1) (*env)->SetIntArrayRegion(env, begin, 0, 1, &jb);
    bytes_cls = (*env)->FindClass(env, "[B");
    if (!bytes_cls) return NULL;
    out = (*env)->NewObjectArray(env, 2, bytes_cls, NULL);
    ra = (*env)->NewByteArray(env, (jsize)len);
    qa = (*env)->NewByteArray(env, (jsize)len);
    if (!out || !ra || !qa) return NULL;                     /* OutOfMemoryError is already pending */
    (*env)->SetByteArrayRegion(env, ra, 0, (jsize)len, (const jbyte *)r);
    (*env)->SetByteArrayRegion(env, qa, 0, (jsize)len, (const jbyte *)q);
    (*env)->SetObjectArrayElement(env, out, 0, ra);
    (*env)->SetObjectArrayElement(env, out, 1, qa);
    return out;
}

/* {number of match sites, bytes of all their strings} of the references refLo .. refHi-1 */
JNIEXPORT jlongArray JNICALL Java_sw_GpuSmithWaterman_nativeRefSitesSizes(JNIEnv *env, jclass cls, jlong batch, jint refLo, jint refHi) {
    char err[640];
    int64_t sz[2] = {0, 0};
    jlong jsz[2];
    jlongArray out;
    (void)cls;
    if (swmi_shim_ref_sites_sizes((swmi_batch *)(intptr_t)batch, refLo, refHi, sz, err, sizeof err) != SWMI_OK) { throw_msg(env, err); return NULL; }
    out = (*env)->NewLongArray(env, 2);
    if (!out) return NULL;
    jsz[0] = (jlong)sz[0]; jsz[1] = (jlong)sz[1];
    (*env)->SetLongArrayRegion(env, out, 0, 2, jsz);
    return out;
}

/* MapRef's output of a whole range of references into the caller's arrays: ONE call per partition (swmi_shim.h) */
JNIEXPORT void JNICALL Java_sw_GpuSmithWaterman_nativeRefSitesPacked(
        JNIEnv *env, jclass cls, jlong batch, jint refLo, jint refHi, jintArray totals, jlongArray degenerate, jlongArray siteFirst,
        jintArray begins, jintArray lens, jlongArray strOff, jbyteArray blob) {
    char err[640];
    jint *t, *bg, *ln;
    jlong *dg, *sf, *so;
    jbyte *bl;
    int rc = SWMI_ERR_NOMEM;
    (void)cls;
    if (!totals || !degenerate || !siteFirst || !begins || !lens || !strOff || !blob) { throw_msg(env, "nativeRefSitesPacked: null array argument"); return; }
    t = (*env)->GetIntArrayElements(env, totals, NULL);
    dg = (*env)->GetLongArrayElements(env, degenerate, NULL);
    sf = (*env)->GetLongArrayElements(env, siteFirst, NULL);
    bg = (*env)->GetIntArrayElements(env, begins, NULL);
    ln = (*env)->GetIntArrayElements(env, lens, NULL);
    so = (*env)->GetLongArrayElements(env, strOff, NULL);
    bl = (*env)->GetByteArrayElements(env, blob, NULL);
    if (t && dg && sf && bg && ln && so && bl) {
        jsize ns = (*env)->GetArrayLength(env, begins);
        if ((*env)->GetArrayLength(env, lens) < ns) ns = (*env)->GetArrayLength(env, lens);
        if ((*env)->GetArrayLength(env, strOff) < ns) ns = (*env)->GetArrayLength(env, strOff);
        rc = swmi_shim_ref_sites_packed((swmi_batch *)(intptr_t)batch, refLo, refHi,
                                        (int32_t *)t, (int64_t)(*env)->GetArrayLength(env, totals),
                                        (int64_t *)dg, (int64_t)(*env)->GetArrayLength(env, degenerate),
                                        (int64_t *)sf, (int64_t)(*env)->GetArrayLength(env, siteFirst),
                                        (int32_t *)bg, (int32_t *)ln, (int64_t *)so, (int64_t)ns,
                                        (signed char *)bl, (int64_t)(*env)->GetArrayLength(env, blob), err, sizeof err);
    } else {
        snprintf(err, sizeof err, "nativeRefSitesPacked: out of memory pinning the output arrays");
    }
    /* mode 0: copy back (if the VM handed out copies) and release */
    if (t) (*env)->ReleaseIntArrayElements(env, totals, t, 0);
    if (dg) (*env)->ReleaseLongArrayElements(env, degenerate, dg, 0);
    if (sf) (*env)->ReleaseLongArrayElements(env, siteFirst, sf, 0);
    if (bg) (*env)->ReleaseIntArrayElements(env, begins, bg, 0);
    if (ln) (*env)->ReleaseIntArrayElements(env, lens, ln, 0);
    if (so) (*env)->ReleaseLongArrayElements(env, strOff, so, 0);
    if (bl) (*env)->ReleaseByteArrayElements(env, blob, bl, 0);
    if (rc != SWMI_OK) throw_msg(env, err);
}
