/*
 * swmi_jni.c -- thin JNI shim over include/swmi.h for the Java class sw.GpuSmithWaterman
 * (bindings/java/sw/GpuSmithWaterman.java).  It only unwraps arguments and forwards; all work is in
 * libswmi.so.  NOT compiled in this repository's build: the build image has no JDK (no jni.h).  On a
 * machine with a JDK:
 *     gcc -shared -fPIC -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -Iinclude \
 *         bindings/jni/swmi_jni.c -Lsparksmithwaterman_amd/lib -lswmi -o libswmi_jni.so
 *
 * Replaces the per-pair call  new SmithWaterman.OptAlignments().call(seqs, alignScores, alignTypes)
 * at src/sw/Distribution.java:421-422 by ONE native call per partition (per-pair JNI calls would
 * drown in call overhead).
 */
#include <jni.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "swmi.h"

static void throw_rt(JNIEnv *env, const char *where) {
    char buf[640];
    snprintf(buf, sizeof buf, "%s: %s", where, swmi_last_error());
    (*env)->ThrowNew(env, (*env)->FindClass(env, "java/lang/RuntimeException"), buf);
}

JNIEXPORT jlong JNICALL Java_sw_GpuSmithWaterman_nativeCreate(JNIEnv *env, jclass cls, jint device) {
    swmi_ctx *ctx = NULL;
    if (swmi_create(device, &ctx) != SWMI_OK) { throw_rt(env, "swmi_create"); return 0; }
    return (jlong)(intptr_t)ctx;
}

JNIEXPORT void JNICALL Java_sw_GpuSmithWaterman_nativeDestroy(JNIEnv *env, jclass cls, jlong ctx) {
    swmi_destroy((swmi_ctx *)(intptr_t)ctx);
}

/* refBytes/readBytes: direct ByteBuffers of ISO-8859-1 bytes; refOff/readOff: long[n+1] */
JNIEXPORT jlong JNICALL Java_sw_GpuSmithWaterman_nativeAlignBatch(
        JNIEnv *env, jclass cls, jlong ctx, jint match, jint mismatch, jint gap, jint tieMode, jbyteArray types,
        jobject refBytes, jlongArray refOff, jint nRefs, jobject readBytes, jlongArray readOff, jint nReads) {
    swmi_params p;
    swmi_default_params(&p);
    p.match = match; p.mismatch = mismatch; p.gap = gap; p.tie_mode = tieMode;
    (*env)->GetByteArrayRegion(env, types, 0, 4, (jbyte *)p.types);
    jlong *ro = (*env)->GetLongArrayElements(env, refOff, NULL);
    jlong *qo = (*env)->GetLongArrayElements(env, readOff, NULL);
    swmi_batch *b = NULL;
    int rc = swmi_align_batch((swmi_ctx *)(intptr_t)ctx, &p,
                              (const uint8_t *)(*env)->GetDirectBufferAddress(env, refBytes), (const uint64_t *)ro, (uint32_t)nRefs,
                              (const uint8_t *)(*env)->GetDirectBufferAddress(env, readBytes), (const uint64_t *)qo, (uint32_t)nReads, &b);
    (*env)->ReleaseLongArrayElements(env, refOff, ro, JNI_ABORT);
    (*env)->ReleaseLongArrayElements(env, readOff, qo, JNI_ABORT);
    if (rc != SWMI_OK) { throw_rt(env, "swmi_align_batch"); return 0; }
    return (jlong)(intptr_t)b;
}

JNIEXPORT void JNICALL Java_sw_GpuSmithWaterman_nativeFreeBatch(JNIEnv *env, jclass cls, jlong ctx, jlong batch) {
    swmi_batch_free((swmi_ctx *)(intptr_t)ctx, (swmi_batch *)(intptr_t)batch);
}

JNIEXPORT jint JNICALL Java_sw_GpuSmithWaterman_nativeRefTotal(JNIEnv *env, jclass cls, jlong batch, jint ref) {
    int32_t t = 0;
    if (swmi_ref_total((swmi_batch *)(intptr_t)batch, (uint32_t)ref, &t) != SWMI_OK) throw_rt(env, "swmi_ref_total");
    return t;
}

JNIEXPORT jlong JNICALL Java_sw_GpuSmithWaterman_nativeRefSiteCount(JNIEnv *env, jclass cls, jlong batch, jint ref) {
    uint64_t n = 0;
    if (swmi_ref_n_match_sites((swmi_batch *)(intptr_t)batch, (uint32_t)ref, &n) != SWMI_OK) throw_rt(env, "swmi_ref_n_match_sites");
    return (jlong)n;
}

/* fills begin[0] and returns {refAligned, readAligned} as ISO-8859-1 byte arrays */
JNIEXPORT jobjectArray JNICALL Java_sw_GpuSmithWaterman_nativeRefSite(JNIEnv *env, jclass cls, jlong batch, jint ref, jlong k, jintArray begin) {
    int32_t b = 0; const char *r = NULL, *q = NULL; uint32_t len = 0;
    if (swmi_ref_match_site((swmi_batch *)(intptr_t)batch, (uint32_t)ref, (uint64_t)k, &b, &r, &q, &len) != SWMI_OK) {
        throw_rt(env, "swmi_ref_match_site");
        return NULL;
    }
    jint jb = b;
    (*env)->SetIntArrayRegion(env, begin, 0, 1, &jb);
    jobjectArray out = (*env)->NewObjectArray(env, 2, (*env)->FindClass(env, "[B"), NULL);
    jbyteArray ra = (*env)->NewByteArray(env, (jsize)len), qa = (*env)->NewByteArray(env, (jsize)len);
    (*env)->SetByteArrayRegion(env, ra, 0, (jsize)len, (const jbyte *)r);
    (*env)->SetByteArrayRegion(env, qa, 0, (jsize)len, (const jbyte *)q);
    (*env)->SetObjectArrayElement(env, out, 0, ra);
    (*env)->SetObjectArrayElement(env, out, 1, qa);
    return out;
}
