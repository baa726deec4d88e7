/*
 * swmi_shim.h -- the JNI shim's logic without JNI: argument checks + the exact C-ABI call sequence the
 * Java_sw_GpuSmithWaterman_* functions of swmi_jni.c perform.  Plain C99 over include/swmi.h only, so it is compiled
 * and run by the tests (tests/c/shim_kat.c, gcc -std=c99 -Wall -Werror -pedantic) although the build image has no
 * JDK; swmi_jni.c only unwraps JNI types and forwards here.
 *
 * Replaces the per-pair call  new SmithWaterman.OptAlignments().call(seqs, alignScores, alignTypes)
 * at src/sw/Distribution.java:421-422 by ONE native call per partition.
 */
#ifndef SWMI_SHIM_H
#define SWMI_SHIM_H
#include <stddef.h>
#include <stdint.h>
#include "swmi.h"

#ifdef __cplusplus
extern "C" {
#endif

/* nativeAlignBatch: `types` = the alignTypes characters narrowed to bytes (types_len must be 4); ref_bytes / read_bytes
 * = addresses of the direct ByteBuffers (NULL when the buffer is not direct) and their capacities; ref_off / read_off =
 * the long[n+1] offset arrays.  Returns SWMI_OK and the batch, or a negative swmi_status with a message in err. */
int swmi_shim_align_batch(swmi_ctx *ctx, int32_t match, int32_t mismatch, int32_t gap, int32_t tie_mode,
                          const signed char *types, size_t types_len,
                          const void *ref_bytes, int64_t ref_cap, const int64_t *ref_off, int32_t n_refs,
                          const void *read_bytes, int64_t read_cap, const int64_t *read_off, int32_t n_reads,
                          swmi_batch **out, char *err, size_t err_len);

/* nativeRefTotal / nativeRefSiteCount / nativeRefSite */
int swmi_shim_ref_total(const swmi_batch *b, int32_t ref, int32_t *total, char *err, size_t err_len);
int swmi_shim_ref_site_count(swmi_batch *b, int32_t ref, int64_t *n, char *err, size_t err_len);
int swmi_shim_ref_site(swmi_batch *b, int32_t ref, int64_t k, int32_t *begin, const char **ref_aln, const char **read_aln,
                       uint32_t *len, char *err, size_t err_len);

/* nativeRefSitesSizes / nativeRefSitesPacked: MapRef's output of the references ref_lo .. ref_hi-1 in ONE call
 * (swmi_ref_sites_packed, include/swmi.h).  The Java side asks for the sizes, allocates its int[] / long[] / byte[] once per
 * partition, and builds every String from a slice of one byte[] -- instead of three JNI calls and two array allocations per
 * match site.  `sizes` receives {number of sites, bytes of all strings}.  Lengths are checked against the arrays' lengths. */
int swmi_shim_ref_sites_sizes(swmi_batch *b, int32_t ref_lo, int32_t ref_hi, int64_t sizes[2], char *err, size_t err_len);
int swmi_shim_ref_sites_packed(swmi_batch *b, int32_t ref_lo, int32_t ref_hi,
                               int32_t *totals, int64_t n_totals, int64_t *degenerate, int64_t n_degenerate,
                               int64_t *site_first, int64_t n_site_first,
                               int32_t *begins, int32_t *lens, int64_t *str_off, int64_t n_sites_cap,
                               signed char *blob, int64_t blob_cap, char *err, size_t err_len);

#ifdef __cplusplus
}
#endif
#endif
