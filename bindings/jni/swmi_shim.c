/* swmi_shim.c -- see swmi_shim.h.  C99, no JNI types. */
#include <stdio.h>
#include <string.h>
#include "swmi_shim.h"

static int shim_fail(int code, char *err, size_t err_len, const char *where, const char *msg) {
    if (err && err_len) snprintf(err, err_len, "%s: %s", where, msg);
    return code;
}

static int check_side(const char *what, const void *bytes, int64_t cap, const int64_t *off, int32_t n,
                      char *err, size_t err_len) {
    int32_t k;
    if (n < 0) return shim_fail(SWMI_ERR_INVALID, err, err_len, what, "negative sequence count");
    if (!off) return shim_fail(SWMI_ERR_INVALID, err, err_len, what, "offset array is null");
    if (off[0] != 0) return shim_fail(SWMI_ERR_INVALID, err, err_len, what, "offsets must start at 0");
    for (k = 0; k < n; k++)
        if (off[k + 1] < off[k]) return shim_fail(SWMI_ERR_INVALID, err, err_len, what, "offsets decrease");
    if (off[n] > 0 && !bytes)
        return shim_fail(SWMI_ERR_INVALID, err, err_len, what, "byte buffer is null or not a direct ByteBuffer");
    if (off[n] > cap) return shim_fail(SWMI_ERR_INVALID, err, err_len, what, "offsets run past the buffer's capacity");
    return SWMI_OK;
}

int swmi_shim_align_batch(swmi_ctx *ctx, int32_t match, int32_t mismatch, int32_t gap, int32_t tie_mode,
                          const signed char *types, size_t types_len,
                          const void *ref_bytes, int64_t ref_cap, const int64_t *ref_off, int32_t n_refs,
                          const void *read_bytes, int64_t read_cap, const int64_t *read_off, int32_t n_reads,
                          swmi_batch **out, char *err, size_t err_len) {
    swmi_params p;
    int rc;
    if (!out) return shim_fail(SWMI_ERR_INVALID, err, err_len, "nativeAlignBatch", "out is null");
    *out = NULL;
    if (!ctx) return shim_fail(SWMI_ERR_INVALID, err, err_len, "nativeAlignBatch", "context handle is 0");
    if (!types || types_len != 4)
        return shim_fail(SWMI_ERR_INVALID, err, err_len, "nativeAlignBatch", "alignTypes must hold exactly 4 characters");
    if ((rc = check_side("references", ref_bytes, ref_cap, ref_off, n_refs, err, err_len)) != SWMI_OK) return rc;
    if ((rc = check_side("reads", read_bytes, read_cap, read_off, n_reads, err, err_len)) != SWMI_OK) return rc;
    swmi_default_params(&p);
    p.match = match; p.mismatch = mismatch; p.gap = gap; p.tie_mode = tie_mode;
    memcpy(p.types, types, 4);
    /* jlong and uint64_t have the same size and the offsets were checked non-negative */
    rc = swmi_align_batch(ctx, &p, (const uint8_t *)ref_bytes, (const uint64_t *)(const void *)ref_off, (uint32_t)n_refs,
                          (const uint8_t *)read_bytes, (const uint64_t *)(const void *)read_off, (uint32_t)n_reads, out);
    if (rc != SWMI_OK) return shim_fail(rc, err, err_len, "swmi_align_batch", swmi_last_error());
    return SWMI_OK;
}

int swmi_shim_ref_total(const swmi_batch *b, int32_t ref, int32_t *total, char *err, size_t err_len) {
    int rc;
    if (ref < 0) return shim_fail(SWMI_ERR_RANGE, err, err_len, "nativeRefTotal", "negative reference index");
    rc = swmi_ref_total(b, (uint32_t)ref, total);
    return rc == SWMI_OK ? rc : shim_fail(rc, err, err_len, "swmi_ref_total", swmi_last_error());
}

int swmi_shim_ref_site_count(swmi_batch *b, int32_t ref, int64_t *n, char *err, size_t err_len) {
    uint64_t v = 0;
    int rc;
    if (ref < 0) return shim_fail(SWMI_ERR_RANGE, err, err_len, "nativeRefSiteCount", "negative reference index");
    rc = swmi_ref_n_match_sites(b, (uint32_t)ref, &v);
    if (rc != SWMI_OK) return shim_fail(rc, err, err_len, "swmi_ref_n_match_sites", swmi_last_error());
    if (n) *n = (int64_t)v;
    return SWMI_OK;
}

int swmi_shim_ref_site(swmi_batch *b, int32_t ref, int64_t k, int32_t *begin, const char **ref_aln, const char **read_aln,
                       uint32_t *len, char *err, size_t err_len) {
    int rc;
    if (ref < 0 || k < 0) return shim_fail(SWMI_ERR_RANGE, err, err_len, "nativeRefSite", "negative index");
    rc = swmi_ref_match_site(b, (uint32_t)ref, (uint64_t)k, begin, ref_aln, read_aln, len);
    return rc == SWMI_OK ? rc : shim_fail(rc, err, err_len, "swmi_ref_match_site", swmi_last_error());
}

int swmi_shim_ref_sites_sizes(swmi_batch *b, int32_t ref_lo, int32_t ref_hi, int64_t sizes[2], char *err, size_t err_len) {
    uint64_t ns = 0, nb = 0;
    int rc;
    if (ref_lo < 0 || ref_hi < ref_lo) return shim_fail(SWMI_ERR_RANGE, err, err_len, "nativeRefSitesSizes", "bad reference range");
    if (!sizes) return shim_fail(SWMI_ERR_INVALID, err, err_len, "nativeRefSitesSizes", "sizes is null");
    rc = swmi_ref_sites_packed(b, (uint32_t)ref_lo, (uint32_t)ref_hi, NULL, NULL, NULL, NULL, NULL, NULL, 0, NULL, 0, &ns, &nb);
    if (rc != SWMI_OK) return shim_fail(rc, err, err_len, "swmi_ref_sites_packed", swmi_last_error());
    sizes[0] = (int64_t)ns; sizes[1] = (int64_t)nb;
    return SWMI_OK;
}

int swmi_shim_ref_sites_packed(swmi_batch *b, int32_t ref_lo, int32_t ref_hi,
                               int32_t *totals, int64_t n_totals, int64_t *degenerate, int64_t n_degenerate,
                               int64_t *site_first, int64_t n_site_first,
                               int32_t *begins, int32_t *lens, int64_t *str_off, int64_t n_sites_cap,
                               signed char *blob, int64_t blob_cap, char *err, size_t err_len) {
    uint64_t ns = 0, nb = 0;
    int64_t n;
    int rc;
    if (ref_lo < 0 || ref_hi < ref_lo) return shim_fail(SWMI_ERR_RANGE, err, err_len, "nativeRefSitesPacked", "bad reference range");
    n = (int64_t)ref_hi - ref_lo;
    if (!totals || !degenerate || !site_first || !begins || !lens || !str_off || (!blob && blob_cap > 0))
        return shim_fail(SWMI_ERR_INVALID, err, err_len, "nativeRefSitesPacked", "null array argument");
    if (n_totals < n || n_degenerate < n || n_site_first < n + 1 || n_sites_cap < 0 || blob_cap < 0)
        return shim_fail(SWMI_ERR_INVALID, err, err_len, "nativeRefSitesPacked", "an output array is shorter than the reference range needs");
    /* jint / jlong and the C-ABI's unsigned types have the same sizes; counts were checked non-negative */
    rc = swmi_ref_sites_packed(b, (uint32_t)ref_lo, (uint32_t)ref_hi, totals, (uint64_t *)(void *)degenerate,
                               (uint64_t *)(void *)site_first, begins, (uint32_t *)(void *)lens, (uint64_t *)(void *)str_off,
                               (uint64_t)n_sites_cap, (uint8_t *)(void *)blob, (uint64_t)blob_cap, &ns, &nb);
    if (rc != SWMI_OK) return shim_fail(rc, err, err_len, "swmi_ref_sites_packed", swmi_last_error());
    return SWMI_OK;
}
