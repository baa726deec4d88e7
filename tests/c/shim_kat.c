/* shim_kat.c -- plain C99 over include/swmi.h: performs exactly the JNI shim's call sequence (bindings/jni/swmi_shim.c:
 * swmi_create -> swmi_align_batch -> swmi_ref_total / _n_match_sites / _match_site -> swmi_batch_free) on KAT-3
 * (ref AAAA, read AACA, scores 1/-1/-1) and prints what MapRef.call returns
 * (Distribution.java:403-436); tests/test_boundary.py compares the output with tests/golden/kat.json.
 * Built with gcc -std=c99 -Wall -Wextra -Werror -pedantic. */
#include <stdio.h>
#include <string.h>
#include "swmi.h"
#include "swmi_io.h"
#include "swmi_shim.h"

int main(int argc, char **argv) {
    char err[640];
    swmi_ctx *ctx = NULL;
    swmi_batch *b = NULL;
    const char *ref = "AAAA", *read = "AACA";
    const signed char types[4] = {'a', 'i', 'd', '-'};
    int64_t ro[2], qo[2], n = 0, k;
    int32_t total = 0;
    int tie = argc > 1 && strcmp(argv[1], "strict") == 0 ? SWMI_TIE_STRICT : SWMI_TIE_SERIAL;
    int rc;
    ro[0] = 0; ro[1] = 4; qo[0] = 0; qo[1] = 4;
    if (swmi_abi_version() != SWMI_ABI_VERSION) { printf("ERROR abi %d\n", swmi_abi_version()); return 2; }
    if (swmi_create(0, &ctx) != SWMI_OK) { printf("ERROR %s\n", swmi_last_error()); return 3; }
    /* the argument checks of the shim */
    rc = swmi_shim_align_batch(ctx, 1, -1, -1, tie, types, 3, ref, 4, ro, 1, read, 4, qo, 1, &b, err, sizeof err);
    if (rc != SWMI_ERR_INVALID || b) { printf("ERROR short alignTypes accepted\n"); return 4; }
    rc = swmi_shim_align_batch(ctx, 1, -1, -1, tie, types, 4, NULL, 0, ro, 1, read, 4, qo, 1, &b, err, sizeof err);
    if (rc != SWMI_ERR_INVALID || b) { printf("ERROR non-direct buffer accepted\n"); return 4; }
    rc = swmi_shim_align_batch(ctx, 1, -1, -1, tie, types, 4, ref, 3, ro, 1, read, 4, qo, 1, &b, err, sizeof err);
    if (rc != SWMI_ERR_INVALID || b) { printf("ERROR offsets past the capacity accepted\n"); return 4; }
    /* the call sequence */
    rc = swmi_shim_align_batch(ctx, 1, -1, -1, tie, types, 4, ref, 4, ro, 1, read, 4, qo, 1, &b, err, sizeof err);
    if (rc != SWMI_OK) { printf("ERROR %s\n", err); return 5; }
    if (swmi_shim_ref_total(b, 0, &total, err, sizeof err) != SWMI_OK) { printf("ERROR %s\n", err); return 6; }
    if (swmi_shim_ref_site_count(b, 0, &n, err, sizeof err) != SWMI_OK) { printf("ERROR %s\n", err); return 6; }
    printf("%d %ld", (int)total, (long)n);
    for (k = 0; k < n; k++) {
        int32_t begin = 0; const char *ra = NULL, *qa = NULL; uint32_t len = 0;
        if (swmi_shim_ref_site(b, 0, k, &begin, &ra, &qa, &len, err, sizeof err) != SWMI_OK) { printf("ERROR %s\n", err); return 7; }
        printf(" %d:%s/%s", (int)begin, ra, qa);
    }
    printf("\n");
    if (swmi_shim_ref_site(b, 0, n, NULL, NULL, NULL, NULL, err, sizeof err) != SWMI_ERR_RANGE) { printf("ERROR range\n"); return 8; }
    /* the same sites through the bulk accessor (one call per partition): a second output line in the same format */
    {
        int64_t sizes[2] = {0, 0}, deg[1], first[2], off[64];
        int32_t tot[1], begins[64], lens[64];
        signed char blob[1024];
        if (swmi_shim_ref_sites_sizes(b, 0, 1, sizes, err, sizeof err) != SWMI_OK) { printf("ERROR %s\n", err); return 9; }
        if (sizes[0] != n || sizes[0] > 64 || sizes[1] > (int64_t)sizeof blob) { printf("ERROR sizes %ld %ld\n", (long)sizes[0], (long)sizes[1]); return 9; }
        /* arrays too short for the range, then a blob too small for the strings */
        if (swmi_shim_ref_sites_packed(b, 0, 1, tot, 0, deg, 1, first, 2, begins, lens, off, 64, blob, (int64_t)sizeof blob, err, sizeof err) != SWMI_ERR_INVALID) { printf("ERROR short totals accepted\n"); return 9; }
        if (sizes[1] > 0 && swmi_shim_ref_sites_packed(b, 0, 1, tot, 1, deg, 1, first, 2, begins, lens, off, 64, blob, sizes[1] - 1, err, sizeof err) != SWMI_ERR_RANGE) { printf("ERROR short blob accepted\n"); return 9; }
        if (swmi_shim_ref_sites_packed(b, 0, 1, tot, 1, deg, 1, first, 2, begins, lens, off, 64, blob, (int64_t)sizeof blob, err, sizeof err) != SWMI_OK) { printf("ERROR %s\n", err); return 9; }
        printf("%d %ld", (int)tot[0], (long)(deg[0] + first[1] - first[0]));
        for (k = first[0]; k < first[1]; k++)
            printf(" %d:%.*s/%.*s", (int)begins[k], (int)lens[k], (const char *)blob + off[k], (int)lens[k], (const char *)blob + off[k] + lens[k]);
        printf("\n");
    }
    swmi_batch_free(ctx, b);
    swmi_destroy(ctx);
    return 0;
}
