/*
 * sw_oracle.c -- CPU restatement of the reference's Smith-Waterman hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library, and only as the checker / the reported CPU baseline.  The product
 * path (sparksmithwaterman_amd + libswmi.so) never links or calls it.
 *
 * PARITY STATUS: "parity unpinned".  The reference (elizabethfong/SparkSmithWaterman,
 * Java 8 + Spark 1.5.2) ships no tests, golden files or fixtures, and no JVM
 * exists in the build container, so the reference itself cannot be run.  This
 * file is a line-by-line restatement of the cited Java, checked against the four
 * hand-derived known-answer vectors of SURVEY.md section 8(c) (tests/golden/kat.json)
 * and cross-checked against an independent pure-Python transliteration
 * (oracle/sw_oracle_py.py).
 *
 * What is restated (paths relative to /root/reference):
 *   src/sw/SmithWaterman.java:62-92    OptAlignments.call   -> sw_oracle_align()
 *   src/sw/SmithWaterman.java:129-190  ScoreMatrix.call     -> fill()
 *   src/sw/SmithWaterman.java:217-252  GetCellScore.call    -> cell_score()  (tie_mode 0)
 *   src/sw/SmithWaterman.java:277-280  InsDelScore.call     -> inlined add
 *   src/sw/SmithWaterman.java:309-318  AlignmentScore.call  -> align_score()
 *   src/sw/SmithWaterman.java:354-436  GetAlignment.call    -> traceback()
 *   src/sw/DistributedSW.java:305-330  GetCellScore (strict '>') -> cell_score() (tie_mode 1)
 *   src/sw/DistributedSW.java:192-245  per-anti-diagonal max-cell order -> fill_diag_order()
 *   src/sw/DistributedSW.java:479-487  stable sort of alignments by beginning (tie_mode 1)
 *   src/sw/Distribution.java:403-436   MapRef.call          -> sw_oracle_map_ref()
 *   src/sw/Distribution.java:691-694   MatchSiteComp        -> stable sort by begin
 *
 * Domain: sequences are byte strings holding ISO-8859-1 characters.  Java's
 * Character.toUpperCase is restated exactly for that range (see up()).  All score arithmetic
 * is Java int (32-bit two's complement, wrapping), done here in uint32_t.
 */
#define _POSIX_C_SOURCE 200809L
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>
#include <time.h>

#define SW_TIE_SERIAL 0   /* SmithWaterman.java: '>=' chain, priority a > i > d   */
#define SW_TIE_STRICT 1   /* DistributedSW.java: '>'  chain, priority d > i > a   */

typedef struct {
    int32_t begin;        /* 1-based ref index of first aligned column, 0 if empty */
    int32_t end_i, end_j; /* the max cell the traceback started from               */
    char   *ref_aln;      /* NUL-terminated, '_' = gap                             */
    char   *read_aln;
} sw_oracle_aln;

typedef struct {
    int32_t        score;
    int64_t        n_aln;
    sw_oracle_aln *aln;
    /* optional matrix dump, (m+1) x (n+1), row-major */
    int32_t       *H;
    char          *T;
    int64_t        m, n;
} sw_oracle_result;

static char EMPTY[1] = { 0 };

/* SmithWaterman.java:311-312 -- Character.toUpperCase restricted to ISO-8859-1 input:
 * a-z and 0xE0-0xFE (except the division sign 0xF7) drop 0x20; 0xB5 (micro) and 0xFF (y-diaeresis)
 * map outside Latin-1 to characters nothing else maps to, so they only equal themselves. */
static inline unsigned char up(unsigned char c) {
    if (c >= 'a' && c <= 'z') return (unsigned char)(c - 32);
    if (c >= 0xE0 && c <= 0xFE && c != 0xF7) return (unsigned char)(c - 32);
    return c;
}

/* SmithWaterman.java:309-318 */
static inline int32_t align_score(int32_t nw, unsigned char refb, unsigned char readb,
                                  int32_t match, int32_t mismatch) {
    if (up(refb) == up(readb)) return (int32_t)((uint32_t)nw + (uint32_t)match);
    return (int32_t)((uint32_t)nw + (uint32_t)mismatch);
}

/* SmithWaterman.java:217-252 (tie_mode 0) / DistributedSW.java:305-330 (tie_mode 1).
 * cells = {NW, N, W}; returns score, writes the alignment-type char. */
static inline int32_t cell_score(int32_t nw, int32_t nn, int32_t ww,
                                 unsigned char refb, unsigned char readb,
                                 const int32_t sc[3], const char ty[4], int tie_mode,
                                 char *type_out) {
    int32_t max = 0;
    char t = ty[3];
    int32_t tmp;
    if (tie_mode == SW_TIE_SERIAL) {
        tmp = (int32_t)((uint32_t)ww + (uint32_t)sc[2]);       /* deletion  :227 */
        if (tmp >= max) { max = tmp; t = ty[2]; }
        tmp = (int32_t)((uint32_t)nn + (uint32_t)sc[2]);       /* insertion :235 */
        if (tmp >= max) { max = tmp; t = ty[1]; }
        tmp = align_score(nw, refb, readb, sc[0], sc[1]);      /* alignment :244 */
        if (tmp >= max) { max = tmp; t = ty[0]; }
    } else {
        tmp = (int32_t)((uint32_t)ww + (uint32_t)sc[2]);       /* DistributedSW:309 */
        if (tmp > max) { max = tmp; t = ty[2]; }
        tmp = (int32_t)((uint32_t)nn + (uint32_t)sc[2]);       /* :317 */
        if (tmp > max) { max = tmp; t = ty[1]; }
        tmp = align_score(nw, refb, readb, sc[0], sc[1]);      /* :325 */
        if (tmp > max) { max = tmp; t = ty[0]; }
    }
    *type_out = t;
    return max;
}

typedef struct { int32_t i, j; } cell_t;
typedef struct { cell_t *v; int64_t n, cap; } cell_list;

static void cl_clear(cell_list *l) { l->n = 0; }
static int cl_add(cell_list *l, int32_t i, int32_t j) {
    if (l->n == l->cap) {
        int64_t nc = l->cap ? l->cap * 2 : 16;
        cell_t *nv = (cell_t *)realloc(l->v, (size_t)nc * sizeof(cell_t));
        if (!nv) return -1;
        l->v = nv; l->cap = nc;
    }
    l->v[l->n].i = i; l->v[l->n].j = j; l->n++;
    return 0;
}

/* max bookkeeping, SmithWaterman.java:176-185 == DistributedSW.java:228-238 */
static inline int note_max(cell_list *l, int32_t *maxScore, int32_t score, int32_t i, int32_t j) {
    if (score > *maxScore) { cl_clear(l); *maxScore = score; return cl_add(l, i, j); }
    else if (score == *maxScore) return cl_add(l, i, j);
    return 0;
}

/* SmithWaterman.java:129-190: init all cells (142-149), row-major fill (157-187). */
static int fill(int32_t *H, char *T, int64_t m, int64_t n,
                const unsigned char *ref, const unsigned char *read,
                const int32_t sc[3], const char ty[4], int tie_mode,
                int32_t *maxScore, cell_list *maxCells) {
    int64_t W = n + 1;
    for (int64_t i = 0; i <= m; i++)
        for (int64_t j = 0; j <= n; j++) { H[i * W + j] = 0; T[i * W + j] = ty[3]; }
    *maxScore = 0;
    for (int64_t i = 1; i <= m; i++) {
        for (int64_t j = 1; j <= n; j++) {
            char t;
            int32_t s = cell_score(H[(i - 1) * W + j - 1], H[(i - 1) * W + j], H[i * W + j - 1],
                                   ref[j - 1], read[i - 1], sc, ty, tie_mode, &t);
            H[i * W + j] = s;
            T[i * W + j] = t;
            if (note_max(maxCells, maxScore, s, (int32_t)i, (int32_t)j)) return -1;
        }
    }
    return 0;
}

/* DistributedSW.java:192-245: one anti-diagonal at a time; within a diagonal the
 * collected cells are sorted by CellResultComp (DistributedSW.java:893-916,
 * ascending j) before max bookkeeping.  Diagonals start at (1,1), then walk
 * down the first column and along the bottom row (GetNextStart, :836-880);
 * each diagonal runs up-right from its start cell (GetInitList, :655-700).
 * The cell values do not depend on the visiting order; only the max-cell list
 * order does. */
static int fill_diag_order(int32_t *H, char *T, int64_t m, int64_t n,
                           const unsigned char *ref, const unsigned char *read,
                           const int32_t sc[3], const char ty[4],
                           int32_t *maxScore, cell_list *maxCells) {
    int64_t W = n + 1;
    for (int64_t i = 0; i <= m; i++)
        for (int64_t j = 0; j <= n; j++) { H[i * W + j] = 0; T[i * W + j] = ty[3]; }
    *maxScore = 0;
    /* diagonal d holds cells with i + j == d, d = 2 .. m + n; ascending j */
    for (int64_t d = 2; d <= m + n; d++) {
        int64_t jlo = d - m; if (jlo < 1) jlo = 1;
        int64_t jhi = d - 1; if (jhi > n) jhi = n;
        for (int64_t j = jlo; j <= jhi; j++) {
            int64_t i = d - j;
            char t;
            int32_t s = cell_score(H[(i - 1) * W + j - 1], H[(i - 1) * W + j], H[i * W + j - 1],
                                   ref[j - 1], read[i - 1], sc, ty, SW_TIE_STRICT, &t);
            H[i * W + j] = s;
            T[i * W + j] = t;
            if (note_max(maxCells, maxScore, s, (int32_t)i, (int32_t)j)) return -1;
        }
    }
    return 0;
}

/* SmithWaterman.java:354-436 == DistributedSW.java:523-595. */
static int traceback(const int32_t *H, const char *T, int64_t n,
                     const unsigned char *ref, const unsigned char *read,
                     const char ty[4], int32_t ci, int32_t cj, sw_oracle_aln *out) {
    int64_t W = n + 1;
    int64_t i = ci, j = cj;
    int32_t score = H[i * W + j];
    int32_t beginning = 0;
    out->end_i = ci; out->end_j = cj;
    if (score <= 0) {                 /* loop never runs: (0, "", "") */
        out->begin = 0; out->ref_aln = EMPTY; out->read_aln = EMPTY;
        return 0;
    }
    int64_t cap = i + j + 1, len = 0;
    char *sr = (char *)malloc((size_t)cap + 1), *sq = (char *)malloc((size_t)cap + 1);
    if (!sr || !sq) { free(sr); free(sq); return -1; }
    while (score > 0) {
        beginning = (int32_t)j;                                   /* :383 */
        char a = T[i * W + j];
        if (a == ty[0])      { sr[len] = (char)ref[j - 1]; sq[len] = (char)read[i - 1]; i--; j--; }
        else if (a == ty[1]) { sr[len] = '_';              sq[len] = (char)read[i - 1]; i--; }
        else                 { sr[len] = (char)ref[j - 1]; sq[len] = '_';               j--; }
        len++;
        score = H[i * W + j];
    }
    /* pop the stack: reverse */
    for (int64_t a = 0, b = len - 1; a < b; a++, b--) {
        char t = sr[a]; sr[a] = sr[b]; sr[b] = t;
        t = sq[a]; sq[a] = sq[b]; sq[b] = t;
    }
    sr[len] = 0; sq[len] = 0;
    out->begin = beginning; out->ref_aln = sr; out->read_aln = sq;
    return 0;
}

static int cmp_begin(const void *a, const void *b) { /* unused: qsort is not stable */
    (void)a; (void)b; return 0;
}

/* stable merge sort by begin (Collections.sort is a stable merge sort;
 * comparator = MatchSiteComp, Distribution.java:691-694) */
static void stable_sort_by_begin(sw_oracle_aln *a, int64_t n) {
    (void)cmp_begin;
    if (n < 2) return;
    sw_oracle_aln *tmp = (sw_oracle_aln *)malloc((size_t)n * sizeof(*tmp));
    for (int64_t w = 1; w < n; w *= 2) {
        for (int64_t lo = 0; lo < n; lo += 2 * w) {
            int64_t mid = lo + w < n ? lo + w : n, hi = lo + 2 * w < n ? lo + 2 * w : n;
            int64_t x = lo, y = mid, k = lo;
            while (x < mid && y < hi) tmp[k++] = (a[y].begin < a[x].begin) ? a[y++] : a[x++];
            while (x < mid) tmp[k++] = a[x++];
            while (y < hi) tmp[k++] = a[y++];
        }
        memcpy(a, tmp, (size_t)n * sizeof(*tmp));
    }
    free(tmp);
}

void sw_oracle_free(sw_oracle_result *r) {
    if (!r) return;
    for (int64_t k = 0; k < r->n_aln; k++) {
        if (r->aln[k].ref_aln != EMPTY) free(r->aln[k].ref_aln);
        if (r->aln[k].read_aln != EMPTY) free(r->aln[k].read_aln);
    }
    free(r->aln); free(r->H); free(r->T); free(r);
}

/* Matrix storage a caller may lend to sw_oracle_align_buf so that a worker thread does not go
 * through mmap/munmap for every pair (the JVM's allocator does not either). */
typedef struct { int32_t *H; char *T; size_t cap; } sw_oracle_scratch;

static sw_oracle_result *align_impl(const unsigned char *ref, int64_t n,
                                    const unsigned char *read, int64_t m,
                                    const int32_t scores[3], const char types[4],
                                    int tie_mode, int keep_matrices, sw_oracle_scratch *sc);

/* SmithWaterman.java:62-92 (tie_mode 0) / DistributedSW.java:77-104 (tie_mode 1). */
sw_oracle_result *sw_oracle_align(const unsigned char *ref, int64_t n,
                                  const unsigned char *read, int64_t m,
                                  const int32_t scores[3], const char types[4],
                                  int tie_mode, int keep_matrices) {
    return align_impl(ref, n, read, m, scores, types, tie_mode, keep_matrices, NULL);
}

static sw_oracle_result *align_impl(const unsigned char *ref, int64_t n,
                                    const unsigned char *read, int64_t m,
                                    const int32_t scores[3], const char types[4],
                                    int tie_mode, int keep_matrices, sw_oracle_scratch *sc) {
    sw_oracle_result *r = (sw_oracle_result *)calloc(1, sizeof(*r));
    if (!r) return NULL;
    size_t cells = (size_t)(m + 1) * (size_t)(n + 1);
    int32_t *H;
    char *T;
    if (sc) {
        if (sc->cap < cells) {
            free(sc->H); free(sc->T);
            sc->H = (int32_t *)malloc(cells * sizeof(int32_t));
            sc->T = (char *)malloc(cells);
            sc->cap = (sc->H && sc->T) ? cells : 0;
        }
        H = sc->H; T = sc->T;
        keep_matrices = 0;
    } else {
        H = (int32_t *)malloc(cells * sizeof(int32_t));
        T = (char *)malloc(cells);
    }
    cell_list mc = { 0, 0, 0 };
    if (!H || !T) goto fail;
    int rc = (tie_mode == SW_TIE_STRICT)
        ? fill_diag_order(H, T, m, n, ref, read, scores, types, &r->score, &mc)
        : fill(H, T, m, n, ref, read, scores, types, SW_TIE_SERIAL, &r->score, &mc);
    if (rc) goto fail;
    r->n_aln = mc.n;
    r->aln = (sw_oracle_aln *)calloc((size_t)(mc.n ? mc.n : 1), sizeof(sw_oracle_aln));
    if (!r->aln) goto fail;
    for (int64_t k = 0; k < mc.n; k++)
        if (traceback(H, T, n, ref, read, types, mc.v[k].i, mc.v[k].j, &r->aln[k])) goto fail;
    if (tie_mode == SW_TIE_STRICT) stable_sort_by_begin(r->aln, r->n_aln);   /* DistributedSW:480 */
    free(mc.v);
    r->m = m; r->n = n;
    if (keep_matrices) { r->H = H; r->T = T; } else if (!sc) { free(H); free(T); }
    return r;
fail:
    free(mc.v);
    if (!sc) { free(H); free(T); }
    if (r) { r->H = NULL; r->T = NULL; sw_oracle_free(r); }
    return NULL;
}

/* accessors for ctypes */
int32_t sw_oracle_score(const sw_oracle_result *r) { return r->score; }
int64_t sw_oracle_n_aln(const sw_oracle_result *r) { return r->n_aln; }
int32_t sw_oracle_aln_begin(const sw_oracle_result *r, int64_t k) { return r->aln[k].begin; }
int32_t sw_oracle_aln_end_i(const sw_oracle_result *r, int64_t k) { return r->aln[k].end_i; }
int32_t sw_oracle_aln_end_j(const sw_oracle_result *r, int64_t k) { return r->aln[k].end_j; }
const char *sw_oracle_aln_ref(const sw_oracle_result *r, int64_t k) { return r->aln[k].ref_aln; }
const char *sw_oracle_aln_read(const sw_oracle_result *r, int64_t k) { return r->aln[k].read_aln; }
const int32_t *sw_oracle_H(const sw_oracle_result *r) { return r->H; }
const char *sw_oracle_T(const sw_oracle_result *r) { return r->T; }

/* Distribution.java:403-436 MapRef.call: total = sum of scores over reads,
 * matchSites = concatenation in read order, then stable sort by begin (:428).
 * Returned as one sw_oracle_result whose .score is the total. */
sw_oracle_result *sw_oracle_map_ref(const unsigned char *ref, int64_t n,
                                    const unsigned char *reads, const int64_t *read_off,
                                    int64_t n_reads, const int32_t scores[3],
                                    const char types[4], int tie_mode) {
    sw_oracle_result *tot = (sw_oracle_result *)calloc(1, sizeof(*tot));
    if (!tot) return NULL;
    int64_t cap = 0;
    uint32_t total = 0;
    for (int64_t q = 0; q < n_reads; q++) {
        sw_oracle_result *r = sw_oracle_align(ref, n, reads + read_off[q],
                                              read_off[q + 1] - read_off[q],
                                              scores, types, tie_mode, 0);
        if (!r) { sw_oracle_free(tot); return NULL; }
        total += (uint32_t)r->score;                                  /* :424 */
        if (tot->n_aln + r->n_aln > cap) {
            cap = (tot->n_aln + r->n_aln) * 2;
            tot->aln = (sw_oracle_aln *)realloc(tot->aln, (size_t)cap * sizeof(sw_oracle_aln));
        }
        memcpy(tot->aln + tot->n_aln, r->aln, (size_t)r->n_aln * sizeof(sw_oracle_aln)); /* :425 */
        tot->n_aln += r->n_aln;
        r->n_aln = 0;                 /* ownership of the strings moved */
        sw_oracle_free(r);
    }
    stable_sort_by_begin(tot->aln, tot->n_aln);                        /* :428 */
    tot->score = (int32_t)total;
    return tot;
}

/* ------------------------------------------------------------------------
 * CPU baseline leg for bench.py ("kind": "port"): the same full path
 * (full int matrix + type matrix, row-major fill, all tied max cells, stack
 * traceback), one pair per task, over nthreads pthreads -- the local[*]
 * equivalent of BASELINE.md Plan B.  Returns wall seconds; writes the summed
 * score and alignment count so the work cannot be optimised away.
 * ------------------------------------------------------------------------ */
typedef struct {
    const unsigned char *refs; const int64_t *ref_off; int64_t n_refs;
    const unsigned char *reads; const int64_t *read_off; int64_t n_reads;
    const int32_t *scores; const char *types; int tie_mode;
    int64_t reps;
    int64_t next;                       /* task counter (atomic) */
    pthread_mutex_t mu;
    pthread_barrier_t start, stop;      /* the clock runs between the two: thread creation and joins are outside it */
    int64_t sum_score, sum_aln, cells;
    int32_t *pair_score; int64_t *pair_naln;   /* optional per-pair outputs (pair = ref * n_reads + read) */
} bench_job;

static void *bench_worker(void *p) {
    bench_job *b = (bench_job *)p;
    int64_t ls = 0, la = 0, lc = 0;
    sw_oracle_scratch scr = { NULL, NULL, 0 };
    const int64_t per_rep = b->n_refs * b->n_reads, total = per_rep * b->reps;
    pthread_barrier_wait(&b->start);
    for (;;) {
        int64_t t = __atomic_fetch_add(&b->next, 1, __ATOMIC_RELAXED);
        if (t >= total) break;
        t %= per_rep;
        int64_t r = t / b->n_reads, q = t % b->n_reads;
        int64_t n = b->ref_off[r + 1] - b->ref_off[r], m = b->read_off[q + 1] - b->read_off[q];
        sw_oracle_result *res = align_impl(b->refs + b->ref_off[r], n,
                                           b->reads + b->read_off[q], m,
                                           b->scores, b->types, b->tie_mode, 0, &scr);
        if (res) {
            ls += res->score; la += res->n_aln; lc += m * n;
            if (b->pair_score) b->pair_score[t] = res->score;
            if (b->pair_naln) b->pair_naln[t] = res->n_aln;
            sw_oracle_free(res);
        }
    }
    pthread_barrier_wait(&b->stop);
    free(scr.H); free(scr.T);
    pthread_mutex_lock(&b->mu);
    b->sum_score += ls; b->sum_aln += la; b->cells += lc;
    pthread_mutex_unlock(&b->mu);
    return NULL;
}

/* `reps` passes over the refs x reads task list by a pool of `nthreads` threads that exists before the clock starts
 * (a persistent pool: starting 256 threads costs as much as aligning a few pairs).  The sums cover all passes. */
double sw_oracle_bench(const unsigned char *refs, const int64_t *ref_off, int64_t n_refs,
                       const unsigned char *reads, const int64_t *read_off, int64_t n_reads,
                       const int32_t scores[3], const char types[4], int tie_mode,
                       int nthreads, int reps, int64_t *sum_score, int64_t *sum_aln, int64_t *cells,
                       int32_t *pair_score, int64_t *pair_naln) {
    bench_job b;
    memset(&b, 0, sizeof(b));
    b.refs = refs; b.ref_off = ref_off; b.n_refs = n_refs;
    b.reads = reads; b.read_off = read_off; b.n_reads = n_reads;
    b.scores = scores; b.types = types; b.tie_mode = tie_mode;
    b.reps = reps < 1 ? 1 : reps;
    b.pair_score = pair_score; b.pair_naln = pair_naln;
    pthread_mutex_init(&b.mu, NULL);
    if (nthreads < 1) nthreads = 1;
    pthread_barrier_init(&b.start, NULL, (unsigned)nthreads + 1);
    pthread_barrier_init(&b.stop, NULL, (unsigned)nthreads + 1);
    pthread_t *th = (pthread_t *)malloc((size_t)nthreads * sizeof(pthread_t));
    struct timespec t0, t1;
    for (int k = 0; k < nthreads; k++) pthread_create(&th[k], NULL, bench_worker, &b);
    pthread_barrier_wait(&b.start);
    clock_gettime(CLOCK_MONOTONIC, &t0);
    pthread_barrier_wait(&b.stop);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    for (int k = 0; k < nthreads; k++) pthread_join(th[k], NULL);
    free(th);
    pthread_barrier_destroy(&b.start);
    pthread_barrier_destroy(&b.stop);
    pthread_mutex_destroy(&b.mu);
    if (sum_score) *sum_score = b.sum_score;
    if (sum_aln) *sum_aln = b.sum_aln;
    if (cells) *cells = b.cells;
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}
