/*
 * swmi.h -- C ABI of the MI355X-native Smith-Waterman batch aligner (libswmi.so).
 *
 * This is the drop-in boundary for ONE hot path of elizabethfong/SparkSmithWaterman:
 * the matrix fill + tied-maximum search + traceback that
 *     JavaRDD.mapToPair(new MapRef())                 src/sw/Distribution.java:337-338
 *       -> MapRef.call(tuple)                         src/sw/Distribution.java:403-436
 *         -> SmithWaterman.OptAlignments.call(...)    src/sw/SmithWaterman.java:62-92
 * runs once per (reference, read) pair.  The reference has no FFI of its own; the
 * entry points below are what a JNI binding for that seam binds (INTEGRATION.md
 * shows the Java/JNI side).  Plain C types only: pointers, sizes, POD structs.
 *
 * Conventions
 *   - every function returning int returns SWMI_OK (0) or a negative swmi_status;
 *     the message for the last failure on the calling thread: swmi_last_error().
 *   - sequences are byte strings (Java chars narrowed to ISO-8859-1).  Two bases are
 *     equal iff Character.toUpperCase of the two chars is equal, as AlignmentScore
 *     tests (src/sw/SmithWaterman.java:309-318), reproduced exactly for Latin-1.  A JNI
 *     caller holding chars above U+00FF must canonicalise them first.
 *   - a batch is the cross product refs x reads, pair index = ref * n_reads + read:
 *     the order in which MapRef.call loops (Distribution.java:419-426).
 *   - a context is bound to one GPU and owns one HIP stream; calls on one context
 *     are serialised internally, use one context per host thread for concurrency
 *     (Spark runs MapRef on every executor thread).
 *   - there is NO CPU fallback: if no gfx950 device / kernel image is usable the
 *     calls fail with SWMI_ERR_NO_DEVICE.
 */
#ifndef SWMI_H
#define SWMI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SWMI_ABI_VERSION 3

typedef enum swmi_status {
    SWMI_OK               =  0,
    SWMI_ERR_INVALID      = -1,  /* bad argument (null pointer, non-monotone offsets, ...)   */
    SWMI_ERR_NO_DEVICE    = -2,  /* no usable MI355X / HIP runtime error at start-up          */
    SWMI_ERR_HIP          = -3,  /* a HIP call failed; see swmi_last_error()                  */
    SWMI_ERR_NOMEM        = -4,  /* host or device allocation failed                          */
    SWMI_ERR_UNSUPPORTED  = -5,  /* e.g. alignTypes a/i/d not pairwise distinct, len >= 2^30  */
    SWMI_ERR_RANGE        = -6   /* index out of range in an accessor                         */
} swmi_status;

/* Which reference aligner's tie-breaking is reproduced. */
#define SWMI_TIE_SERIAL 0  /* SmithWaterman.GetCellScore, '>=' chain: a > i > d (SmithWaterman.java:223-249);
                              max cells in row-major order (SmithWaterman.java:157-185)                        */
#define SWMI_TIE_STRICT 1  /* DistributedSW.GetCellScore, '>' chain: d > i > a (DistributedSW.java:305-330);
                              max cells per anti-diagonal, ascending j (DistributedSW.java:209-239), then a
                              stable sort of the alignments by beginning (DistributedSW.java:480)              */

/* alignScores {match, mismatch, gap} and alignTypes {a, i, d, none}: the two arrays
 * OptAlignments.call takes (SmithWaterman.java:47-57); defaults Distribution.java:36-37. */
typedef struct swmi_params {
    int32_t match;       /* default  5 */
    int32_t mismatch;    /* default -3 */
    int32_t gap;         /* default -4 (linear) */
    int32_t tie_mode;    /* SWMI_TIE_SERIAL | SWMI_TIE_STRICT */
    char    types[4];    /* default {'a','i','d','-'}; only used to validate distinctness */
} swmi_params;

typedef struct swmi_ctx    swmi_ctx;     /* per-(thread, device) context                  */
typedef struct swmi_batch  swmi_batch;   /* sequences resident in HBM + device workspaces */

/* ---- library / context --------------------------------------------------------- */
int         swmi_abi_version(void);
const char *swmi_last_error(void);                       /* thread-local, never NULL */
int         swmi_device_count(int *count);
int         swmi_create(int device, swmi_ctx **out);     /* device = HIP ordinal      */
void        swmi_destroy(swmi_ctx *ctx);
void        swmi_default_params(swmi_params *p);

/* Tuning knobs (all optional).  cell_cap: tied-maximum cells kept per pair in the
 * fast path (pairs with more are re-run on the GPU with an exact-size list);
 * max_workspace_bytes: cap on the per-batch workspace arena (larger batches are run in
 * chunks); profiling = 1 brackets every stage with HIP events (sweep, traceback, D2H: three marker packets per run, ~10 us of a
 * 0.17 ms step), 2 only the sweep (two packets; swmi_timing.traceback_ms stays 0), 0 (default) nothing; zero_copy (default 1): kernels
 * write results straight into pinned host memory instead of a D2H copy; device_strings (default 1): the traceback kernels
 * write both aligned strings of every alignment behind its packed ops (from the caller's bytes as uploaded), so the
 * alignment accessors hand out pointers; 0: records carry the 2-bit ops only and the host builds a string when it is asked
 * for (smaller result stream, e.g. when only a few of many alignments will ever be read); mode selects the kernel
 * pipeline -- results are identical in all of them:
 *   1 (default)  the sweep computes scores only, leaves lane-state checkpoints and ONE maximum per
 *                32-step window; the traceback re-sweeps the windows holding the pair's maximum to list
 *                its cells, and the windows each alignment path crosses to get direction bits (into LDS).
 *                Needs mismatch <= 0 and gap <= 0; other scores run as mode 2.
 *   2            as 1, but tied maxima are tracked during the sweep (per-step test + rare handler).
 *   0            the sweep writes the whole 2-bit direction field to HBM and the traceback reads it
 *                (cheaper when most pairs have many tied maxima).  Batches with pairs longer than about
 *                16 k bases (m + n) run as mode 1/2: mode 0's traceback tiles leave too little LDS for them.
 *  -1            the same as 1 (kept for callers that passed "automatic").
 * tb_split: grain of the mode-1 traceback.  0: one workgroup per pair (lists the maximum cells, then up to four waves walk
 *                the alignments, the others re-sweep windows for them) -- best when pairs have one or a few alignments;
 *                1: split -- one wavefront per checkpoint window lists cells, one wavefront per alignment walks; -1
 *                (default): split for launches of fewer than 64 pairs, and for batches of up to 256 pairs in which a sample
 *                of the pairs (aligned once, on the first run of a batch) averages >= auto_ties_x100 / 100 tied maxima
 *                per pair -- periodic references, the reference's own EngineerData sets.
 * resident: -1 (default) pairs whose whole direction field fits 20 KB of LDS and whose reference is at most 8 x the read
 *                (80 bp reads x 400 bp references, the reference's EngineerData shapes) are handled start to finish by one
 *                wavefront -- two sweeps inside LDS, all alignments walked at once, one per lane; 0 never; 1 whenever the
 *                field fits 40 KB.
 * tfused: 1: the usual pair -- fast symbols on both sides, scores within int4, gap < 0, a read of at most 256 bases, a reference
 *                of at most 2560 -- is swept in the TRANSPOSED layout (reference columns on the lanes, the read streaming
 *                through) and traced back in the same launch (sw_tfused_kernel: block tasks and walk items shared by the
 *                wavefronts of a workgroup); 0 or -1 (default): never -- measured slower than the two-kernel pipeline.
 * scores_only (default 0): 1 = the sweep only: every pair's score and MapRef's totals (swmi_pair_score, swmi_ref_total(s),
 *                swmi_stream_totals, swmi_batch_pair_results with n_alignments = NULL); the tied-maximum lists and the alignments
 *                are not computed and their accessors fail.  For a driver that reduces to the winning references first and aligns
 *                only those in full (the reference's driver discards every other reference's alignments, Distribution.java:341-353).
 * stream_keep_records (default 1): 0 = a stream drops every chunk's alignment records once its scores, counts and totals are
 *                taken -- for a driver that reduces to the winning references and aligns those again (Distribution.java:341-353
 *                discards every other reference's alignments too); the alignment accessors of such chunks fail.
 * Further knobs: spin_us (how long a run polls its stream before it blocks, default 2000); col_chunks (0 automatic,
 * 1 never, N > 1 force up to N column chunks per pair: a launch of few pairs with long references is swept by several
 * wavefronts per pair -- a read of more than 256 rows by several strip pipelines); debug_strip_spins / debug_reverse_strips (tests of the strip pipeline's give-up path). */
int         swmi_set_option(swmi_ctx *ctx, const char *name, int64_t value);

/* ---- staged path: upload once, run many times (what bench.py times) ------------- */
/* ref_off/read_off have n+1 entries, off[0] == 0, non-decreasing; lengths < 2^30.  */
int  swmi_batch_upload(swmi_ctx *ctx,
                       const uint8_t *ref_bytes, const uint64_t *ref_off, uint32_t n_refs,
                       const uint8_t *read_bytes, const uint64_t *read_off, uint32_t n_reads,
                       swmi_batch **out);
/* Fill + direction field + max-cell lists + device traceback for every pair, then
 * the compact result records device->host.  Synchronous on return. */
int  swmi_batch_run(swmi_ctx *ctx, swmi_batch *b, const swmi_params *p);
void swmi_batch_free(swmi_ctx *ctx, swmi_batch *b);
/* The same run on the context's own host thread: swmi_batch_run_async returns at once, swmi_batch_wait blocks until
 * the run has finished and returns its status (one run in flight per context; results and accessors as after
 * swmi_batch_run, to be used after the wait).  What a Spark task uses to prepare its next partition -- or bench.py's
 * rank to do the previous shard's reduce -- while the GPU works. */
int  swmi_batch_run_async(swmi_ctx *ctx, swmi_batch *b, const swmi_params *p);
int  swmi_batch_wait(swmi_ctx *ctx);

/* Stage timings of the last swmi_batch_run with option "profiling" = 1 (ms, HIP events
 * on the context's stream): fill kernel, traceback kernel, D2H; launches = number of
 * fill launches the figures sum over. */
typedef struct swmi_timing {
    float    fill_ms, traceback_ms, d2h_ms, total_ms;
    uint32_t fill_launches, rerun_pairs;
    uint64_t cells;             /* sum of m*n over the pairs of the run            */
    uint64_t dir_bytes;         /* direction-field bytes written                   */
    uint32_t strip_fallbacks;   /* launches repeated with the one-wavefront sweep after the strip pipeline gave up */
    uint32_t col_chunks;        /* column chunks the sweep of the run was split into (0: one sweep per pair): one wavefront each, one strip pipeline each for reads of several strips */
    uint32_t resident_pairs;    /* pairs handled whole by one wavefront with the direction field in LDS             */
    uint32_t tfused_pairs;      /* pairs swept in the transposed layout and traced back by the same wavefront        */
} swmi_timing;
int  swmi_batch_timing(const swmi_batch *b, swmi_timing *t);
/* The kernel pipeline (0, 1 or 2, see swmi_set_option "mode") the last run of the batch used. */
int  swmi_batch_mode(const swmi_batch *b, int *mode);

/* ---- results of the last run (host memory owned by the batch) ------------------- */
#define SWMI_PAIR_DEGENERATE 0x1u   /* max score 0: every one of the m*n cells is a "max cell" and
                                       yields (0,"","") -- SmithWaterman.java:154,182-185,378-380 */
uint64_t swmi_batch_n_pairs(const swmi_batch *b);
int      swmi_pair_score(const swmi_batch *b, uint64_t pair, int32_t *score);
int      swmi_pair_n_alignments(const swmi_batch *b, uint64_t pair, uint64_t *n, uint32_t *flags);
/* all pairs at once: scores[n] and/or n_alignments[n] (either may be NULL), n = swmi_batch_n_pairs */
int      swmi_batch_pair_results(const swmi_batch *b, int32_t *scores, uint64_t *n_alignments, uint64_t n);
/* k-th alignment of the pair in OptAlignments order.  *ref_aln / *read_aln point to
 * NUL-terminated strings owned by the batch (valid until the next run/free);
 * characters keep the caller's original case, gaps are '_' (SmithWaterman.java:356). */
int      swmi_pair_alignment(swmi_batch *b, uint64_t pair, uint64_t k,
                             int32_t *begin, int32_t *end_i, int32_t *end_j,
                             const char **ref_aln, const char **read_aln, uint32_t *len);

/* Builds the record index and both strings of EVERY alignment of the batch in one native call (what a caller that
 * consumes all of OptAlignments' output pays); returns the number of alignments (degenerate pairs count m*n) and
 * of characters built. */
int      swmi_batch_materialise_all(swmi_batch *b, uint64_t *n_alignments, uint64_t *n_chars);

/* ---- MapRef view: per reference, over all reads (Distribution.java:403-436) ------ */
/* total = sum over reads of the pair scores (Java int, wrapping) (:424). */
int      swmi_ref_total(const swmi_batch *b, uint32_t ref, int32_t *total);
/* all n_refs totals at once into totals[n_refs] (what the driver's max/top-K reduce consumes, :341-353) */
int      swmi_ref_totals(const swmi_batch *b, int32_t *totals, uint32_t n);
/* matchSites = the reads' alignment lists concatenated in read order (:425), then
 * stably sorted by ascending begin (:428, MatchSiteComp :691-694). */
int      swmi_ref_n_match_sites(swmi_batch *b, uint32_t ref, uint64_t *n);
int      swmi_ref_match_site(swmi_batch *b, uint32_t ref, uint64_t k, int32_t *begin,
                             const char **ref_aln, const char **read_aln, uint32_t *len);

/* MapRef's output for the references ref_lo .. ref_hi-1 in ONE call -- what a per-partition binding hands back
 * (Distribution.java:419-433) instead of three calls per match site.  Per reference r (index r - ref_lo): totals[] (:424),
 * degenerate[] = how many leading (0, "", "") sites it has (pairs whose maximum is 0 contribute m*n each,
 * SmithWaterman.java:154,182-185; they are counted, not listed) and site_first[] .. site_first[+1] = its real match sites in
 * MapRef order (stable sort by begin, :428) within begins[] / lens[] / str_off[]: site s has refAligned at blob + str_off[s]
 * and readAligned at blob + str_off[s] + lens[s], lens[s] bytes each, no terminators.  site_first has ref_hi - ref_lo + 1
 * entries.  *n_sites / *blob_bytes always receive the sizes needed: call with begins = NULL to ask for them, then with
 * buffers of at least that capacity (too small: SWMI_ERR_RANGE).  totals / degenerate / site_first may be NULL. */
int      swmi_ref_sites_packed(swmi_batch *b, uint32_t ref_lo, uint32_t ref_hi,
                               int32_t *totals, uint64_t *degenerate, uint64_t *site_first,
                               int32_t *begins, uint32_t *lens, uint64_t *str_off, uint64_t sites_cap,
                               uint8_t *blob, uint64_t blob_cap, uint64_t *n_sites, uint64_t *blob_bytes);

/* ---- one-shot path: what a per-partition JNI call binds -------------------------- */
/* upload + run; results are read with the accessors above; free with swmi_batch_free. */
int  swmi_align_batch(swmi_ctx *ctx, const swmi_params *p,
                      const uint8_t *ref_bytes, const uint64_t *ref_off, uint32_t n_refs,
                      const uint8_t *read_bytes, const uint64_t *read_off, uint32_t n_reads,
                      swmi_batch **out);

/* ---- streaming: a reference set larger than one batch ------------------------------ */
/* The reference reads a whole FASTA file (InOutOps.GetRefSeqs, InOutOps.java:115-168) and then maps it
 * (Distribution.java:329-338).  A stream aligns the reads given at open against references that arrive in chunks:
 * host threads parse / copy chunk k+1 into pinned memory while `slots` workers (each its own HIP stream and device
 * buffers on the context's GPU) upload the raw bytes, canonicalise them on the GPU, run the full path and keep the
 * results of chunks k, k-1, ...  After swmi_stream_finish every chunk's results are a swmi_batch to which all
 * swmi_pair_* / swmi_ref_* accessors apply (reference indices local to the chunk; swmi_stream_chunk gives the offset).
 * slots = 0 and chunk_bytes = 0 pick the defaults (3 slots, 32 MiB of sequence per chunk). */
typedef struct swmi_stream swmi_stream;
typedef struct swmi_stream_stats {
    double   push_ms;            /* wall time inside swmi_stream_push_file                                  */
    double   parse_ms;           /* summed over the parser threads                                          */
    double   upload_ms, run_ms;  /* summed over the slot workers: H2D + encode, sweep + traceback + results */
    double   gpu_sweep_ms, gpu_traceback_ms;   /* HIP-event times, option "profiling" = 1                   */
    uint64_t bytes, cells;
    uint32_t chunks, pad;
} swmi_stream_stats;
int      swmi_stream_open(swmi_ctx *ctx, const swmi_params *p, const uint8_t *read_bytes, const uint64_t *read_off,
                          uint32_t n_reads, uint32_t slots, uint64_t chunk_bytes, swmi_stream **out);
/* references from memory (copied before the call returns; kept for the alignment strings) */
int      swmi_stream_push(swmi_stream *s, const uint8_t *ref_bytes, const uint64_t *ref_off, uint32_t n_refs);
/* references from a FASTA file with GetRefSeqs' line rules (swmi_io.h), parsed segment-wise by parse_threads host
 * threads (0: 6); the file stays mapped until the stream is closed and a reference's bytes are re-read from it when
 * one of its alignment strings is asked for.  One file per stream. */
int      swmi_stream_push_file(swmi_stream *s, const char *path, const char *delimiter, uint32_t parse_threads);
int      swmi_stream_finish(swmi_stream *s);                  /* blocks until every chunk is done */
uint64_t swmi_stream_n_refs(const swmi_stream *s);
uint32_t swmi_stream_n_chunks(const swmi_stream *s);
int      swmi_stream_chunk(swmi_stream *s, uint32_t k, swmi_batch **batch, uint64_t *first_ref);
int      swmi_stream_totals(const swmi_stream *s, int32_t *totals, uint64_t n);   /* MapRef totals of all references */
int      swmi_stream_metadata(const swmi_stream *s, uint64_t ref, char *buf, size_t cap);
int      swmi_stream_get_stats(const swmi_stream *s, swmi_stream_stats *st);
void     swmi_stream_close(swmi_stream *s);

#ifdef __cplusplus
}
#endif
#endif /* SWMI_H */
