// swmi_device.h -- structures shared by the host runtime (swmi_api.cpp) and the gfx950 kernels
// (swmi_kernels.hip).  HBM data layout of one batch:
//
//   seqw   uint32[]   every sequence as a BYTE image: 1 canonical code per base, 4 per dword, image start
//                     16-byte aligned.  Codes: the eight fast symbols A,C,G,T,N,U,R,Y (any case) -> 0,4,...,28 --
//                     the bit offset of the symbol's entry in a row's 8 x int4 score profile: the sweep turns a
//                     code into the one-hot word 1 << code with one SDWA shift and looks the score up with
//                     v_dot8_i32_i4 -- every other (upper-cased) byte value -> a distinct code outside that set.
//                     Each image is followed by SWMI_SEQ_PAD_WORDS zero dwords so a 16-step block may
//                     over-read past the end.
//   refs / reads      SeqDesc per sequence.
//   pairs  PairDesc[] one per (ref, read) pair of the launch (any order; the host sorts by work).
//   dir    uint32[]   direction field, 2 bits per DP cell.  For one pair and one strip of 64*R
//                     read rows it is laid out [w][k][lane]: dword ((w*R + k)*64 + lane) holds the
//                     16 anti-diagonal steps t = 16w .. 16w+15 of row i = strip*64R + lane*R + k + 1,
//                     step t at bits 2*(15 - t%16) (+1), where lane `lane` is at column j = t - lane + 1.
//                     Code: bit0 = alignment chosen, else bit1 = insertion chosen, else deletion.
//                     Every store instruction therefore writes 256 contiguous bytes.
//   cells  uint2[]    per pair, up to cell_cap (i, j) coordinates of the tied maximum cells.
//   out    PairOut[]  per pair score / count / flags.
//   rec_tab AlnRec[]  one entry per alignment, in the order the traceback kernels reserved them;
//   arena  uint32[]   the alignments' variable-length payloads (the two aligned strings, or the packed ops).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define SWMI_HD __host__ __device__
#else
#define SWMI_HD
#endif

#define SWMI_SEQ_PAD_WORDS 24u
#define SWMI_RMAX          4          // rows per lane in the widest kernel variant
#ifndef SWMI_CK_BLOCKS
#define SWMI_CK_BLOCKS     2u         // mode 1: a lane-state checkpoint every 2 blocks = 32 anti-diagonal steps
#endif                                 // (-DSWMI_CK_BLOCKS=4u builds the 64-step variant measured in profiles/r02/ck_blocks.md)
#define SWMI_CODE_PAD      0x1FFu     // never equals a base code (codes are 0..255)

// op codes of an alignment record (2 bits per traceback step)
#define SWMI_DIR_D 0u
#define SWMI_DIR_I 1u
#define SWMI_DIR_A 2u

// PairOut.flags
#define SWMI_F_DEGENERATE   0x1u   // max score 0: all m*n cells tie (no alignment records emitted)
#define SWMI_F_CELL_OVF     0x2u   // more tied max cells than cell_cap: host re-runs the pair
#define SWMI_F_ARENA_OVF    0x4u   // a record of this pair did not fit the arena: host grows it, re-runs
#define SWMI_F_DONE         0x8u   // the pair was handled whole by sw_resident_pairs_kernel: the traceback kernels leave it alone

struct SeqDesc {
    uint32_t len;     // bases
    uint32_t boff;    // dword offset of the byte image in seqw (multiple of 4)
    uint32_t acgt;    // 1 if every base is one of the eight fast symbols (codes 0,4,...,28)
    uint32_t pad;
};

struct PairDesc {
    uint32_t ref_id;
    uint32_t read_id;
    uint32_t out_id;      // index into out[] / cells[] (the pair's position in the batch)
    uint32_t pad;         // flags below (mode 1)
    uint64_t dir_off;     // dword offset of this pair's direction field in dir[]
    uint64_t seam_off;    // dword offset of the strip-seam rows, n+1 int32 per strip (multi-strip pairs only)
};

// PairDesc.pad of a single-strip pair whose sweep is split into column chunks (mode 1): this flag + the number of chunks
#define SWMI_PAD_COLS 0x80000000u
// PairDesc.pad of a pair handled start to finish by sw_resident_pairs_kernel (mode 1, small pairs): every other kernel skips it
#define SWMI_PAD_RESIDENT 0x40000000u

// One column chunk of a pair's mode-1 sweep (sw_sweep_winmax_cols_kernel): the wavefront starts a fresh sweep at
// reference column col0 + 1 (col0 a multiple of 32) -- far enough to the left that every cell from checkpoint window
// g_lo on is exact, because a positive-score path cannot span more columns than that -- and owns the checkpoint
// windows g_lo .. g_hi-1: it writes exactly the checkpoints and window maxima a one-wavefront sweep would have written.
struct ColItem {
    uint32_t pair;        // index into FillArgs.pairs
    uint32_t col0;
    uint32_t g_lo, g_hi;
};

// One wavefront of the strip pipeline (sw_sweep_winmax_strips_kernel): strip `strip` of pair `pair`, for the whole reference
// (col0 = 0, g_lo = 0, g_hi >= the strip's windows) or for ONE COLUMN CHUNK of it -- ColItem's rule holds for every strip as
// long as the seam rows of the halo are swept too: all strips of a chunk start at column col0 + 1 from a zero state, the
// chunk owns the checkpoint windows g_lo .. g_hi-1 of every strip, and strip s runs 64 steps further than strip s+1 (whose
// lane 0 needs the seam that far).  A chunk keeps PRIVATE seam rows for its own strips (its halo values are not exact) and
// writes the columns it owns into the pair's shared seam rows, which the traceback's replay reads.
struct StripItem {
    uint32_t pair, strip;
    uint32_t col0, g_lo, g_hi;
    uint32_t prog;          // index of the first progress slot of this (pair, chunk): + strip
    uint32_t priv_stride;   // dwords per private seam row (0: not chunked, the strips hand over through the shared rows)
    uint32_t pad;
    uint64_t priv_off;      // dword offset of the chunk's private seam rows in seam[]: row s at priv_off + s * priv_stride
};

struct PairOut {
    int32_t  score;
    uint32_t flags;
    uint64_t n_cells;     // number of tied maximum cells (m*n when degenerate)
};

// One alignment = one entry of the RECORD TABLE (dense, 8 dwords each, in the order the records were reserved) + its payload
// in the arena.  With TraceArgs.raw set: the two aligned strings GetAlignment returns (SmithWaterman.java:418-431), refAligned
// then readAligned, n_ops/4 + 1 dwords each (NUL-terminated, NUL-padded).  Otherwise: ceil(n_ops/16) dwords of 2-bit ops (op t
// of the traceback, first = the max cell, at bits 2*(t%16) of dword t/16), from which the host builds the strings on demand.  The table is what the host indexes: a
// dense array reads at memory bandwidth, where headers scattered through the arena were one dependent cache miss per record
// (0.11 ms per 1161 records of a pinned block the GPU had just written).
struct AlnRec {
    uint32_t out_id;
    uint32_t rank;        // position of the alignment in OptAlignments order
    int32_t  begin;       // GetAlignment's `beginning` (SmithWaterman.java:378-383)
    int32_t  end_i, end_j;
    uint32_t n_ops;
    uint32_t off_lo, off_hi;   // dword offset of the payload in the arena
};
#define SWMI_RECTAB_WORDS 8u

// ONE 64-bit counter reserves both an alignment's payload and its table entry: the low 36 bits count arena dwords (they may
// exceed the capacity: what does not fit is dropped and the host re-runs with the size the counter then shows), the high 28
// bits count records.  One atomic per alignment instead of two: device-scope atomics of every XCD on one line queue up, and
// two per record made the traceback launch 10 us longer (profiles/r03/ab_one_atomic.txt).
struct ArenaHdr {
    unsigned long long reserved;     // n_records << 36 | used_words
    unsigned long long pad;
};
#define SWMI_HDR_WORD_BITS 36u
#define SWMI_HDR_WORD_MASK ((1ull << SWMI_HDR_WORD_BITS) - 1ull)

struct FillArgs {
    const uint32_t *seqw;
    const SeqDesc  *refs;
    const SeqDesc  *reads;
    const PairDesc *pairs;
    uint32_t       *dir;
    int32_t        *seam;
    PairOut        *out;
    uint2          *cells;
    const uint64_t *cells_off;   // per out_id offset into cells[] (null: out_id * cell_cap)
    const uint32_t *cells_cap;   // per out_id capacity        (null: cell_cap)
    ArenaHdr       *hdr;         // zeroed by the fill kernel for the traceback kernel that follows it
    unsigned long long *dbg;     // optional diagnostics: per pair {slow-path entries, s_memtime ticks}
    uint32_t        dbg_thr0;    // diagnostics only: initial threshold (timing runs without the rare path)
    uint32_t        dbg_pad;
    uint32_t        n_pairs;
    uint32_t        cell_cap;
    int32_t         match, mismatch, gap;
    uint32_t        strict;      // tie mode
    uint32_t        mode;        // 0 = direction field in HBM, 1 = checkpoints + window maxima, 2 = checkpoints + event-tracked maxima
    uint32_t        skip_multi;  // mode 1: the pairs of several strips are left to sw_sweep_winmax_strips_kernel
    // mode 1, reads longer than one strip: one wavefront per STRIP, pipelined through the seam rows
    const StripItem *strip_items; // the strips of a (pair, column chunk) consecutive and ascending
    uint32_t       *progress;    // per strip item: 16-step blocks finished (the item's index is PairDesc.pad + strip)
    uint32_t        n_strip_items;
    uint32_t        pad3;
    uint32_t       *err_host;    // host-mapped word, set to 1 when a strip gave up waiting for its producer
    const ColItem  *col_items;   // mode 1: column chunks of single-strip pairs (few pairs, long references)
    uint32_t        n_col_items;
    uint32_t        strip_spins; // spin budget of the strip pipeline before it gives up (0: default)
    uint32_t       *q_reset;     // split traceback: its walk-item counter, zeroed by the sweep kernel (a memset would be one more launch)
};

struct TraceArgs {
    const uint32_t *seqw;
    const SeqDesc  *refs;
    const SeqDesc  *reads;
    const PairDesc *pairs;
    const uint32_t *dir;
    PairOut        *out;
    const uint2    *cells;
    const uint64_t *cells_off;
    const uint32_t *cells_cap;
    ArenaHdr       *hdr;
    uint32_t       *arena;
    uint64_t        arena_cap_words;
    uint32_t        n_pairs;
    uint32_t        cell_cap;
    int32_t         match, mismatch, gap;
    uint32_t        strict;
    uint32_t        lds_words;   // staging words per block for the ops of one alignment
    uint32_t        lds_read_words;   // LDS dwords reserved for the longest read's codes
    unsigned long long *dbg;     // optional diagnostics: per pair {ticks, walk ticks, steps, tiles}
    const int32_t  *seam;        // mode 1: strip seam rows for the replay of multi-strip pairs
    PairOut        *out_host;    // zero-copy results: host-mapped mirror of out[] (written by slot 0 of every pair), or null
    uint32_t       *ovf_host;    // zero-copy results: set to 1 when a record did not fit the (host-mapped) arena
    uint32_t        mode;        // same as FillArgs.mode
    uint32_t        pad2;        // set by the launcher of sw_traceback_winmax_kernel: a second set of window tiles is there (speculative staging)
    // split traceback (mode 1, tie-heavy or tiny batches): sw_detect_windows_kernel lists the maximum cells one wavefront
    // per candidate WINDOW and queues one walk item per cell, sw_walk_items_kernel walks one alignment per wavefront
    const uint32_t *win_off;     // per pair of the launch: index of its first window among all windows (n_pairs + 1 entries)
    uint32_t       *q_count;     // walk items queued
    uint4          *q_items;     // {pair index of the launch, i, j, 0}
    uint32_t        q_cap;
    uint32_t        pad4;
    // the two aligned strings of every alignment, written behind its record (swmi_emit.h): the caller's bytes as uploaded
    const uint8_t  *raw;         // null: records carry the 2-bit ops only and the host builds the strings
    const uint64_t *raw_off;     // byte offsets into raw: n_refs + 1 for the references, then n_reads + 1 for the reads
    uint32_t        raw_reads_at;   // index of the reads' first offset in raw_off (= n_refs + 1)
    uint32_t        rec_tab_cap;    // entries the record table holds
    AlnRec         *rec_tab;        // the record table (host-mapped with zero-copy results)
};

// extra arguments of sw_resident_pairs_kernel (kept out of TraceArgs: the traceback kernels are at their SGPR limit)
struct ResidentArgs {
    const uint32_t *res_items;   // indices into pairs[]: pairs whose whole direction field fits a wavefront's share of LDS
    uint32_t        n_res;
    uint32_t        res_lds_words;   // LDS dwords per wavefront (the largest resident pair of the launch)
    uint32_t        res_cell_cap;    // maximum cells a resident pair may list in LDS (more: SWMI_F_CELL_OVF, re-run by the ordinary path)
    uint32_t        res_ops_words;   // dwords of packed ops a lane can stage (the longest possible path of the launch)
};

// extra arguments of sw_tfused_kernel (swmi_tfused.hip): pairs swept in the transposed layout and traced back by the same wavefront
struct TFusedArgs {
    const uint32_t *items;       // indices into pairs[]
    uint32_t        n_items;
    uint32_t        lds_words;   // LDS dwords per wavefront: tile | cells | staged ops | reference codes | read codes
    uint32_t        tile_words;  // direction tile of one block: ceil((m_max + 63) / 16) * 64 * SWMI_TF_BR
    uint32_t        cell_cap;    // maximum cells a block lists per pass (a block with more is re-swept once per `cell_cap` cells)
    uint32_t        pad0;
    uint32_t        ref_words;   // LDS dwords for the longest reference's codes
    uint32_t        read_words;  // ... and the longest read's
    uint32_t        stage_words; // LDS dwords for the ops of ONE alignment staged one per byte (the longest possible path)
    uint32_t        debug_marks; // diagnostics: workgroup 0 leaves progress marks in the host-mapped result header
    uint32_t        n_helpers;   // wavefronts per workgroup (0..2) beyond the four that sweep: they only take block tasks
};
#define SWMI_TF_HELPERS 2u          // wavefronts per workgroup beyond the four sweepers, at most
#define SWMI_TF_BMAX   40u          // columns per lane of the transposed sweep: references up to 64 * 40 = 2560 bases
#ifndef SWMI_TF_BR
#define SWMI_TF_BR     4u           // columns per lane of a re-swept block: 256 columns (5: 320 columns, fewer walks leave the block, but every
                                    // re-sweep costs 14 % more -- measured 0.166 against 0.149 ms at the headline)
#endif
#define SWMI_TF_MAX_M  256u         // reads up to 256 bases (LDS tile of a block: (m + 63) / 16 KB)
// columns per lane for a reference of n bases (even: the kernel is instantiated for 2, 4, ..., SWMI_TF_BMAX)
SWMI_HD static inline uint32_t swmi_tf_cols_per_lane(uint32_t n) {
    uint32_t b = (n + 63u) / 64u;
    b = (b + 1u) & ~1u;
    return b < 2u ? 2u : b;
}
// dwords of column checkpoints of a pair swept by sw_tfused_kernel: one 64-lane row per step, m + L - 1 <= m + 63 steps
SWMI_HD static inline uint64_t swmi_tf_ck_words(uint32_t m) { return ((uint64_t)m + 64u) * 64u; }

#define SWMI_RANK_BY_CELL 0xFFFFFFFFu   // AlnRec.rank of the split traceback: the host orders a pair's records by (end_i, end_j)
#define SWMI_DETECT_LCAP  64u           // cells one window may hold before the pair is handed to the exact-size re-run

// rows per lane for a read of m bases
SWMI_HD static inline uint32_t swmi_rows_per_lane(uint32_t m) {
    uint32_t r = (m + 63u) / 64u;
    return r < 1u ? 1u : (r > SWMI_RMAX ? SWMI_RMAX : r);
}
// dwords of per-pair workspace.  mode 0: the direction field; mode 1: lane-state checkpoints + one maximum per
// checkpoint window; mode 2: lane-state checkpoints.
// tfused: the context lets sw_tfused_kernel take pairs (option "tfused" = 1): such a pair keeps its column checkpoints in the
// same region, which must then hold them.
SWMI_HD static inline uint64_t swmi_dir_words(uint32_t m, uint32_t n, uint32_t mode, bool tfused = false) {
    uint32_t R = swmi_rows_per_lane(m);
    uint64_t strips = ((uint64_t)m + 64u * R - 1u) / (64u * R);
    uint64_t wblocks = ((uint64_t)n + 63u + 15u) / 16u;   // T = n + 63 steps at most
    if (mode == 0) return strips * wblocks * R * 64u;
    uint64_t n_ck = (wblocks + SWMI_CK_BLOCKS - 1u) / SWMI_CK_BLOCKS;
    uint64_t words = strips * (n_ck * (R + 2u) * 64u + (mode == 1 ? ((n_ck + 63u) & ~(uint64_t)63u) : 0u));
    if (tfused && mode == 1 && m <= 256u && n <= 64u * 40u && ((uint64_t)m + 64u) * 64u > words) words = ((uint64_t)m + 64u) * 64u;
    return words;
}
// int32 seam rows of a pair whose read spans several strips: one row of n+1 per strip
SWMI_HD static inline uint64_t swmi_seam_words(uint32_t m, uint32_t n) {
    if (m <= 64u * SWMI_RMAX) return 0;
    uint64_t strips = ((uint64_t)m + 64u * SWMI_RMAX - 1u) / (64u * SWMI_RMAX);
    return strips * ((uint64_t)n + 1u);
}
