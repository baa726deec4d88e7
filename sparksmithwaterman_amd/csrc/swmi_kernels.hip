// swmi_kernels.hip -- gfx950 (MI355X / CDNA4) kernels of the Smith-Waterman hot path.
//
// Replaces, for a whole batch of (reference, read) pairs at once, what the reference does per pair in
//   ScoreMatrix.call   src/sw/SmithWaterman.java:129-190  (fill, max-cell list)
//   GetCellScore.call  src/sw/SmithWaterman.java:217-252  (cell rule, tie order)   [DistributedSW.java:305-330 for strict]
//   GetAlignment.call  src/sw/SmithWaterman.java:354-436  (traceback)
//
// sw_fill_kernel: ONE WAVEFRONT (64 lanes) PER PAIR, anti-diagonal systolic sweep.
//   lane l owns R consecutive read rows (i = strip*64R + l*R + k + 1, k < R) and at step t works on
//   reference column j = t - l + 1, so the 64 lanes sit on one anti-diagonal band.  Per step a lane needs
//     W  = its own H of the previous step              (register)
//     N  = lane l-1's bottom-row H of the previous step (one DPP wave_shr:1, no LDS)
//     NW = the N it received one step earlier           (register carry)
//     the reference base of column j                    (flows down the lanes by a second DPP shift;
//                                                        lane 0 is fed from a scalar register)
//   int32 scores live in registers only; nothing but the 2-bit direction field is written to HBM, as
//   256-byte coalesced stores ([w][k][lane] layout, swmi_device.h).  Integer recurrence: no MFMA.
//   Rows beyond 64*R (long reads) are processed strip after strip; the seam row between two strips
//   goes through a small per-pair buffer.
//   Tied maxima: a wave-uniform threshold `thr` (scalar register) holds the running maximum; only when
//   some lane reaches it does the wave leave the hot loop to append (i,j) to the pair's cell list
//   (clearing it on a strict increase, exactly like SmithWaterman.java:176-185).
//
// sw_traceback_kernel: one wavefront per pair; orders the tied cells as the reference would list
//   them, walks each path through the direction field while tracking the score arithmetically
//   (H(pred) = H - delta, so `while (score > 0)` of SmithWaterman.java:380 needs no score matrix),
//   stages the 2-bit ops in LDS and appends one variable-length record per alignment to an arena.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdlib.h>
#include "swmi_device.h"
#include "swmi_emit.h"

#define WAVE 64
#ifndef SWMI_HELPER_SLEEP
#define SWMI_HELPER_SLEEP 8          // x 64 cycles between two polls of an idle helper wavefront of the traceback (measured: profiles/r02/ab_helper_sleep.txt)
#endif
#define BALLOT(pred) __builtin_amdgcn_ballot_w64(pred)
// LDS hand-offs between lanes of ONE wavefront: DS operations of a wave execute in order, so a compiler-level
// fence is all that is needed (a workgroup barrier would deadlock the fused kernel's 4 independent waves)
#define WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)

// ------------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------------
// v_mov_b32_dpp wave_shr:1 : lane l receives lane l-1's value, lane 0 keeps `old`.
__device__ __forceinline__ int wave_shr1(int old, int src) {
    return __builtin_amdgcn_update_dpp(old, src, 0x138 /*wave_shr:1*/, 0xf, 0xf, false);
}
// same, lane 0 receives 0 (bound_ctrl)
__device__ __forceinline__ int wave_shr1_zero(int src) {
    return __builtin_amdgcn_update_dpp(0, src, 0x138 /*wave_shr:1*/, 0xf, 0xf, true);
}

// wave-wide signed max through the DPP network (no LDS): 4 row steps, 2 row broadcasts, result read from lane 63.
// The DPP modifier sits on the v_max itself (a lane without a source lane, or outside the row mask, is simply not written: it
// keeps its value, which is what max(v, v) gave before): 6 VALU + the wait states a DPP read of a just-written register
// needs, where `update_dpp` + max compiled to a copy, a v_mov_b32_dpp and a v_max per step -- 24 instructions per window
// close of the sweep, 0.4 per anti-diagonal step.
__device__ __forceinline__ int wave_max_i32(int v) {
#ifndef SWMI_NO_ASM
    asm volatile("s_nop 1\n\t"
                 "v_max_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_max_i32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_max_i32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_max_i32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"      // lane 15 of every row holds the row's max
                 "v_max_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n\t"   // into rows 1 and 3
                 "v_max_i32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 1"         // into rows 2 and 3: lane 63 holds the wave's max
                 : "+v"(v));
#else
#define SWMI_DPP_MAX(ctrl, rmask)                                                          \
    { int o_ = __builtin_amdgcn_update_dpp(v, v, ctrl, rmask, 0xf, false); v = v > o_ ? v : o_; }
    SWMI_DPP_MAX(0x111, 0xf)   // row_shr:1
    SWMI_DPP_MAX(0x112, 0xf)   // row_shr:2
    SWMI_DPP_MAX(0x114, 0xf)   // row_shr:4
    SWMI_DPP_MAX(0x118, 0xf)   // row_shr:8   -> lane 15 of every row holds the row's max
    SWMI_DPP_MAX(0x142, 0xa)   // row_bcast:15 into rows 1 and 3
    SWMI_DPP_MAX(0x143, 0xc)   // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave's max
#undef SWMI_DPP_MAX
#endif
    return __builtin_amdgcn_readlane(v, 63);
}

__device__ __forceinline__ uint32_t lanemask_lt_count(uint64_t mask) {
    // number of set bits of `mask` below this lane
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// load served by L2 (bypasses this CU's L1): for data another wave -- or this wave, earlier -- stored in the same launch
__device__ __forceinline__ uint32_t ld_l2(const uint32_t *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// A value every lane of the wave holds alike, moved to a scalar register: what hangs on it (loop bounds, branches,
// base addresses) then runs on the scalar unit instead of as exec-masked vector code.
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint64_t uni64(uint64_t v) { return ((uint64_t)uni((uint32_t)(v >> 32)) << 32) | uni((uint32_t)v); }

__device__ __forceinline__ uint32_t seq_code(const uint32_t *__restrict__ w, uint32_t pos) {
    return (w[pos >> 2] >> (8u * (pos & 3u))) & 0xFFu;
}

// The 8 x int4 score profiles of a lane's R rows (fast symbols): nibble c/4 of row k's profile is the score of the row's base
// against reference symbol c.  The R base codes are loaded FIRST, unconditionally (images are padded), so that the loads are in
// flight together: a load behind each `row < m` test serialised R memory latencies in front of every window re-sweep.
template <int R>
__device__ __forceinline__ void build_profiles(int (&q)[R], const uint32_t *__restrict__ readw, const uint32_t row0, const uint32_t m,
                                               const int match, const int mismatch) {
    uint32_t c[R];
#pragma unroll
    for (int k = 0; k < R; ++k) c[k] = seq_code(readw, row0 + k) & 28u;      // 0, 4, ..., 28
#pragma unroll
    for (int k = 0; k < R; ++k) {
        uint32_t p = (uint32_t)(mismatch & 0xF) * 0x11111111u;
        if (row0 + k < m) p = (p & ~(0xFu << c[k])) | ((uint32_t)(match & 0xF) << c[k]);
        q[k] = (int)p;
    }
}

// ------------------------------------------------------------------------------------------------
// the R cells of one lane in one step (previous column's H in hin, this column's H to hout)
// ------------------------------------------------------------------------------------------------
// the fast cell stream looks scores up in a profile of 8 x int4 (v_dot8_i32_i4): match and mismatch must fit
#define SWMI_SCORES_FIT(A) ((A).match >= -8 && (A).match <= 7 && (A).mismatch >= -8 && (A).mismatch <= 7)

#include "swmi_cells_gen.inc"   // CellsAsm<R, ACGT, STRICT, DIRS>: hand-scheduled instruction stream

// Plain C++ statement of the same update (build with -DSWMI_NO_ASM to A/B against the asm stream).
template <int R, bool ACGT, bool STRICT, bool DIRS>
struct CellsRef {
    static __device__ __forceinline__ void step(const int (&hin)[R], int (&hout)[R], uint32_t (&acc)[R], const int (&q)[R],
                                                int rb, int diag, int up, int gap, int vmat, int vmis) {
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const int left = hin[k];
            int sc;
            if (ACGT) sc = rb ? __builtin_amdgcn_sbfe(q[k], (unsigned)__builtin_ctz((unsigned)rb), 4u) : 0;   // rb = 1 << 4*symbol
            else      sc = (rb == q[k]) ? vmat : vmis;
            const int a = diag + sc;                       // SmithWaterman.java:244 / AlignmentScore :309-318
            const int t2 = (up > left ? up : left) + gap;   // :227, :235 (InsDelScore :277-280)
            int hv = a > t2 ? a : t2;
            hv = hv > 0 ? hv : 0;                           // `int max = 0` :223
            if (DIRS) {
                const bool bi = STRICT ? (up > left) : (up >= left);
                const bool ba = STRICT ? (a > t2) : (a >= t2);
                acc[k] = (acc[k] << 2) | (bi ? 2u : 0u) | (ba ? 1u : 0u);
            }
            diag = left;
            up = hv;
            hout[k] = hv;
        }
    }
};

#ifdef SWMI_NO_ASM
template <int R, bool ACGT, bool STRICT, bool DIRS> using Cells = CellsRef<R, ACGT, STRICT, DIRS>;
#else
template <int R, bool ACGT, bool STRICT, bool DIRS> using Cells = CellsAsm<R, ACGT, STRICT, DIRS>;
#endif

// What a 16-step block does besides the scores:
//   SWMI_MODE_FIELD   sweep, direction bits packed and stored to HBM, tied maxima tracked by events          (mode 0 sweep)
//   SWMI_MODE_SCORE   sweep, scores only (5 VALU per cell) + checkpoints, tied maxima tracked by events      (mode 2 sweep)
//   SWMI_MODE_REPLAY  a checkpoint-to-checkpoint window re-swept, direction bits to LDS, nothing tracked     (mode 1/2 traceback)
//   SWMI_MODE_WINMAX  sweep, scores only + checkpoints; per lane only a running maximum (no compare, no branch,
//                     no cell list): the wave reduces it to ONE maximum per checkpoint window                 (mode 1 sweep)
//   SWMI_MODE_DETECT  a window re-swept like REPLAY that also lists its cells equal to the pair's maximum     (mode 1 traceback)
#define SWMI_MODE_FIELD  0
#define SWMI_MODE_SCORE  1
#define SWMI_MODE_REPLAY 2
#define SWMI_MODE_WINMAX 3
#define SWMI_MODE_DETECT 4

// ------------------------------------------------------------------------------------------------
// rare path: at step t some lane reached the running maximum.  thr / cnt are wave-uniform; they travel
// packed in one 64-bit value.
// ------------------------------------------------------------------------------------------------
template <int R>
__device__ __forceinline__ unsigned long long
record_max_cells(int h0, int h1, int h2, int h3, uint32_t t, uint32_t lane_eff, uint32_t n, uint32_t row0, uint32_t m,
                 int thr, uint32_t cnt, uint2 *__restrict__ cells, uint32_t ccap) {
    const int hh[4] = {h0, h1, h2, h3};
    const uint32_t c0 = t - lane_eff;                  // column index j-1 the lane worked on at step t
    const bool active = c0 < n;
    int v[R];
    int cand = -1;
#pragma unroll
    for (int k = 0; k < R; ++k) {
        v[k] = (active && row0 + k < m) ? hh[k] : -1;   // rows past the read and lanes off their range never count
        cand = cand > v[k] ? cand : v[k];
    }
    if (BALLOT(cand >= thr) == 0) return ((unsigned long long)(uint32_t)thr << 32) | cnt;   // stale trigger
    // strict increase: climb to the wave's maximum by lane hops (no reduction network needed: few lanes exceed)
    uint64_t gt = BALLOT(cand > thr);
    while (gt) {                                        // SmithWaterman.java:176-181
        thr = __builtin_amdgcn_readlane(cand, (int)__builtin_ctzll(gt));
        cnt = 0;
        gt = BALLOT(cand > thr);
    }
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const bool hit = v[k] == thr;                   // :182-185
        const uint64_t hm = BALLOT(hit);
        if (hm) {
            const uint32_t pos = cnt + lanemask_lt_count(hm);
            if (hit && pos < ccap) cells[pos] = make_uint2(row0 + k + 1, c0 + 1u);
            cnt += (uint32_t)__popcll(hm);
        }
    }
    return ((unsigned long long)(uint32_t)thr << 32) | cnt;
}

// ------------------------------------------------------------------------------------------------
// sweep state of one wavefront
// ------------------------------------------------------------------------------------------------
#ifdef SWMI_STRIP_DIAG
#define SWMI_SD(...) __VA_ARGS__
#else
#define SWMI_SD(...)
#endif
template <int R>
struct FillState {
    int h[R];            // H of the lane's rows: read by even steps of a block, written by odd ones
    int g[R];            // ... and the other way round (ping-pong, see fill_block16)
    uint32_t acc[R];     // direction bits of the last <= 16 steps
    int q[R];            // ACGT (= fast symbols): the row's 8 x int4 score profile; else the read's base code
    int nprev, rb;
    int thr;             // wave-uniform running maximum
    uint32_t cnt;        // wave-uniform number of cells equal to thr
    uint64_t ev_prev;    // lanes whose previous step reached thr (handled one step late, see below)
    int lmax;            // WINMAX: this lane's maximum H since the last checkpoint
    uint32_t events;     // slow-path entries (diagnostics only)
    bool dbg_skip;       // diagnostics only
#ifdef SWMI_STRIP_DIAG
    unsigned long long dg_pub;    // ticks spent waiting before progress publications
#endif
};

// read-side operands of this lane's rows, and a zero H column
template <int R, bool ACGT>
__device__ __forceinline__ void setup_rows(FillState<R> &S, const uint32_t *__restrict__ readw, uint32_t row0, uint32_t m,
                                           int match, int mismatch) {
    if (ACGT) {
        build_profiles<R>(S.q, readw, row0, m, match, mismatch);      // 8 signed score nibbles indexed by the reference code
    } else {
        uint32_t c[R];
#pragma unroll
        for (int k = 0; k < R; ++k) c[k] = seq_code(readw, row0 + k);
#pragma unroll
        for (int k = 0; k < R; ++k) S.q[k] = row0 + k < m ? (int)c[k] : (int)SWMI_CODE_PAD;
    }
#pragma unroll
    for (int k = 0; k < R; ++k) {
        S.h[k] = 0;
        S.g[k] = 0;
        S.acc[k] = 0;
    }
    S.nprev = 0;                                     // N received one step earlier = NW of this step
    S.rb = 0;                                        // reference base operand of this lane's current column
}

template <int R>
__device__ __forceinline__ void handle_pending(FillState<R> &S, const int (&hv)[R], uint32_t t, uint32_t lane_eff,
                                               uint32_t n, uint32_t row0, uint32_t m, uint2 *__restrict__ cells, uint32_t ccap) {
    if (S.dbg_skip) {          // diagnostics: price of the branch alone (results are wrong in this mode)
        S.events++;
        S.thr += 1;
        return;
    }
    const unsigned long long tc = record_max_cells<R>(
        hv[0], R > 1 ? hv[R > 1 ? 1 : 0] : 0, R > 2 ? hv[R > 2 ? 2 : 0] : 0, R > 3 ? hv[R > 3 ? 3 : 0] : 0,
        t, lane_eff, n, row0, m, S.thr, S.cnt, cells, ccap);
    S.events++;
    S.thr = __builtin_amdgcn_readfirstlane((int)(tc >> 32));
    S.cnt = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)tc);
}

typedef uint32_t Words4 __attribute__((ext_vector_type(4)));
typedef const Words4 __attribute__((address_space(4))) *ConstWords4;     // constant address space: uniform loads become s_load

// acc with lane `l` (0..15, a constant after unrolling) replaced by the scalar v
__device__ __forceinline__ int writelane_const(int acc, const int v, const uint32_t l) {
#define SWMI_WL(L) case L: asm("v_writelane_b32 %0, %1, " #L : "+v"(acc) : "s"(v)); break;
    switch (l) {
    SWMI_WL(0) SWMI_WL(1) SWMI_WL(2) SWMI_WL(3) SWMI_WL(4) SWMI_WL(5) SWMI_WL(6) SWMI_WL(7)
    SWMI_WL(8) SWMI_WL(9) SWMI_WL(10) SWMI_WL(11) SWMI_WL(12) SWMI_WL(13) SWMI_WL(14) SWMI_WL(15)
    }
#undef SWMI_WL
    return acc;
}

// 16 anti-diagonal steps t = t0 .. t0+15.  PRED: lanes may be outside their column range (ramp-up /
// ramp-down blocks); otherwise every lane below `lact` is inside it for all 16 steps.
//
// The H registers ping-pong between S.h (read by even steps) and S.g (read by odd steps), so after step t
// the values of step t-1 are still there.  That lets the tied-maximum test of step t-1 -- a compare into
// an SGPR pair -- be branched on one step later, when its result has long arrived, instead of stalling the
// wave on a VALU->scalar-branch dependency every step (measured: 57 of 197 cycles per step).
template <int R, bool ACGT, bool STRICT, bool MULTI, bool PRED, int MODE, bool PIPE = false>
__device__ __forceinline__ void fill_block16(FillState<R> &S, const uint4 w, const uint32_t t0,
                                             const uint32_t lane, const uint32_t lane_eff,
                                             const uint32_t n, const uint32_t m, const uint32_t row0,
                                             const int gap, const int vmat, const int vmis,
                                             const int seamv, const bool reads_seam, const bool feeds_seam,
                                             int32_t *__restrict__ seam_out,
                                             uint2 *__restrict__ cells, const uint32_t ccap,
                                             uint32_t *pub_slot = nullptr, const uint32_t pub_val = 0u,
                                             int32_t *__restrict__ seam_sh = nullptr, const uint32_t own_lo = 0u, const uint32_t own_hi = 0u) {
    constexpr bool DIRS = MODE == SWMI_MODE_FIELD || MODE == SWMI_MODE_REPLAY || MODE == SWMI_MODE_DETECT;
    constexpr bool TRACK = MODE == SWMI_MODE_FIELD || MODE == SWMI_MODE_SCORE;     // deferred tied-maximum events
    constexpr bool LMAX = MODE == SWMI_MODE_WINMAX;                                  // per-lane running maximum only
    constexpr bool DETECT = MODE == SWMI_MODE_DETECT;                                // list the cells equal to S.thr
    constexpr bool FEEDS = MODE == SWMI_MODE_FIELD || MODE == SWMI_MODE_SCORE || MODE == SWMI_MODE_WINMAX;   // sweep (writes seam rows)
    using C = Cells<R, ACGT, DIRS ? STRICT : false, DIRS>;
    int seam_acc = 0;
    // PIPE: "the blocks before this one are complete" (pub_val) is published as late as possible before this block's own
    // seam stores: the wait then covers stores that were issued a block ago, not a moment ago
    auto publish = [&]() {
        if (PIPE && pub_val) {
            SWMI_SD(const unsigned long long dg2 = __builtin_amdgcn_s_memtime();)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            SWMI_SD(S.dg_pub += __builtin_amdgcn_s_memtime() - dg2;)
            if (lane == 0) __hip_atomic_store(pub_slot, pub_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    if (PRED) publish();
#pragma unroll
    for (uint32_t s = 0; s < 16; ++s) {
        const int (&hin)[R] = (s & 1u) ? S.g : S.h;
        int (&hout)[R] = (s & 1u) ? S.h : S.g;
        const uint32_t wsel = s < 4 ? w.x : s < 8 ? w.y : s < 12 ? w.z : w.w;
        // base code of column t0+s+1 (lane 0).  ACGT: codes are 0, 4, ..., 28 and travel down the lanes ONE-HOT
        // (1 << code) so that one v_dot8_i32_i4 with the row's score profile yields NW + s(ref, read); the SDWA byte
        // select makes extract + shift a single instruction.
        int feed;
        if (ACGT) {
            switch (s & 3u) {
            case 0:  asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD" : "=v"(feed) : "v"(wsel), "v"(1)); break;
            case 1:  asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "=v"(feed) : "v"(wsel), "v"(1)); break;
            case 2:  asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD" : "=v"(feed) : "v"(wsel), "v"(1)); break;
            default: asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD" : "=v"(feed) : "v"(wsel), "v"(1)); break;
            }
        } else {
            feed = (int)((wsel >> (8u * (s & 3u))) & 0xFFu);
        }
        S.rb = wave_shr1(feed, S.rb);
        int nin;
        if (MULTI) {
            // seamv is 0 in every lane of a strip with no seam above it: no branch on reads_seam (it cost an exec-masked
            // branch per step, 9 instructions where 3 do)
            nin = wave_shr1(__builtin_amdgcn_readlane(seamv, s), hin[R - 1]);
        } else {
            nin = wave_shr1_zero(hin[R - 1]);
        }
        int mrow = -1;
        if (PRED) {
            const uint32_t c0 = t0 + s - lane_eff;                         // column index j-1 of this lane
            if (c0 < n) {
                C::step(hin, hout, S.acc, S.q, S.rb, S.nprev, nin, gap, vmat, vmis);
                if (TRACK || DETECT) {
                    mrow = hout[0];
#pragma unroll
                    for (int k = 1; k < R; ++k) mrow = mrow > hout[k] ? mrow : hout[k];
                }
                if (LMAX) {
#pragma unroll
                    for (int k = 0; k < R; ++k) S.lmax = S.lmax > hout[k] ? S.lmax : hout[k];
                }
                if (MULTI && FEEDS && feeds_seam && lane == WAVE - 1) {
                    if (PIPE) __hip_atomic_store(seam_out + c0 + 1, hout[R - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    else      seam_out[c0 + 1] = hout[R - 1];
                    if (PIPE && seam_sh && c0 + 1u > own_lo && c0 + 1u <= own_hi) seam_sh[c0 + 1] = hout[R - 1];   // (column chunk: the columns it owns)
                }
            } else {
#pragma unroll
                for (int k = 0; k < R; ++k) hout[k] = hin[k];             // a lane off its range keeps its state
            }
        } else {
            C::step(hin, hout, S.acc, S.q, S.rb, S.nprev, nin, gap, vmat, vmis);
            if (TRACK || DETECT) {
                mrow = hout[0];
#pragma unroll
                for (int k = 1; k < R; ++k) mrow = mrow > hout[k] ? mrow : hout[k];
            }
            if (LMAX) {
                if (R == 3) {
                    // 6 new values per two steps = three v_max3: the last row of an even step waits for the odd one
                    // (its register is still live there thanks to the ping-pong)
                    if (s & 1u) {
                        int x = S.lmax > hin[2] ? S.lmax : hin[2];  x = x > hout[0] ? x : hout[0];
                        x = x > hout[1] ? x : hout[1];              S.lmax = x > hout[2] ? x : hout[2];
                    } else {
                        const int x = S.lmax > hout[0] ? S.lmax : hout[0];
                        S.lmax = x > hout[1] ? x : hout[1];
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < R; ++k) S.lmax = S.lmax > hout[k] ? S.lmax : hout[k];
                }
            }
            // seam row: lane 63's last row of this step goes to lane s of seam_acc (v_readlane + v_writelane); one 64-byte
            // store per block below instead of an exec-masked branch, a 64-bit address and a one-lane store per step
            if (MULTI && FEEDS) seam_acc = writelane_const(seam_acc, __builtin_amdgcn_readlane(hout[R - 1], WAVE - 1), s);
        }
        S.nprev = nin;
        if (DETECT) {
            // replay of a window that holds the pair's maximum: list its cells equal to it (immediate branch: 32 steps only)
            if (BALLOT(mrow >= S.thr) != 0) {
                const uint32_t c0d = t0 + s - lane_eff;
                const bool act = c0d < n;
#pragma unroll
                for (int k = 0; k < R; ++k) {
                    const bool hit = act && (row0 + k < m) && (hout[k] == S.thr);      // SmithWaterman.java:182-185
                    const uint64_t hm = BALLOT(hit);
                    if (hm) {
                        const uint32_t pos = S.cnt + lanemask_lt_count(hm);
                        if (hit && pos < ccap) cells[pos] = make_uint2(row0 + k + 1, c0d + 1u);
                        S.cnt += (uint32_t)__popcll(hm);
                    }
                }
            }
        }
        if (TRACK) {
            const uint64_t ev = BALLOT(mrow >= S.thr);      // all 64 lanes vote: thr / cnt stay wave-uniform
            if (__builtin_expect(S.ev_prev != 0, 0))                           // step t0+s-1, values still in hin
                handle_pending<R>(S, hin, t0 + s - 1u, lane_eff, n, row0, m, cells, ccap);
            S.ev_prev = ev;
        }
    }
    if (!PRED) publish();
    if (MULTI && FEEDS && !PRED && feeds_seam && lane < 16u) {
        // a steady block of a strip that feeds a seam has all 64 lanes on rows: lane 63 was on column t0 + s - 62 (1-based)
        // at step s.  PIPE: another wavefront (possibly on another XCD) is already reading this row: device-coherent store
        const uint32_t col = t0 - (WAVE - 2u) + lane;
        int32_t *dst = seam_out + col;
        if (PIPE) __hip_atomic_store(dst, seam_acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else      *dst = seam_acc;
        if (PIPE && seam_sh && col > own_lo && col <= own_hi) seam_sh[col] = seam_acc;
    }
}

// geometry shared by the sweep and its replay
struct StripGeom {
    uint32_t rps, n_strips, wblocks, n_ck;
    uint64_t strip_words;        // dwords of workspace per strip
    uint64_t wmax_off;           // mode 1: offset of the strip's per-window maxima inside its workspace
};
// hmode = the pipeline the host selected: 0 direction field, 1 checkpoints + window maxima, 2 checkpoints only
template <int R>
__device__ __forceinline__ StripGeom strip_geom(uint32_t m, uint32_t n, uint32_t hmode) {
    StripGeom g;
    g.rps = WAVE * R;
    g.n_strips = (m + g.rps - 1) / g.rps;
    g.wblocks = (n + 63u + 15u) / 16u;                       // 16-step blocks reserved per strip
    g.n_ck = (g.wblocks + SWMI_CK_BLOCKS - 1u) / SWMI_CK_BLOCKS;
    g.wmax_off = (uint64_t)g.n_ck * (R + 2) * WAVE;
    g.strip_words = hmode == 0 ? (uint64_t)g.wblocks * R * WAVE
                               : g.wmax_off + (hmode == 1 ? (uint64_t)((g.n_ck + 63u) & ~63u) : 0u);
    return g;
}

// ------------------------------------------------------------------------------------------------
// fill: one pair, one wavefront.  R rows per lane; ACGT = both sequences made of the eight fast symbols
// (A,C,G,T,N,U,R,Y -- the template parameter kept its first name) and match/mismatch fit a
// signed nibble (profile lookup by v_dot8_i32_i4 instead of compare+select); STRICT = DistributedSW tie order;
// MULTI = more than one strip of 64*R rows (seam rows through memory); MODE = FIELD or SCORE.
// ------------------------------------------------------------------------------------------------
// PIPE (mode 1, MULTI): this wavefront sweeps only strip `my_strip`; the wavefront of strip s-1 runs a few blocks ahead
// and publishes its progress, the one of strip s+1 follows -- a systolic pipeline of strips over wavefronts, so a
// 10 kbp read is swept in about the time of ONE strip instead of 40.
#define SWMI_PIPE_PUBLISH 2u      // blocks between two publications of a strip's progress
template <int R, bool ACGT, bool STRICT, bool MULTI, int MODE, bool PIPE = false>
__device__ __forceinline__ void fill_pair(const FillArgs &A, const PairDesc pd, const uint32_t lane, const uint32_t my_strip = 0u,
                                          const StripItem *item = nullptr) {
    const SeqDesc rd = A.refs[pd.ref_id];
    const SeqDesc qd = A.reads[pd.read_id];
    // PIPE: the geometry in scalar registers, so that the block counter is one and the reference words can come through
    // the scalar cache: a vector load's s_waitcnt vmcnt also waits for every store issued before it, the device-scope seam
    // stores among them
    const uint32_t n_full = PIPE ? uni(rd.len) : rd.len, m = PIPE ? uni(qd.len) : qd.len;
    // PIPE, column chunk (StripItem): the sweep starts at reference column col0 + 1 from a zero state and owns the windows
    // g_lo .. g_hi-1; the whole reference is the chunk {0, 0, all windows}
    const uint32_t col0 = PIPE ? uni(item->col0) : 0u;
    const uint32_t g_lo = PIPE ? uni(item->g_lo) : 0u, g_hi = PIPE ? uni(item->g_hi) : 0xFFFFFFFFu;
    const uint32_t priv_stride = PIPE ? uni(item->priv_stride) : 0u;
    const uint32_t *__restrict__ refw = A.seqw + (PIPE ? uni(rd.boff) + (col0 >> 2) : rd.boff);
    const uint32_t *__restrict__ readw = A.seqw + qd.boff;
    const int match = A.match, mismatch = A.mismatch, gap = A.gap;
    constexpr uint32_t HMODE = MODE == SWMI_MODE_FIELD ? 0u : (MODE == SWMI_MODE_WINMAX ? 1u : 2u);
    const StripGeom G = strip_geom<R>(m, n_full, HMODE);
    int pair_max = 0;        // WINMAX: maximum over the finished windows

    const uint64_t cbase = A.cells_off ? A.cells_off[pd.out_id] : (uint64_t)pd.out_id * A.cell_cap;
    const uint32_t ccap = A.cells_cap ? A.cells_cap[pd.out_id] : A.cell_cap;
    uint2 *__restrict__ cells = A.cells + cbase;

    FillState<R> S;
    S.thr = A.dbg ? (int)A.dbg_thr0 : 1;   // a max of 0 never enters the list: that is the degenerate case
    S.cnt = 0;
    S.ev_prev = 0;
    S.events = 0;
    S.dbg_skip = A.dbg && (A.dbg_pad != 0);
#ifdef SWMI_STRIP_DIAG
    S.dg_pub = 0;
#endif
    const unsigned long long t_start = A.dbg ? __builtin_amdgcn_s_memtime() : 0ull;

    const uint32_t s_begin = PIPE ? my_strip : 0u, s_end = PIPE ? my_strip + 1u : G.n_strips;
    uint32_t *__restrict__ progress = PIPE ? A.progress + uni(item->prog) : nullptr;
    for (uint32_t s = s_begin; s < s_end; ++s) {
        const uint32_t row0 = s * G.rps + lane * R;        // 0-based first row of this lane
        const uint32_t rows_left = m - s * G.rps;
        const uint32_t lact = rows_left >= G.rps ? WAVE : (rows_left + R - 1) / R;   // lanes holding rows
        // columns this strip sweeps (from col0 + 1 on).  A column chunk that is not the pair's last ends after its last
        // window -- 64 steps later per strip below this one, whose lane 0 needs the seam that far -- unless that is past
        // the reference's end
        uint32_t n = n_full - col0;
        bool truncated = false;
        if (PIPE && g_hi < G.n_ck) {
            const uint32_t ext = 16u * SWMI_CK_BLOCKS * g_hi - col0 + WAVE * (G.n_strips - 1u - s);
            if (ext < n) { n = ext; truncated = true; }
        }
        const uint32_t T = truncated ? n : n + lact - 1; // steps of this strip
        const uint32_t lane_eff = lane < lact ? lane : 0x40000000u;   // lanes without rows are never in range
        setup_rows<R, ACGT>(S, readw, row0, m, match, mismatch);
        S.lmax = -1;

        uint32_t *__restrict__ wsp = A.dir + pd.dir_off + s * G.strip_words + lane;   // this strip's workspace
        // WINMAX: one maximum per checkpoint window.  Lanes without rows are left out; the pad rows of the last lane with
        // rows cannot exceed the real cells they derive from (mismatch <= 0 and gap <= 0 are required for this mode), so the
        // pair's maximum is exact and a window can at worst be listed without holding a maximum cell.
        auto close_window = [&](uint32_t g) {
            const int wm = wave_max_i32(lane < lact ? S.lmax : -1);
            if (lane == 0) A.dir[pd.dir_off + s * G.strip_words + G.wmax_off + g] = (uint32_t)wm;
            pair_max = pair_max > wm ? pair_max : wm;
            S.lmax = -1;
        };
        const int32_t *seam_in = nullptr;
        int32_t *seam_out = nullptr;
        int32_t *seam_sh = nullptr;                                   // column chunk: the pair's shared row, for the columns the chunk owns
        if (MULTI) {
            int32_t *sb = A.seam + pd.seam_off;                       // row s = H of the strip's last read row
            seam_in = sb + (uint64_t)(s > 0 ? s - 1 : 0) * (n_full + 1);
            seam_out = sb + (uint64_t)s * (n_full + 1);
            if (PIPE && priv_stride) {
                int32_t *pb = A.seam + item->priv_off;                // the chunk's own rows, indexed by local column
                seam_sh = seam_out + col0;
                seam_in = pb + (uint64_t)(s > 0 ? s - 1 : 0) * priv_stride;
                seam_out = pb + (uint64_t)s * priv_stride;
            }
        }
        const uint32_t step_w = 16u * SWMI_CK_BLOCKS;
        const uint32_t own_lo = step_w * g_lo > col0 ? step_w * g_lo - col0 : 0u;                  // owned local columns: own_lo < c <= own_hi
        const uint32_t own_hi = g_hi < G.n_ck ? step_w * g_hi - col0 : 0xFFFFFFFFu;
        const bool feeds_seam = MULTI && (s + 1 < G.n_strips);
        const bool reads_seam = MULTI && (s > 0);

        const uint32_t nblk = (T + 15u) / 16u;
        const uint4 *__restrict__ refq = reinterpret_cast<const uint4 *>(refw);   // images are 16-byte aligned
        const ConstWords4 refq_s = (ConstWords4)(uintptr_t)refw;                  // the same through the scalar cache (read-only data)
        auto ref_words = [&](uint32_t i) -> uint4 {
            if (PIPE) { const Words4 v = refq_s[i]; return make_uint4(v.x, v.y, v.z, v.w); }
            return refq[i];
        };
        uint4 wnext = ref_words(0u);
        // The seam row above this strip is read in GROUPS of 64 columns (4 blocks): seam_in[64g + 1 + lane], one coalesced
        // load per group, issued one group ahead; a block takes its 16 values (N of lane 0 for its 16 steps) from the
        // group register with one ds_bpermute.
        // PIPE: column c of the seam row is stored by the producer's lane 63 at step c + 62, so the columns of block x are
        // complete when the producer has finished block x + 4; the producer publishes its progress every SWMI_PIPE_PUBLISH
        // blocks and the polled value is kept, so a consumer that is behind does not poll at all (a poll and a wait for
        // the stores' acknowledgements per block: 0.415 ms at 257 x 4000; every 4 blocks, 24 polls per 254 blocks: 0.357;
        // every 2 blocks with the next group asked for 2 blocks ahead instead of 4 costs two strips 2 % and gains a
        // 40-strip pipeline 3 %: profiles/r03/strip_pipeline.md).
        uint32_t nblk_prod = (n_full - col0 + WAVE - 1u + 15u) / 16u;
        if (truncated || (PIPE && g_hi < G.n_ck)) {
            const uint32_t ext_p = 16u * SWMI_CK_BLOCKS * g_hi - col0 + WAVE * (G.n_strips - s);      // (strip s-1's extent)
            if (ext_p < n_full - col0) nblk_prod = ext_p / 16u;
        }
        bool gave_up = false;
        uint32_t prod_seen = 0u;
#ifdef SWMI_STRIP_DIAG
        // -DSWMI_STRIP_DIAG + SWMI_DEBUG_FILL=1: where a strip's wavefront waits (s_memtime ticks, 10 ns)
        unsigned long long dg_poll = 0, dg_grp = 0, dg_polls = 0;
#endif
        auto load_group = [&](uint32_t g) -> int {
            const uint32_t col = 64u * g + 1u + lane;
            if (64u * g >= n) return 0;
            if (PIPE) {
                const uint32_t need = 4u * g + 8u < nblk_prod ? 4u * g + 8u : nblk_prod;
                // the give-up is progress-based: the budget (~60 ms of s_sleep by default) restarts whenever the producer
                // advances, so a slow producer is waited for and only one that does not move at all is abandoned -- the host
                // then re-runs the chunk with the one-wavefront sweep, which needs no other workgroup (swmi_api.cpp)
                const uint32_t budget = A.strip_spins ? A.strip_spins : (1u << 18);
                uint32_t spins = 0;
                SWMI_SD(const unsigned long long dg0 = __builtin_amdgcn_s_memtime(); if (prod_seen < need) dg_polls++;)
                while (prod_seen < need && !gave_up) {
                    const uint32_t p = __builtin_amdgcn_readfirstlane(
                        __hip_atomic_load(progress + (s - 1u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                    if (p != prod_seen) { prod_seen = p; spins = 0; continue; }
                    __builtin_amdgcn_s_sleep(8);
                    if (++spins > budget) {
                        gave_up = true;
                        if (lane == 0 && A.err_host) *A.err_host = 1u;
                    }
                }
                SWMI_SD(dg_poll += __builtin_amdgcn_s_memtime() - dg0;)
            }
            return col <= n ? __hip_atomic_load(seam_in + col, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
        };
        int seam_grp = 0, seam_grp_next = reads_seam ? load_group(0u) : 0;
        for (uint32_t tb = 0; tb < nblk; ++tb) {
            const uint4 w = wnext;                       // base codes of columns 16tb+1 .. 16tb+16
            wnext = ref_words(tb + 1u);                  // prefetch (images are padded)
            const uint32_t t0 = 16u * tb;
            const uint32_t tbg = tb + (col0 >> 4);       // the block's number in the pair's own sweep (col0 is a multiple of 32)
            const uint32_t gw = tbg / SWMI_CK_BLOCKS;    // ... and its window
            if (MODE == SWMI_MODE_WINMAX && (tbg % SWMI_CK_BLOCKS) == 0u && tb > 0u) {
                if (gw > g_lo && gw <= g_hi) close_window(gw - 1u); else S.lmax = -1;
            }
            if ((MODE == SWMI_MODE_SCORE || MODE == SWMI_MODE_WINMAX) && (tbg % SWMI_CK_BLOCKS) == 0u && gw >= g_lo && gw < g_hi) {
                // checkpoint: everything a replay of steps t0.. needs from this lane ([ck][slot][lane], 256 B stores)
                uint32_t *__restrict__ ck = wsp + (uint64_t)gw * (R + 2) * WAVE;
#pragma unroll
                for (int k = 0; k < R; ++k) ck[k * WAVE] = (uint32_t)S.h[k];
                ck[R * WAVE] = (uint32_t)S.nprev;
                ck[(R + 1) * WAVE] = (uint32_t)S.rb;
            }
            if (reads_seam && (tb & 3u) == 0u) {
                SWMI_SD(const unsigned long long dg1 = __builtin_amdgcn_s_memtime();)
                seam_grp = seam_grp_next;
                SWMI_SD(asm volatile("s_waitcnt vmcnt(0)" : "+v"(seam_grp) :: "memory"); dg_grp += __builtin_amdgcn_s_memtime() - dg1;)
            }
            // the next group is asked for two blocks before it is needed, not four: every block of distance is a block a
            // strip trails the one above it, and a 10 kbp read is a pipeline of 40 strips
            if (reads_seam && (tb & 3u) == 2u) seam_grp_next = load_group(tb / 4u + 1u);
            // lanes 0..15: the block's 16 values (0 in every lane of a strip without a seam above it)
            const int seamv = reads_seam ? __builtin_amdgcn_ds_bpermute((int)(((tb & 3u) << 6) + ((lane & 15u) << 2)), seam_grp) : 0;
            const bool steady = (t0 + 1u >= lact) && (t0 + 15u < n);     // (a truncated strip never leaves the reference)
            // progress value p = "blocks 0 .. p-1 are complete", published every SWMI_PIPE_PUBLISH blocks, one block late
            uint32_t *pub_slot = PIPE ? progress + s : nullptr;
            const uint32_t pub_val = (PIPE && feeds_seam && tb > 0u && (tb % SWMI_PIPE_PUBLISH) == 0u) ? tb : 0u;
            if (steady)
                fill_block16<R, ACGT, STRICT, MULTI, false, MODE, PIPE>(S, w, t0, lane, lane_eff, n, m, row0, gap, match, mismatch,
                                                                        seamv, reads_seam, feeds_seam, seam_out, cells, ccap,
                                                                        pub_slot, pub_val, seam_sh, own_lo, own_hi);
            else
                fill_block16<R, ACGT, STRICT, MULTI, true, MODE, PIPE>(S, w, t0, lane, lane_eff, n, m, row0, gap, match, mismatch,
                                                                       seamv, reads_seam, feeds_seam, seam_out, cells, ccap,
                                                                       pub_slot, pub_val, seam_sh, own_lo, own_hi);
            if (PIPE && feeds_seam && tb + 1u == nblk) {
                // the strip is complete once its last seam stores have left the CU (they are device-coherent stores)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0) __hip_atomic_store(progress + s, nblk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }

            if (MODE == SWMI_MODE_FIELD) {
                // ---- end of a 16-step block: one coalesced 256 B store per row slot -------------------
                // lanes that finished their last column inside this block still owe the missing shifts
                const int miss = (int)(t0 + 15u) - ((int)(lane + n) - 1);
#pragma unroll
                for (int k = 0; k < R; ++k) {
                    uint32_t v = S.acc[k];
                    if (miss > 0 && miss < 16) v <<= 2 * miss;
                    wsp[((uint64_t)tb * R + k) * WAVE] = v;
                }
            }
        }
        if (MODE == SWMI_MODE_WINMAX) {
            const uint32_t gl = (nblk - 1u + (col0 >> 4)) / SWMI_CK_BLOCKS;
            if (gl >= g_lo && gl < g_hi) close_window(gl);
        }
#ifdef SWMI_STRIP_DIAG
        if (PIPE && A.dbg && lane == 0 && s < 2u) {
            const unsigned long long tot = __builtin_amdgcn_s_memtime() - t_start;
            // strip 0: {lifetime, waiting before publications}; strip 1: {lifetime, polls, waiting in polls, waiting for seam groups}
            A.dbg[2 * pd.out_id + s] = s == 0u ? (tot << 32) | (S.dg_pub & 0xFFFFFFFFull)
                                               : (tot << 40) | ((dg_polls & 0xFFull) << 32) | ((dg_poll & 0xFFFFull) << 16) | (dg_grp & 0xFFFFull);
        }
#endif
        // the tied-maximum test of the strip's last step is still pending (16 steps per block: its H is in S.h)
        if (S.ev_prev != 0) {
            handle_pending<R>(S, S.h, 16u * nblk - 1u, lane_eff, n, row0, m, cells, ccap);
            S.ev_prev = 0;
        }
        if (MULTI) __threadfence();    // seam row of this strip visible before the next strip reads it
    }

    if (PIPE) {
        // combine the strips: one atomicMax each into the record sw_sweep_winmax_kernel zeroed one launch earlier; the
        // traceback kernels complete it (finish_pair).  No fence: see sweep_fast.
        if (lane == 0 && pair_max > 0) atomicMax(&A.out[pd.out_id].score, pair_max);
        return;
    }
    if (lane == 0) {
        PairOut o;
        if (MODE == SWMI_MODE_WINMAX) {
            // the cells holding the maximum are listed by the traceback kernel (n_cells follows there)
            if (pair_max <= 0) { o.score = 0; o.flags = SWMI_F_DEGENERATE; o.n_cells = (uint64_t)m * n_full; }
            else               { o.score = pair_max; o.flags = 0u; o.n_cells = 0; }
        } else if (S.cnt == 0) { o.score = 0; o.flags = SWMI_F_DEGENERATE; o.n_cells = (uint64_t)m * n_full; }
        else            { o.score = S.thr; o.flags = S.cnt > ccap ? SWMI_F_CELL_OVF : 0u; o.n_cells = S.cnt; }
        A.out[pd.out_id] = o;
        if (A.dbg) {
            A.dbg[2 * pd.out_id] = S.events;
            A.dbg[2 * pd.out_id + 1] = __builtin_amdgcn_s_memtime() - t_start;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// mode-1 sweep, fast symbols, one strip (m <= 256): the headline path.  Same results, checkpoints and window maxima
// as fill_pair<..., SWMI_MODE_WINMAX>, from a shorter instruction stream (tools/gen_step.py: 3 VALU per cell, the
// neighbour exchanges folded into DPP arithmetic, 14.5 instructions per step at R = 3 instead of 17.8).
// COLS: this wavefront sweeps only the column chunk `ci` of the pair (swmi_device.h: ColItem).
// ------------------------------------------------------------------------------------------------
#include "swmi_step_gen.inc"   // SweepStep4Asm<R>: four steps per asm statement

template <int R>
struct SweepState {
    int h[R], g[R];      // H of the lane's rows: h is read by even steps and written by odd ones, g the other way round
    int hp[R];           // max(H + gap, 0) of the same rows, updated in place
    int q[R];            // the row's 8 x int4 score profile
    int rbx, rby;        // one-hot reference symbol: rbx is what an even step consumes (it prepares rby for the odd one)
    int lmax;            // this lane's maximum H since the last window boundary
};

// plain statement of one step of SweepStep4Asm::run.  Used for the blocks in which some lane has run past the last column
// (`in_range` false: the lane keeps its state), and for every block when built with -DSWMI_NO_ASM.
template <int R>
__device__ __forceinline__ void sweep_step_ref(const int (&hin)[R], int (&hout)[R], int (&hp)[R], const int (&q)[R],
                                               const int rb, int &rbn, const uint32_t feed_code, const uint32_t gm,
                                               int &lm, const bool in_range) {
    const int nw = wave_shr1_zero(hout[R - 1]);      // lane l-1's bottom row two steps ago = NW of row 0 (lane 0: 0)
    const int upp = wave_shr1_zero(hp[R - 1]);       // max(N + gap, 0) of row 0 (lane 0: 0)
    if (in_range) {
        int diag = nw, up = upp;
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const int a = __builtin_amdgcn_sdot8(q[k], rb, diag, false);    // SmithWaterman.java:244 (AlignmentScore :309-318)
            diag = hin[k];
            int hv = a > up ? a : up;                                         // :227-240: max(W + gap, N + gap, 0) is max(hp, hp)
            hv = hv > hp[k] ? hv : hp[k];
            hout[k] = hv;
            hp[k] = (uint32_t)hv > gm ? (int)((uint32_t)hv - gm) : 0;
            up = hp[k];
            lm = lm > hv ? lm : hv;
        }
    } else {
#pragma unroll
        for (int k = 0; k < R; ++k) hout[k] = hin[k];
    }
    rbn = wave_shr1((int)(1u << (feed_code & 31u)), rb);
}

// 16 steps of a block in which some lane runs past the last column.  Such a lane goes on computing -- nobody reads its values:
// the lane below it took what it needed one step earlier -- but its window maximum must not see them: SweepStepTailAsm updates
// the maximum under a lane mask.  (-DSWMI_NO_ASM: the plain statement, which also leaves such a lane's state alone.)
template <int R>
__device__ __forceinline__ void sweep_tail_block(SweepState<R> &S, const uint4 w, const uint32_t wnext_x, const uint32_t t0,
                                                 const uint32_t lane_eff, const uint32_t n, const int one, const uint32_t gm) {
#ifndef SWMI_NO_ASM
    const uint32_t c = t0 - lane_eff;              // column index (0-based) of this lane at step t0; lanes without rows: far outside
    SweepStepTailAsm<R, 0>::run(S.h, S.g, S.hp, S.q, S.rbx, S.rby, w.x, w.y, one, gm, S.lmax, c + 0u, n);
    SweepStepTailAsm<R, 1>::run(S.h, S.g, S.hp, S.q, S.rbx, S.rby, w.x, w.y, one, gm, S.lmax, c + 1u, n);
    SweepStepTailAsm<R, 2>::run(S.h, S.g, S.hp, S.q, S.rbx, S.rby, w.x, w.y, one, gm, S.lmax, c + 2u, n);
    SweepStepTailAsm<R, 3>::run(S.h, S.g, S.hp, S.q, S.rbx, S.rby, w.x, w.y, one, gm, S.lmax, c + 3u, n);
    SweepStepTailAsm<R, 0>::run(S.h, S.g, S.hp, S.q, S.rbx, S.rby, w.y, w.z, one, gm, S.lmax, c + 4u, n);
    SweepStepTailAsm<R, 1>::run(S.h, S.g, S.hp, S.q, S.rbx, S.rby, w.y, w.z, one, gm, S.lmax, c + 5u, n);
    SweepStepTailAsm<R, 2>::run(S.h, S.g, S.hp, S.q, S.rbx, S.rby, w.y, w.z, one, gm, S.lmax, c + 6u, n);
    SweepStepTailAsm<R, 3>::run(S.h, S.g, S.hp, S.q, S.rbx, S.rby, w.y, w.z, one, gm, S.lmax, c + 7u, n);
    SweepStepTailAsm<R, 0>::run(S.h, S.g, S.hp, S.q, S.rbx, S.rby, w.z, w.w, one, gm, S.lmax, c + 8u, n);
    SweepStepTailAsm<R, 1>::run(S.h, S.g, S.hp, S.q, S.rbx, S.rby, w.z, w.w, one, gm, S.lmax, c + 9u, n);
    SweepStepTailAsm<R, 2>::run(S.h, S.g, S.hp, S.q, S.rbx, S.rby, w.z, w.w, one, gm, S.lmax, c + 10u, n);
    SweepStepTailAsm<R, 3>::run(S.h, S.g, S.hp, S.q, S.rbx, S.rby, w.z, w.w, one, gm, S.lmax, c + 11u, n);
    SweepStepTailAsm<R, 0>::run(S.h, S.g, S.hp, S.q, S.rbx, S.rby, w.w, wnext_x, one, gm, S.lmax, c + 12u, n);
    SweepStepTailAsm<R, 1>::run(S.h, S.g, S.hp, S.q, S.rbx, S.rby, w.w, wnext_x, one, gm, S.lmax, c + 13u, n);
    SweepStepTailAsm<R, 2>::run(S.h, S.g, S.hp, S.q, S.rbx, S.rby, w.w, wnext_x, one, gm, S.lmax, c + 14u, n);
    SweepStepTailAsm<R, 3>::run(S.h, S.g, S.hp, S.q, S.rbx, S.rby, w.w, wnext_x, one, gm, S.lmax, c + 15u, n);
#else
#pragma unroll
    for (uint32_t s = 0; s < 16; ++s) {
        const uint32_t s1 = s + 1u;
        const uint32_t wf = s1 < 4 ? w.x : s1 < 8 ? w.y : s1 < 12 ? w.z : s1 < 16 ? w.w : wnext_x;
        const uint32_t code = (wf >> (8u * (s1 & 3u))) & 0xFFu;
        const bool in_range = (t0 + s - lane_eff) < n;
        if (s & 1u) sweep_step_ref<R>(S.g, S.h, S.hp, S.q, S.rby, S.rbx, code, gm, S.lmax, in_range);
        else        sweep_step_ref<R>(S.h, S.g, S.hp, S.q, S.rbx, S.rby, code, gm, S.lmax, in_range);
    }
#endif
}

template <int R, bool COLS>
__device__ __forceinline__ void sweep_fast(const FillArgs &A, const PairDesc pd, const uint32_t lane, const ColItem ci) {
    // (moving the pair's geometry to scalar registers with readfirstlane makes the block loop scalar, and the sweep 3 % slower:
    // measured, tools/ab_headline.py -- the vector-side loop control overlaps the asm groups better than the scalar one)
    const SeqDesc rd = A.refs[pd.ref_id];
    const SeqDesc qd = A.reads[pd.read_id];
    const uint32_t n_full = rd.len, m = qd.len;
    const StripGeom G = strip_geom<R>(m, n_full, 1u);
    const uint32_t col0 = COLS ? ci.col0 : 0u;
    const uint32_t g_lo = COLS ? ci.g_lo : 0u, g_hi = COLS ? ci.g_hi : G.n_ck;
    const bool last = g_hi >= G.n_ck;
    const uint32_t n = (last ? n_full : 16u * SWMI_CK_BLOCKS * g_hi) - col0;   // columns of this wavefront's (virtual) reference
    const uint32_t *__restrict__ refw = A.seqw + rd.boff + (col0 >> 2);
    const uint32_t *__restrict__ readw = A.seqw + qd.boff;
    const uint32_t lact = m >= G.rps ? WAVE : (m + R - 1) / R;           // lanes holding rows
    const uint32_t lane_eff = lane < lact ? lane : 0x40000000u;
    const uint32_t T = n + lact - 1;
    const uint32_t nblk = last ? (T + 15u) / 16u : n / 16u;               // (a chunk that is not the last ends on a window boundary)
    const uint32_t gm = (uint32_t)(-(int64_t)A.gap);                      // mode 1: gap <= 0
    const int one = 1;
    int pair_max = 0;

    const unsigned long long dbg_t0 = A.dbg ? __builtin_amdgcn_s_memtime() : 0ull;
    SweepState<R> S;
    build_profiles<R>(S.q, readw, lane * R, m, A.match, A.mismatch);
#pragma unroll
    for (int k = 0; k < R; ++k) {
        S.h[k] = 0; S.g[k] = 0; S.hp[k] = 0;
    }
    S.lmax = -1;
    const uint4 *__restrict__ refq = reinterpret_cast<const uint4 *>(refw);   // 16-byte aligned: col0 is a multiple of 32
    uint4 wnext = refq[0];
    S.rby = 0;
    S.rbx = wave_shr1((int)(1u << (wnext.x & 31u)), 0);                   // lane 0: column 1; nothing has flowed further yet

    uint32_t *__restrict__ wsp = A.dir + pd.dir_off + lane;
    auto close_window = [&](uint32_t g) {
        const int wm = wave_max_i32(lane < lact ? S.lmax : -1);
        if (lane == 0) A.dir[pd.dir_off + G.wmax_off + g] = (uint32_t)wm;
        pair_max = pair_max > wm ? pair_max : wm;
        S.lmax = -1;
    };
    for (uint32_t tb = 0; tb < nblk; ++tb) {
        const uint4 w = wnext;                            // base codes of (local) columns 16tb+1 .. 16tb+16
        wnext = refq[tb + 1];                             // prefetch (images are padded)
        const uint32_t tbg = tb + (col0 >> 4);            // the block's number in the pair's own sweep
        if ((tbg % SWMI_CK_BLOCKS) == 0u) {
            const uint32_t g = tbg / SWMI_CK_BLOCKS;
            if (tb > 0u) {
                if (g > g_lo) close_window(g - 1u); else S.lmax = -1;
            }
            if (g >= g_lo) {
                // checkpoint in the layout the replay expects: H of the rows, the N received one step earlier, the
                // reference operand of the last step ([ck][slot][lane], 256 B stores)
                uint32_t *__restrict__ ck = wsp + (uint64_t)g * (R + 2) * WAVE;
#pragma unroll
                for (int k = 0; k < R; ++k) ck[k * WAVE] = (uint32_t)S.h[k];
                ck[R * WAVE] = (uint32_t)wave_shr1_zero(S.g[R - 1]);
                ck[(R + 1) * WAVE] = (uint32_t)S.rby;
            }
        }
        const uint32_t t0 = 16u * tb;
#ifndef SWMI_NO_ASM
        if (t0 + 15u < n) {
            // lane 0 stays inside the reference for all 16 steps: nobody has to be masked (a lane that has not started
            // yet computes zeros from its zero operands, lanes without rows compute values nobody reads)
            SweepStep4Asm<R>::run(S.h, S.g, S.hp, S.q, S.rbx, S.rby, w.x, w.y, one, gm, S.lmax);
            SweepStep4Asm<R>::run(S.h, S.g, S.hp, S.q, S.rbx, S.rby, w.y, w.z, one, gm, S.lmax);
            SweepStep4Asm<R>::run(S.h, S.g, S.hp, S.q, S.rbx, S.rby, w.z, w.w, one, gm, S.lmax);
            SweepStep4Asm<R>::run(S.h, S.g, S.hp, S.q, S.rbx, S.rby, w.w, wnext.x, one, gm, S.lmax);
        } else
#endif
        {
            sweep_tail_block<R>(S, w, wnext.x, t0, lane_eff, n, one, gm);
        }
    }
    {
        const uint32_t gl = (nblk - 1u + (col0 >> 4)) / SWMI_CK_BLOCKS;
        if (gl >= g_lo) close_window(gl);
    }
    if (lane != 0) return;
    if (A.dbg) {      // diagnostics: where the wave ran (HW_ID: wave, SIMD, CU, SE ...) and how long
        A.dbg[2 * pd.out_id] = (unsigned long long)__builtin_amdgcn_s_getreg((4 /*HW_REG_HW_ID*/) | (0 << 6) | (31 << 11)) |
                               ((unsigned long long)__builtin_amdgcn_s_getreg((20 /*HW_REG_XCC_ID*/) | (0 << 6) | (31 << 11)) << 32);
        A.dbg[2 * pd.out_id + 1] = __builtin_amdgcn_s_memtime() - dbg_t0;
    }
    PairOut *o = &A.out[pd.out_id];
    if (COLS) {
        // combine the chunks: ONE atomicMax each (the record was zeroed by sw_sweep_winmax_kernel, one launch earlier) and
        // nothing else -- no completion count, no fence: a fence here writes back the XCD's L2, full of the checkpoints just
        // stored, and cost every chunk's launch ~30 us (profiles/r02/col_chunks.md).  The traceback kernels, one launch
        // later, read the final maximum and mark the pair degenerate when it is 0 (finish_pair).
        if (pair_max > 0) atomicMax(&o->score, pair_max);
        return;
    }
    PairOut v;                 // the cells holding the maximum are listed by the traceback kernel (n_cells follows there)
    if (pair_max <= 0) { v.score = 0; v.flags = SWMI_F_DEGENERATE; v.n_cells = (uint64_t)m * n_full; }
    else               { v.score = pair_max; v.flags = 0u; v.n_cells = 0; }
    *o = v;
}

template <bool COLS>
__device__ __forceinline__ void sweep_fast_dispatch(const FillArgs &A, const PairDesc pd, uint32_t lane, uint32_t m, const ColItem ci) {
    const uint32_t R = swmi_rows_per_lane(m);
    if (R == 1)      sweep_fast<1, COLS>(A, pd, lane, ci);
    else if (R == 2) sweep_fast<2, COLS>(A, pd, lane, ci);
    else if (R == 3) sweep_fast<3, COLS>(A, pd, lane, ci);
    else             sweep_fast<4, COLS>(A, pd, lane, ci);
}

template <bool ACGT, bool STRICT, int MODE>
__device__ __forceinline__ void fill_dispatch(const FillArgs &A, const PairDesc pd, uint32_t lane, uint32_t m) {
    const uint32_t R = swmi_rows_per_lane(m);
    if (R == 1)      fill_pair<1, ACGT, STRICT, false, MODE>(A, pd, lane);
    else if (R == 2) fill_pair<2, ACGT, STRICT, false, MODE>(A, pd, lane);
    else if (R == 3) fill_pair<3, ACGT, STRICT, false, MODE>(A, pd, lane);
    else if (m <= WAVE * SWMI_RMAX) fill_pair<SWMI_RMAX, ACGT, STRICT, false, MODE>(A, pd, lane);
    else             fill_pair<SWMI_RMAX, ACGT, STRICT, true, MODE>(A, pd, lane);
}

// 4 wavefronts per workgroup, one pair each: the 4 waves of a workgroup land on the 4 SIMDs of a CU, so a
// grid of n_pairs/4 workgroups spreads evenly over the SIMDs.
#define FILL_WAVES 4
template <int MODE>
__device__ __forceinline__ void fill_entry(const FillArgs &A) {
    const uint32_t pair = blockIdx.x * FILL_WAVES + (threadIdx.x >> 6);
    if (pair >= A.n_pairs) return;
    const uint32_t lane = threadIdx.x & 63u;
    if (pair == 0 && lane == 0 && A.hdr) { A.hdr->reserved = 0; A.hdr->pad = 0; }   // arena reset for the traceback kernel that follows
    if (pair == 0 && lane == 0 && A.q_reset) *A.q_reset = 0u;                        // ... and the split traceback's item counter
    const PairDesc pd = A.pairs[pair];
    if (MODE == SWMI_MODE_WINMAX && (pd.pad & SWMI_PAD_RESIDENT)) return;            // sw_resident_pairs_kernel does the whole pair
    const SeqDesc rd = A.refs[pd.ref_id];
    const SeqDesc qd = A.reads[pd.read_id];
    if (MODE == SWMI_MODE_WINMAX && A.skip_multi && qd.len > WAVE * SWMI_RMAX) {
        // swept strip by strip (sw_sweep_winmax_strips_kernel, next launch): start the record its strips complete by atomics
        if (lane == 0) { PairOut z; z.score = 0; z.flags = 0u; z.n_cells = 0; A.out[pd.out_id] = z; }
        return;
    }
    // profile lookup needs both sequences pure ACGT and scores that fit a signed byte
    const bool acgt = rd.acgt && qd.acgt &&
                      SWMI_SCORES_FIT(A);
    if (MODE == SWMI_MODE_WINMAX && qd.len <= WAVE * SWMI_RMAX && (pd.pad & SWMI_PAD_COLS)) {
        // swept chunk by chunk (sw_sweep_winmax_cols_kernel, next launch): start the record its chunks complete by atomics
        if (lane == 0) { PairOut z; z.score = 0; z.flags = 0u; z.n_cells = 0; A.out[pd.out_id] = z; }
        return;
    }
    if (MODE == SWMI_MODE_WINMAX && acgt && qd.len <= WAVE * SWMI_RMAX && A.gap <= 0) {
        sweep_fast_dispatch<false>(A, pd, lane, qd.len, ColItem{0u, 0u, 0u, 0u});
        return;
    }
    if (MODE == SWMI_MODE_SCORE || MODE == SWMI_MODE_WINMAX) {   // scores do not depend on the tie order
        if (acgt) fill_dispatch<true, false, MODE>(A, pd, lane, qd.len);
        else      fill_dispatch<false, false, MODE>(A, pd, lane, qd.len);
    } else if (acgt) {
        if (A.strict) fill_dispatch<true, true, MODE>(A, pd, lane, qd.len);
        else          fill_dispatch<true, false, MODE>(A, pd, lane, qd.len);
    } else {
        if (A.strict) fill_dispatch<false, true, MODE>(A, pd, lane, qd.len);
        else          fill_dispatch<false, false, MODE>(A, pd, lane, qd.len);
    }
}

extern "C" __global__ void __launch_bounds__(WAVE * FILL_WAVES)
sw_fill_kernel(const FillArgs A) { fill_entry<SWMI_MODE_FIELD>(A); }

extern "C" __global__ void __launch_bounds__(WAVE * FILL_WAVES)
sw_fill_score_kernel(const FillArgs A) { fill_entry<SWMI_MODE_SCORE>(A); }

// (at most 128 VGPRs -- 14 spills, none in the fast stream: with two batches in flight this kernel's wavefront shares its SIMD
//  with the other batch's traceback wavefronts, 128 VGPRs each: three of them fit beside it, at 137 only two.  Two in flight
//  0.124 -> 0.119 ms per step, one at a time 0.1655 -> 0.167: profiles/r03/ab_sweep_128vgpr_after_lds.txt)
extern "C" __global__ void __launch_bounds__(WAVE * FILL_WAVES) __attribute__((amdgpu_waves_per_eu(4, 4)))
sw_sweep_winmax_kernel(const FillArgs A) { fill_entry<SWMI_MODE_WINMAX>(A); }

// mode 1, reads of several strips: one wavefront per (pair, strip).  The items are ordered so that a strip's producer
// (the strip above it) sits in the same or an earlier workgroup, i.e. is never dispatched later than its consumer.
extern "C" __global__ void __launch_bounds__(WAVE * FILL_WAVES)
sw_sweep_winmax_strips_kernel(const FillArgs A) {
    const uint32_t item = blockIdx.x * FILL_WAVES + (threadIdx.x >> 6);
    if (item >= A.n_strip_items) return;
    const uint32_t lane = threadIdx.x & 63u;
    const StripItem *it = A.strip_items + item;
    const PairDesc pd = A.pairs[it->pair];
    const SeqDesc rd = A.refs[pd.ref_id];
    const SeqDesc qd = A.reads[pd.read_id];
    const bool acgt = rd.acgt && qd.acgt &&
                      SWMI_SCORES_FIT(A);
    const uint32_t strip = __builtin_amdgcn_readfirstlane(it->strip);     // wave-uniform: "does this strip feed a seam" stays scalar
    if (acgt) fill_pair<SWMI_RMAX, true, false, true, SWMI_MODE_WINMAX, true>(A, pd, lane, strip, it);
    else      fill_pair<SWMI_RMAX, false, false, true, SWMI_MODE_WINMAX, true>(A, pd, lane, strip, it);
}

// mode 1, few pairs with long references: one wavefront per COLUMN CHUNK of a pair (swmi_device.h: ColItem).  The
// chunks of a pair are independent -- each re-derives its left context from a halo no positive-score path can span --
// so a 128 kbp reference against one read is swept by dozens of wavefronts at once instead of one 128 k-step chain.
extern "C" __global__ void __launch_bounds__(WAVE * FILL_WAVES)
sw_sweep_winmax_cols_kernel(const FillArgs A) {
    const uint32_t item = blockIdx.x * FILL_WAVES + (threadIdx.x >> 6);
    if (item >= A.n_col_items) return;
    const uint32_t lane = threadIdx.x & 63u;
    const ColItem ci = A.col_items[item];
    const PairDesc pd = A.pairs[ci.pair];
    sweep_fast_dispatch<true>(A, pd, lane, A.reads[pd.read_id].len, ci);
}

// ------------------------------------------------------------------------------------------------
// traceback: SWMI_TB_SLOTS wavefronts per pair (slot x walks the tied cells x, x+SLOTS, ...).
//
// The walk is a chain of dependent 2-bit lookups; straight from HBM that is ~1 us per step, and even from
// LDS a one-cell-at-a-time scalar walk costs ~100 instruction issues per step.  So:
//  * the wave brings a TILE of the direction field -- every cell whose anti-diagonal step lies in a window
//    of 16-step blocks, all 64*R row slots of the strip -- into LDS, plus the matching window of reference
//    codes and the whole read.  Mode 0 copies it from the HBM direction field (16 blocks, coalesced 256 B
//    loads, all in flight at once); mode 1 RE-SWEEPS SWMI_CK_BLOCKS blocks from the lane-state checkpoint the
//    score-only fill left behind, this time with the direction bits (the same instruction stream as the
//    mode-0 fill), which is cheaper than having every pair pay 4 more VALU per cell in the fill;
//  * it then advances by RUNS: lane x looks at the cell x steps up the current diagonal, a ballot gives the
//    length of the run of "alignment" moves, a second ballot over per-lane prefix scores finds where the
//    tracked score H(pred) = H - s(ref,read) would reach 0 (`while (score > 0)`, SmithWaterman.java:380), and
//    the whole run is emitted at once.  Gap moves (insertion / deletion) are taken one at a time.
// ------------------------------------------------------------------------------------------------
#define SWMI_TB_BLOCKS 16u
#define SWMI_TB_REFWIN_WORDS 96u      // (16*16 + 63) / 4 + slack
#define SWMI_TB_SLOTS 4u
#if SWMI_CK_BLOCKS > 2
#define SWMI_TB_WAVES 4u            // (64-step windows: a team's span of 4 windows is what the reference-window staging holds)
#else
#define SWMI_TB_WAVES 8u            // mode 1: most waves per workgroup (= per pair) of the traceback kernel; the launcher picks 4 or 8
#endif

// The same re-sweep for the usual pair (fast symbols, one strip, gap <= 0), from the shorter instruction stream of
// tools/gen_step.py (DirStep4Asm: the 3-VALU cell + two compares feeding v_addc, neighbour exchange inside the arithmetic;
// 25 instructions per step at R = 3 instead of ~30).  No lane is masked: a lane outside its column range computes bits
// nobody reads, and because every step pushes exactly one bit pair the bits of the real steps sit where the walk expects them.
template <int R, bool STRICT>
__device__ __forceinline__ void replay_window_fast(const TraceArgs &A, const PairDesc pd, const uint32_t n, const uint32_t m,
                                                   const uint32_t *__restrict__ refw, const uint32_t *__restrict__ readw,
                                                   const uint32_t wlo, const uint32_t lane, uint32_t *__restrict__ lds_tile) {
    const uint32_t gm = (uint32_t)(-(int64_t)A.gap);
    const int one = 1;
    int h[R], g[R], hp[R], q[R];
    uint32_t acc[R];
    const uint32_t *__restrict__ ck = A.dir + pd.dir_off + (uint64_t)(wlo / SWMI_CK_BLOCKS) * (R + 2) * WAVE + lane;
    build_profiles<R>(q, readw, lane * R, m, A.match, A.mismatch);
#pragma unroll
    for (int k = 0; k < R; ++k) {
        h[k] = (int)ld_l2(ck + k * WAVE);
        hp[k] = (uint32_t)h[k] > gm ? (int)((uint32_t)h[k] - gm) : 0;
        g[k] = 0;
        acc[k] = 0;
    }
    // the checkpoint holds, per lane, the N it received one step earlier (= NW of its row 0 now); the stream takes that value
    // from the lane above, out of its ping-pong set: hand it back up (wave_shl:1)
    const int nprev = (int)ld_l2(ck + R * WAVE);
    g[R - 1] = __builtin_amdgcn_update_dpp(0, nprev, 0x130 /*wave_shl:1*/, 0xf, 0xf, true);
    const int rb_ck = (int)ld_l2(ck + (R + 1) * WAVE);
    const uint32_t lact = m >= WAVE * R ? WAVE : (m + R - 1) / R;
    const uint32_t T = n + lact - 1, nblk = (T + 15u) / 16u;
    const uint4 *__restrict__ refq = reinterpret_cast<const uint4 *>(refw);
    uint4 wv[SWMI_CK_BLOCKS + 1];
#pragma unroll
    for (uint32_t b = 0; b <= SWMI_CK_BLOCKS; ++b) wv[b] = refq[wlo + b];          // images are padded past the last block
    int rbx = wave_shr1((int)(1u << (wv[0].x & 31u)), rb_ck), rby = rb_ck;
#pragma unroll
    for (uint32_t b = 0; b < SWMI_CK_BLOCKS; ++b) {
        if (wlo + b >= nblk) break;
        const uint4 w = wv[b];
        DirStep4Asm<R, STRICT>::run(h, g, hp, acc, q, rbx, rby, w.x, w.y, one, gm);
        DirStep4Asm<R, STRICT>::run(h, g, hp, acc, q, rbx, rby, w.y, w.z, one, gm);
        DirStep4Asm<R, STRICT>::run(h, g, hp, acc, q, rbx, rby, w.z, w.w, one, gm);
        DirStep4Asm<R, STRICT>::run(h, g, hp, acc, q, rbx, rby, w.w, wv[b + 1].x, one, gm);
#pragma unroll
        for (int k = 0; k < R; ++k) lds_tile[(b * R + k) * WAVE + lane] = acc[k];
    }
}

// re-sweep the window of SWMI_CK_BLOCKS blocks that starts at block `wlo` of strip `s` into lds_tile.
// DETECT: also append the window's cells equal to `maxv` to the pair's cell list; returns the new list length.
template <int R, bool ACGT, bool STRICT, bool MULTI, bool DETECT>
__device__ __forceinline__ uint32_t replay_window(const TraceArgs &A, const PairDesc pd, const uint32_t n, const uint32_t m,
                                                  const uint32_t *__restrict__ refw, const uint32_t *__restrict__ readw,
                                                  const StripGeom G, const uint32_t s, const uint32_t wlo,
                                                  const uint32_t lane, uint32_t *__restrict__ lds_tile,
                                                  const int maxv, const uint32_t cnt_in, uint2 *__restrict__ cells, const uint32_t ccap) {
#ifndef SWMI_NO_ASM
    if constexpr (ACGT && !MULTI && !DETECT) {
        if (A.gap <= 0 && s == 0u) {
            replay_window_fast<R, STRICT>(A, pd, n, m, refw, readw, wlo, lane, lds_tile);
            return cnt_in;
        }
    }
#endif
    constexpr int BM = DETECT ? SWMI_MODE_DETECT : SWMI_MODE_REPLAY;
    FillState<R> S;
    S.thr = DETECT ? maxv : 0x7FFFFFFF; S.cnt = cnt_in; S.ev_prev = 0; S.events = 0; S.dbg_skip = false; S.lmax = -1;
    const uint32_t row0 = s * G.rps + lane * R;
    const uint32_t rows_left = m - s * G.rps;
    const uint32_t lact = rows_left >= G.rps ? WAVE : (rows_left + R - 1) / R;
    const uint32_t T = n + lact - 1;
    const uint32_t lane_eff = lane < lact ? lane : 0x40000000u;
    setup_rows<R, ACGT>(S, readw, row0, m, A.match, A.mismatch);
    const uint32_t *__restrict__ ck = A.dir + pd.dir_off + s * G.strip_words +
                                      (uint64_t)(wlo / SWMI_CK_BLOCKS) * (R + 2) * WAVE + lane;
#pragma unroll
    for (int k = 0; k < R; ++k) S.h[k] = (int)ld_l2(ck + k * WAVE);
    S.nprev = (int)ld_l2(ck + R * WAVE);
    S.rb = (int)ld_l2(ck + (R + 1) * WAVE);
    const int32_t *seam_in = nullptr;
    if (MULTI) seam_in = A.seam + pd.seam_off + (uint64_t)(s > 0 ? s - 1 : 0) * (n + 1);
    const bool reads_seam = MULTI && (s > 0);
    const uint32_t nblk = (T + 15u) / 16u;
    const uint4 *__restrict__ refq = reinterpret_cast<const uint4 *>(refw);
    uint4 wv[SWMI_CK_BLOCKS];                                  // the window's base codes: all loads in flight at once
#pragma unroll
    for (uint32_t b = 0; b < SWMI_CK_BLOCKS; ++b) wv[b] = refq[wlo + b];     // images are padded past nblk
#pragma unroll
    for (uint32_t b = 0; b < SWMI_CK_BLOCKS; ++b) {
        const uint32_t tb = wlo + b;
        if (tb >= nblk) break;
        const uint4 w = wv[b];
        const uint32_t t0 = 16u * tb;
        int seamv = 0;
        if (reads_seam) {
            const uint32_t col = t0 + 1u + (lane & 15u);
            seamv = col <= n ? (int)ld_l2((const uint32_t *)seam_in + col) : 0;
        }
        const bool steady = (t0 + 1u >= lact) && (t0 + 15u < n);
        if (steady)
            fill_block16<R, ACGT, STRICT, MULTI, false, BM>(S, w, t0, lane, lane_eff, n, m, row0, A.gap, A.match, A.mismatch,
                                                            seamv, reads_seam, false, nullptr, cells, ccap);
        else
            fill_block16<R, ACGT, STRICT, MULTI, true, BM>(S, w, t0, lane, lane_eff, n, m, row0, A.gap, A.match, A.mismatch,
                                                           seamv, reads_seam, false, nullptr, cells, ccap);
        const int miss = (int)(t0 + 15u) - ((int)(lane + n) - 1);
#pragma unroll
        for (int k = 0; k < R; ++k) {
            uint32_t v = S.acc[k];
            if (miss > 0 && miss < 16) v <<= 2 * miss;
            lds_tile[(b * R + k) * WAVE + lane] = v;
        }
    }
    return S.cnt;
}

template <int R, bool MULTI, bool DETECT>
__device__ __forceinline__ uint32_t replay_dispatch(const TraceArgs &A, const PairDesc pd, uint32_t n, uint32_t m, bool acgt,
                                                    const uint32_t *__restrict__ refw, const uint32_t *__restrict__ readw,
                                                    const StripGeom G, uint32_t s, uint32_t wlo, uint32_t lane, uint32_t *lds_tile,
                                                    int maxv, uint32_t cnt_in, uint2 *__restrict__ cells, uint32_t ccap) {
    if (acgt) {
        if (A.strict) return replay_window<R, true, true, MULTI, DETECT>(A, pd, n, m, refw, readw, G, s, wlo, lane, lds_tile, maxv, cnt_in, cells, ccap);
        else          return replay_window<R, true, false, MULTI, DETECT>(A, pd, n, m, refw, readw, G, s, wlo, lane, lds_tile, maxv, cnt_in, cells, ccap);
    } else {
        if (A.strict) return replay_window<R, false, true, MULTI, DETECT>(A, pd, n, m, refw, readw, G, s, wlo, lane, lds_tile, maxv, cnt_in, cells, ccap);
        else          return replay_window<R, false, false, MULTI, DETECT>(A, pd, n, m, refw, readw, G, s, wlo, lane, lds_tile, maxv, cnt_in, cells, ccap);
    }
}

// A pair swept by several wavefronts (column chunks, strips) reaches the mode-1 traceback with only its maximum combined:
// a maximum of 0 is the degenerate case -- every one of the m*n cells ties (SmithWaterman.java:154,182-185).  (A pair swept
// by one wavefront was marked by it.)  Returns true if `po` was completed here.
__device__ __forceinline__ bool finish_pair(const TraceArgs &A, const PairDesc pd, PairOut &po) {
    if (po.score > 0 || (po.flags & SWMI_F_DEGENERATE)) return false;
    po.score = 0;
    po.flags = SWMI_F_DEGENERATE;
    po.n_cells = (uint64_t)A.reads[pd.read_id].len * A.refs[pd.ref_id].len;
    return true;
}

// true for the pairs the full-featured mode-1 traceback handles: pure ACGT with int8 scores, the serial tie order
__device__ __forceinline__ bool swmi_common_pair(const TraceArgs &A, const SeqDesc rd, const SeqDesc qd) {
    return rd.acgt && qd.acgt && SWMI_SCORES_FIT(A) && !A.strict;
}

// re-sweeps one window.  FULL: any variant (byte alphabet, DistributedSW tie order, several strips); otherwise only the
// common one -- the call sites of the team code are many, and every variant inlined at each of them is what made this
// file take six minutes to compile.
template <int R, bool DETECT, bool FULL = true>
__device__ __forceinline__ uint32_t replay_any(const TraceArgs &A, const PairDesc pd, uint32_t n, uint32_t m, bool acgt,
                                               const uint32_t *__restrict__ refw, const uint32_t *__restrict__ readw,
                                               const StripGeom G, uint32_t s, uint32_t wlo, uint32_t lane, uint32_t *lds_tile,
                                               int maxv, uint32_t cnt_in, uint2 *__restrict__ cells, uint32_t ccap) {
    if constexpr (!FULL) {
        if constexpr (R == SWMI_RMAX) {
            if (m > WAVE * SWMI_RMAX)          // a read of several strips
                return replay_window<R, true, false, true, DETECT>(A, pd, n, m, refw, readw, G, s, wlo, lane, lds_tile, maxv, cnt_in, cells, ccap);
        }
        return replay_window<R, true, false, false, DETECT>(A, pd, n, m, refw, readw, G, s, wlo, lane, lds_tile, maxv, cnt_in, cells, ccap);
    }
    if constexpr (R == SWMI_RMAX) {
        if (m > WAVE * SWMI_RMAX)
            return replay_dispatch<R, true, DETECT>(A, pd, n, m, acgt, refw, readw, G, s, wlo, lane, lds_tile, maxv, cnt_in, cells, ccap);
    }
    return replay_dispatch<R, false, DETECT>(A, pd, n, m, acgt, refw, readw, G, s, wlo, lane, lds_tile, maxv, cnt_in, cells, ccap);
}

// mode 1: list the pair's maximum cells.  The sweep left one maximum per checkpoint window; every window whose
// maximum equals the pair's is re-swept once with the cell test switched on.
template <int R, bool FULL = true>
__device__ __forceinline__ uint32_t detect_cells(const TraceArgs &A, const PairDesc pd, const PairOut po,
                                                 const uint32_t lane, uint32_t *__restrict__ lds_tile) {
    const SeqDesc rd = A.refs[pd.ref_id];
    const SeqDesc qd = A.reads[pd.read_id];
    const uint32_t n = rd.len, m = qd.len;
    const uint32_t *__restrict__ refw = A.seqw + rd.boff;
    const uint32_t *__restrict__ readw = A.seqw + qd.boff;
    const StripGeom G = strip_geom<R>(m, n, 1u);
    const bool acgt = rd.acgt && qd.acgt && SWMI_SCORES_FIT(A);
    const uint64_t cbase = A.cells_off ? A.cells_off[pd.out_id] : (uint64_t)pd.out_id * A.cell_cap;
    const uint32_t ccap = A.cells_cap ? A.cells_cap[pd.out_id] : A.cell_cap;
    uint2 *__restrict__ cells = const_cast<uint2 *>(A.cells) + cbase;
    uint32_t cnt = 0;
    for (uint32_t s = 0; s < G.n_strips; ++s) {
        const uint32_t *__restrict__ wm = A.dir + pd.dir_off + s * G.strip_words + G.wmax_off;
        for (uint32_t g0 = 0; g0 < G.n_ck; g0 += WAVE) {
            const uint32_t g = g0 + lane;
            const int wv = g < G.n_ck ? (int)wm[g] : -1;
            uint64_t cand = BALLOT(wv == po.score);
            while (cand) {
                const uint32_t gg = g0 + (uint32_t)__builtin_ctzll(cand);
                cand &= cand - 1ull;
                cnt = replay_any<R, true, FULL>(A, pd, n, m, acgt, refw, readw, G, s, gg * SWMI_CK_BLOCKS, lane, lds_tile,
                                          po.score, cnt, cells, ccap);
            }
        }
    }
    return cnt;
}

// COOP (mode 1): the workgroup's waves form `nslots` teams of `ts` waves, one walker (this function) plus ts-1
// helpers (coop_helper below) each.  Instead of re-sweeping one 32-step window at a time, the walker publishes a
// request for up to ts consecutive windows, the team re-sweeps them in parallel into the team's tile, and the walker
// then crosses the whole 32*ts-step span without stopping.  The teams of a workgroup run independently of each other:
// a request is a sequence number in LDS the helpers poll (s_sleep between polls), completion a counter the walker polls;
// there is no workgroup barrier after the cell list.
// shared[]: [0] number of maximum cells, [4+4t ..] team t's request {strip, first block, windows, sequence number
// (~0: the walker is done)}, [20+t] windows team t's helpers have delivered so far.
// Per-pair diagnostics of the traceback (ticks of the walk and of the stagings, iterations, steps: SWMI_DEBUG_FILL=1) are compiled
// in only with -DSWMI_TB_DIAG (make KFLAGS=-DSWMI_TB_DIAG): their counters lived in registers across the whole walk of a kernel
// that sits at its 128-VGPR limit with spills.
#ifdef SWMI_TB_DIAG
#define TB_DBG (A.dbg != nullptr)
#else
#define TB_DBG false
#endif
template <int R, int TMODE, bool COOP, bool FULL = true>
__device__ __forceinline__ void traceback_pair(const TraceArgs &A, const PairDesc pd, const PairOut po,
                                               const uint32_t lane, const uint32_t slot, const uint32_t nslots,
                                               uint32_t *__restrict__ lds, uint32_t *__restrict__ lds_tile,
                                               volatile uint32_t *__restrict__ shared, const uint32_t ts,
                                               uint32_t pre_wlo = 0xFFFFFFFFu, const bool read_staged = false,
                                               const uint4 *__restrict__ one = nullptr, uint32_t *__restrict__ lds_tile_alt = nullptr) {
    const SeqDesc rd = A.refs[pd.ref_id];
    const SeqDesc qd = A.reads[pd.read_id];
    const uint32_t n = rd.len, m = qd.len;
    const uint32_t *__restrict__ refw = A.seqw + rd.boff;
    const uint32_t *__restrict__ readw = A.seqw + qd.boff;
    const StripGeom G = strip_geom<R>(m, n, A.mode);
    const uint32_t rps = G.rps;
    const uint32_t *__restrict__ dirp = A.dir + pd.dir_off;
    const uint32_t umat = (uint32_t)A.match, umis = (uint32_t)A.mismatch, ugap = (uint32_t)A.gap;
    const int dec0 = A.match > A.mismatch ? A.match : A.mismatch;
    const uint32_t udec = dec0 > 0 ? (uint32_t)dec0 : 0u;    // the most one alignment move can lower the tracked score
    const uint32_t ugdec = A.gap > 0 ? (uint32_t)A.gap : 0u;  // ... and one gap move
    (void)udec; (void)ugdec;                                  // (only the SWMI_WALK_DIAGONALS build of the walk uses them)
    const bool acgt = rd.acgt && qd.acgt && SWMI_SCORES_FIT(A);
    // the caller's own bytes of the two sequences (for the aligned strings; loaded here, long before they are needed)
    const uint8_t *__restrict__ raw_ref = A.raw ? A.raw + A.raw_off[pd.ref_id] : nullptr;
    const uint8_t *__restrict__ raw_read = A.raw ? A.raw + A.raw_off[A.raw_reads_at + pd.read_id] : nullptr;

    // (s_setprio for the walker over the helpers sharing its SIMD: measured, no effect)
    uint32_t *lds_ops = lds;                                   // [A.lds_words]      one op per BYTE, staged per alignment
    uint32_t *lds_read = lds_ops + A.lds_words;                // [A.lds_read_words] the read's codes
    uint32_t *lds_ref = lds_read + A.lds_read_words;           // [SWMI_TB_REFWIN_WORDS]
    uint8_t *ops_b = reinterpret_cast<uint8_t *>(lds_ops);
    const uint8_t *read_b = reinterpret_cast<const uint8_t *>(lds_read);
    const uint8_t *ref_b = reinterpret_cast<const uint8_t *>(lds_ref);
    uint32_t req_seq = 0, req_expected = 0;                    // COOP: requests published / windows expected back so far
    // COOP with a second tile buffer (lds_tile_alt): SPECULATIVE staging.  A path only ever moves towards smaller steps, so
    // the span the walk will need next is the one just before the current one: the team's helpers re-sweep it into the other
    // buffer WHILE the walker walks, and the walker finds it ready when it crosses the boundary (it swaps buffers instead of
    // waiting ~7 k cycles for a re-sweep).  A span whose walk ends early costs the helpers -- otherwise idle -- one re-sweep.
    uint32_t *tile_cur = lds_tile, *tile_alt = lds_tile_alt;
    uint32_t spec_wlo = 0xFFFFFFFFu, spec_hi = 0u, spec_rv0 = 0u, spec_rv1 = 0u;
    auto team_wait = [&]() {                                   // every window requested so far has been delivered
        while (__hip_atomic_load(const_cast<uint32_t *>(&shared[20u + slot]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < req_expected)
            __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    };
    // request {flags: bit 0 = buffer (0: the team's first, 1: its second), bit 1 = the walker takes no window of it, bits 2.. =
    // the strip; first block; windows}; the helpers poll the sequence number (coop_helper)
    auto team_request = [&](uint32_t flags, uint32_t first_block, uint32_t nq, uint32_t delivered) {
        ++req_seq;
        req_expected += delivered;
        if (lane == 0) {
            shared[4u + 4u * slot] = flags; shared[5u + 4u * slot] = first_block; shared[6u + 4u * slot] = nq;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __hip_atomic_store(const_cast<uint32_t *>(&shared[7u + 4u * slot]), req_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    };

    const unsigned long long tk0 = TB_DBG ? __builtin_amdgcn_s_memtime() : 0ull;
    unsigned long long tk_walk = 0, tk_stage = 0, n_steps = 0, n_iters = 0;
    if (!read_staged)
        for (uint32_t w = lane; w < (m + 3u) / 4u; w += WAVE) lds_read[w] = readw[w];

    const uint64_t cbase = A.cells_off ? A.cells_off[pd.out_id] : (uint64_t)pd.out_id * A.cell_cap;
    const uint32_t ncell = one ? 1u : (uint32_t)po.n_cells;          // `one`: walk just this cell (split traceback)
    const uint2 *__restrict__ cells = A.cells + cbase;

    // The tied cells are processed in list order; `rank` is the position the reference would list the cell
    // at: row-major (SmithWaterman.java:157-185) or per anti-diagonal, ascending j (DistributedSW.java:209-239).
    for (uint32_t base = 0; base < ncell; base += WAVE) {
        const uint32_t idx = base + lane;
        uint2 mine = make_uint2(0, 0);
        if (one) { mine.x = one->y; mine.y = one->z; }
        else if (idx < ncell) { mine.x = ld_l2(&cells[idx].x); mine.y = ld_l2(&cells[idx].y); }
        const uint64_t mykey = A.strict ? (((uint64_t)(mine.x + mine.y) << 32) | mine.y)
                                        : (((uint64_t)mine.x << 32) | mine.y);
        // (a pair with more tied cells than one wave of lanes: every lane comparing its cell with all others is O(cells^2)
        //  global loads per walker -- 1600 cells on a 128 kbp periodic reference -- so the host orders those records by cell)
        uint32_t rank = (one || ncell > WAVE) ? SWMI_RANK_BY_CELL : 0u;
        if (ncell > 1 && ncell <= WAVE) {
            for (uint32_t o = 0; o < ncell; ++o) {
                uint2 c; c.x = ld_l2(&cells[o].x); c.y = ld_l2(&cells[o].y);
                const uint64_t kk = A.strict ? (((uint64_t)(c.x + c.y) << 32) | c.y) : (((uint64_t)c.x << 32) | c.y);
                rank += (kk < mykey) ? 1u : 0u;
            }
        }
        const uint32_t nhere = ncell - base < WAVE ? ncell - base : WAVE;

        for (uint32_t a = slot; a < nhere; a += nslots) {
            const uint32_t ci = __builtin_amdgcn_readlane((int)mine.x, a);
            const uint32_t cj = __builtin_amdgcn_readlane((int)mine.y, a);
            const uint32_t crank = __builtin_amdgcn_readlane((int)rank, a);

            // ---- walk, SmithWaterman.java:380-409; (i, j, score, n_ops, begin) are wave-uniform ----
            uint32_t i = ci, j = cj;
            uint32_t score = (uint32_t)po.score;
            uint32_t n_ops = 0;
            int begin = 0;
            while ((int)score > 0 && i != 0u && j != 0u) {     // (a positive score at row/column 0 cannot happen with consistent
                const unsigned long long ts0 = TB_DBG ? __builtin_amdgcn_s_memtime() : 0ull;
                                                               //  data; the test keeps a corrupted workspace from walking off the matrix)
                // ---- stage the window that holds the current cell's step ----
                const uint32_t s = (i - 1u) / rps;
                int rho = (int)((i - 1u) - s * rps);                               // row slot within the strip
                const uint32_t t_cur = j - 1u + (uint32_t)rho / R;
                const uint32_t whi = t_cur >> 4;
                uint32_t wlo, nb;
                if (TMODE == 0) {
                    wlo = whi >= SWMI_TB_BLOCKS - 1u ? whi - (SWMI_TB_BLOCKS - 1u) : 0u;
                    nb = whi - wlo + 1u;
                } else if (COOP) {
                    const uint32_t wtop = whi - whi % SWMI_CK_BLOCKS;
                    wlo = wtop >= (ts - 1u) * SWMI_CK_BLOCKS ? wtop - (ts - 1u) * SWMI_CK_BLOCKS : 0u;
                    nb = wtop + SWMI_CK_BLOCKS - wlo;
                } else {
                    wlo = whi - whi % SWMI_CK_BLOCKS;                              // windows start at checkpoints
                    nb = SWMI_CK_BLOCKS;
                }
                // COOP: the span the helpers have been re-sweeping since the last staging, if the walk arrived there
                bool use_spec = false;
                if (COOP && spec_wlo != 0xFFFFFFFFu) {
                    const uint32_t wtop = whi - whi % SWMI_CK_BLOCKS;
                    if (s == 0u && wtop + SWMI_CK_BLOCKS == spec_hi) { use_spec = true; wlo = spec_wlo; nb = spec_hi - spec_wlo; }
                }
                // COOP: the span of the first staging was already re-swept while wave 0 listed the maximum cells
                const bool prestaged = TMODE == 1 && pre_wlo == wlo && s == 0u && !use_spec;
                pre_wlo = 0xFFFFFFFFu;
                const int clo = (int)(16u * wlo) - 63;
                const uint32_t cw0 = clo > 0 ? (uint32_t)clo >> 2 : 0u;           // first dword of the reference window
                const uint32_t cw1 = (16u * (wlo + nb) - 1u) >> 2;
                WAVE_SYNC();
                if (use_spec) {
                    const unsigned long long tq0 = TB_DBG ? __builtin_amdgcn_s_memtime() : 0ull;
                    team_wait();
                    if (TB_DBG) tk_walk += (__builtin_amdgcn_s_memtime() - tq0) << 32;
                    uint32_t *t = tile_cur; tile_cur = tile_alt; tile_alt = t;
                    lds_ref[lane] = spec_rv0;
                    if (lane + WAVE < SWMI_TB_REFWIN_WORDS) lds_ref[lane + WAVE] = spec_rv1;
                    WAVE_SYNC();
                } else if (!prestaged) {
                    // the slice of the reference: loads issued now, stored to LDS after the re-sweep (at most 2 dwords per lane)
                    const uint32_t rx0 = cw0 + lane, rx1 = rx0 + WAVE, rlim = (n + 3u) / 4u;
                    const uint32_t rv0 = (rx0 <= cw1 && rx0 < rlim) ? refw[rx0] : 0u;
                    const uint32_t rv1 = (rx1 <= cw1 && rx1 < rlim) ? refw[rx1] : 0u;
                    if (TMODE == 0) {
                        const uint32_t *__restrict__ src = dirp + s * G.strip_words + (uint64_t)wlo * R * WAVE + lane;
                        for (uint32_t x = 0; x < nb * R; ++x) tile_cur[x * WAVE + lane] = src[(uint64_t)x * WAVE];
                    } else {
                        if (COOP) {
                            team_wait();                                           // (a speculative request nobody needs any more)
                            const uint32_t nq = nb / SWMI_CK_BLOCKS;
                            team_request((tile_cur == lds_tile ? 0u : 1u) | (s << 2), wlo, nq, nq - 1u);     // helpers 1 .. nq-1 deliver one window each
                        }
                        (void)replay_any<R, false, FULL>(A, pd, n, m, acgt, refw, readw, G, s, wlo, lane, tile_cur, 0, 0u, nullptr, 0u);
                    }
                    lds_ref[lane] = rv0;
                    if (lane + WAVE < SWMI_TB_REFWIN_WORDS) lds_ref[lane + WAVE] = rv1;
                    if (COOP) {
                        // wait for the helpers' windows
                        const unsigned long long tq0 = TB_DBG ? __builtin_amdgcn_s_memtime() : 0ull;
                        team_wait();
                        if (TB_DBG) tk_walk += (__builtin_amdgcn_s_memtime() - tq0) << 32;     // (waiting counted in the upper half)
                    }
                    WAVE_SYNC();
                }
                spec_wlo = 0xFFFFFFFFu;
                if (COOP && tile_alt != nullptr && ts > 1u && wlo > 0u && s == 0u) {
                    // the helpers start on the span to the left of this one, ts - 1 windows of it, while the walker walks
                    const uint32_t have = wlo / SWMI_CK_BLOCKS, nq = have < ts - 1u ? have : ts - 1u;
                    spec_hi = wlo;
                    spec_wlo = wlo - nq * SWMI_CK_BLOCKS;
                    const int sclo = (int)(16u * spec_wlo) - 63;
                    const uint32_t scw0 = sclo > 0 ? (uint32_t)sclo >> 2 : 0u, scw1 = (16u * spec_hi - 1u) >> 2;
                    const uint32_t rx0 = scw0 + lane, rx1 = rx0 + WAVE, rlim = (n + 3u) / 4u;
                    spec_rv0 = (rx0 <= scw1 && rx0 < rlim) ? refw[rx0] : 0u;      // its slice of the reference: held in registers meanwhile
                    spec_rv1 = (rx1 <= scw1 && rx1 < rlim) ? refw[rx1] : 0u;
                    team_request((tile_alt == lds_tile ? 0u : 1u) | 2u, spec_wlo, nq, nq);
                }
                const int tmin = (int)(16u * wlo);
                const unsigned long long tw0 = TB_DBG ? __builtin_amdgcn_s_memtime() : 0ull;
                if (TB_DBG) { tk_stage += tw0 - ts0; n_iters += 1ull << 32; }      // (stagings counted in the upper half)

#ifdef SWMI_WALK_CHASE
                // (Alternative walk, -DSWMI_WALK_CHASE: measured 0.084 ms against the 0.072 ms of the run-based walk below at the
                //  headline config -- one wave issues an instruction of ANY kind every ~5.6 cycles, so 23 scalar instructions per
                //  path step cost more than 145 instructions per 4.4 steps.  Kept as the simplest correct statement of the walk.)
                // One LDS round trip brings the 8 x 8 NEIGHBOURHOOD up-left of the current cell into the
                // wave: lane (a, b) = (lane >> 3, lane & 7) looks at cell (i - a, j - b) -- its direction bits, whether its two
                // bases match, whether it is staged at all -- and packs that into one word.  The path is then chased through the
                // neighbourhood by SCALAR code, one v_readlane per step (the lane index is the position in the neighbourhood):
                // direction, H(pred) = H - delta (`while (score > 0)`, SmithWaterman.java:380-409), op, next cell -- about 15
                // scalar instructions per path step and no memory access, 7-15 steps per round trip whatever mix of gaps and
                // alignment moves the path is made of.
                for (;;) {
                    if (TB_DBG) ++n_iters;
                    const uint32_t na = lane >> 3, nb = lane & 7u;
                    const int rho_x = rho - (int)na;
                    const uint32_t rx = rho_x > 0 ? (uint32_t)rho_x : 0u;
                    const uint32_t lx = rx / R, kx = rx - lx * R;
                    const uint32_t jj = j - nb;                                    // column of this lane's cell
                    const int tx = (int)jj - 1 + (int)lx;
                    const bool valid = rho_x >= 0 && j > nb && tx >= tmin;         // same strip, inside the matrix, inside the staged span
                    // all three LDS reads are issued together (one latency); lanes without a cell read element 0
                    const uint32_t dw = tile_cur[valid ? (((uint32_t)tx >> 4) - wlo) * (R * WAVE) + kx * WAVE + lx : 0u];
                    const uint32_t rc = ref_b[valid ? (jj - 1u) - 4u * cw0 : 0u];
                    const uint32_t qc = read_b[valid ? i - 1u - na : 0u];
                    const uint32_t d = (dw >> (2u * (15u - ((uint32_t)tx & 15u)))) & 3u;
                    // everything a path step needs from this cell, worked out by its lane: the move (rows, columns), the op, the
                    // score it takes off (:388-406), whether the move leaves the neighbourhood or reaches row / column 0
                    const uint32_t b0 = d & 1u, b1 = (d >> 1) & 1u;                // alignment chosen; else insertion over deletion
                    const uint32_t di = b0 | b1, dj = b0 | (b1 ^ 1u);
                    const uint32_t op = (b0 << 1) | (b1 & (b0 ^ 1u));              // SWMI_DIR_A = 2, _I = 1, _D = 0
                    const int delta = (int)(b0 ? (rc == qc ? umat : umis) : ugap);
                    const uint32_t last = (na + di > 7u || nb + dj > 7u) ? 1u : 0u;
                    const uint32_t edge = (i - na == di || j - nb == dj) ? 1u : 0u;
                    const int ctrl = valid ? (int)(0x80000000u | (edge << 9) | (last << 8) | (op << 4) | (di << 3) | dj) : 0;
                    // (the compiler does not know that i, j and score are the same in every lane: readfirstlane says so, and the
                    //  chase below then compiles to scalar code with scalar branches instead of an exec-masked vector loop)
                    const uint32_t si = uni(i), sj = uni(j);
                    uint32_t sscore = uni(score);
                    uint32_t idx = 0, sh = 0, opsacc = 0, lastc = 0, lastidx = 0;
                    bool done = false;
                    for (;;) {
                        const uint32_t c = (uint32_t)__builtin_amdgcn_readlane(ctrl, (int)idx);
                        if ((int)c >= 0) break;                                     // not staged (or in the strip above): restage from here
                        sscore -= (uint32_t)__builtin_amdgcn_readlane(delta, (int)idx);
                        opsacc |= ((c >> 4) & 3u) << sh;
                        sh += 2u;
                        lastc = c; lastidx = idx;
                        idx += c & 15u;
                        if ((int)sscore <= 0) { done = true; break; }
                        if (c & 0x300u) {                                           // the move left the neighbourhood (at most 15 ops: one word) ...
                            if (c & 0x200u) { sscore = 0; done = true; }            // ... or reached row / column 0 (cannot happen with a positive score on consistent data)
                            break;
                        }
                    }
                    const uint32_t cnt = sh >> 1;
                    const uint32_t ca = cnt ? (lastidx >> 3) + ((lastc >> 3) & 1u) : 0u, cb = cnt ? (lastidx & 7u) + (lastc & 7u) : 0u;
                    score = sscore;
                    if (cnt) begin = (int)(sj - (lastidx & 7u));                    // `beginning = j` of the last cell visited (:383)
                    if (lane < cnt && n_ops + lane < 4u * A.lds_words) ops_b[n_ops + lane] = (uint8_t)((opsacc >> (2u * lane)) & 3u);
                    n_ops += cnt; i = si - ca; j = sj - cb; rho -= (int)ca;
                    if (done || cnt == 0u || rho < 0) break;                        // finished / the current cell needs another window or strip
                }
#else
                // Three diagonals are inspected at once, 21 lanes each: group 0 runs up from the current cell, group 1
                // from the cell above it (where an insertion leads), group 2 from the cell to its left (a deletion).
                // One iteration then takes: [a gap move] + [the run of alignment moves that follows] + [the gap move
                // that ends the run] -- about 4-5 path steps per LDS round trip on gappy paths, 21+ on clean ones.
                for (;;) {
                    if (TB_DBG) ++n_iters;
                    const uint32_t grp = lane / 21u, x = lane - grp * 21u;        // lane 63: grp 3, idle
                    const uint32_t di = grp == 1u ? 1u : 0u, dj = grp == 2u ? 1u : 0u;
                    const int rho_x = rho - (int)di - (int)x;
                    const uint32_t rx = rho_x > 0 ? (uint32_t)rho_x : 0u;
                    const uint32_t lx = rx / R, kx = rx - lx * R;
                    const uint32_t jj = j - dj - x;                                // column of this lane's cell
                    const int tx = (int)jj - 1 + (int)lx;
                    const bool valid = grp < 3u && rho_x >= 0 && j > dj + x && tx >= tmin;
                    // all three LDS reads are issued together (one latency): direction word, reference code, read code.
                    // Lanes without a cell read element 0 instead of branching around the loads.
                    uint32_t dw = tile_cur[valid ? (((uint32_t)tx >> 4) - wlo) * (R * WAVE) + kx * WAVE + lx : 0u];
                    uint32_t rc = ref_b[valid ? (jj - 1u) - 4u * cw0 : 0u];
                    uint32_t qc = read_b[valid ? i - 1u - di - x : 0u];
                    // (all three in flight before anything waits: left alone, the compiler put the direction word's read
                    //  behind the wait for the two codes -- a second LDS round trip in every iteration of the walk)
                    asm volatile("" : "+v"(dw), "+v"(rc), "+v"(qc));
                    const uint32_t d = (dw >> (2u * (15u - ((uint32_t)tx & 15u)))) & 3u;
                    const bool mt = rc == qc;
                    const uint64_t vmask = BALLOT(valid);
                    if (!(vmask & 1ull)) break;                                    // current cell left the window / the strip: restage
                    const uint64_t amask = BALLOT((d & 1u) != 0u) & vmask;       // alignment chosen
                    const uint64_t imask = BALLOT(d == 2u) & vmask;              // insertion chosen (else deletion)
                    const uint64_t mm = BALLOT(mt);
                    {
                        // ---- fast path: [gap move] + [run of alignment moves] + [gap move], taken in one go when the tracked
                        // score provably stays positive throughout (no move lowers it by more than udec / ugdec) and the cell
                        // after the first gap move was inspected.  A positive score also rules out reaching row/column 0.
                        const uint32_t g1 = (uint32_t)(~amask & 1ull);
                        const uint32_t isI1 = (uint32_t)(imask & 1ull);
                        const uint32_t fb = g1 ? (isI1 ? 21u : 42u) : 0u;
                        const uint32_t vb = (uint32_t)(vmask >> fb) & 0x1FFFFFu;
                        const uint32_t frun = (uint32_t)__builtin_ctz(~((uint32_t)(amask >> fb) & 0x1FFFFFu));   // 0..21
                        const uint32_t has3 = (vb >> frun) & 1u;                   // bit 21 is never set
                        if ((!g1 || (vb & 1u)) && (int)score > (int)((g1 + has3) * ugdec + frun * udec)) {
                            const uint32_t isI3 = (uint32_t)(imask >> (fb + frun)) & 1u & has3;
                            const uint32_t nm = (uint32_t)__builtin_popcountll(mm & ((((1ull << frun) - 1ull)) << fb));
                            const uint32_t i1 = g1 & isI1, total = g1 + frun + has3;
                            score -= (g1 + has3) * ugap + nm * umat + (frun - nm) * umis;
                            const uint32_t j1 = j - (g1 - i1);                     // column after the first gap move
                            begin = has3 ? (int)(j1 - frun) : (frun ? (int)(j1 - frun + 1u) : (int)j);
                            if (lane < total && n_ops + lane < 4u * A.lds_words)
                                ops_b[n_ops + lane] = (uint8_t)(lane < g1 ? (isI1 ? SWMI_DIR_I : SWMI_DIR_D)
                                                               : lane < g1 + frun ? SWMI_DIR_A : (isI3 ? SWMI_DIR_I : SWMI_DIR_D));
                            n_ops += total;
                            const uint32_t dec_i = i1 + frun + isI3;
                            i -= dec_i; rho -= (int)dec_i;
                            j = j1 - frun - (has3 - isI3);
                            continue;
                        }
                    }
                    bool done = false;
                    uint32_t base = 0;                                             // first lane of the diagonal the run is on
                    if (!(amask & 1ull)) {
                        // ---- the current cell is an insertion or a deletion: H(pred) = H - gap   (:395-406) ----
                        begin = (int)j;                                             // :383
                        score -= ugap;
                        uint32_t op;
                        if (imask & 1ull) { --i; --rho; op = SWMI_DIR_I; base = 21u; } else { --j; op = SWMI_DIR_D; base = 42u; }
                        if (lane == 0 && n_ops < 4u * A.lds_words) ops_b[n_ops] = (uint8_t)op;
                        n_ops += 1u;
                        if ((int)score <= 0) break;
                        if (i == 0 || j == 0) { score = 0; break; }
                        if (rho < 0 || !((vmask >> base) & 1ull)) continue;       // next cell not staged: start over from it
                    }
                    // ---- a run of alignment moves on diagonal `base`: H(i-1,j-1) = H - s(ref[j-1], read[i-1])   (:388-394) ----
                    uint32_t run = (uint32_t)__builtin_ctzll(~((amask >> base) & 0x1FFFFFull));     // 0..21
                    if (run > 0) {
                        const uint64_t range = ((1ull << run) - 1ull) << base;
                        if ((int)score > (int)(run * udec)) {
                            // no move lowers the score by more than udec: it stays positive through the whole run
                            const uint32_t nm = (uint32_t)__builtin_popcountll(mm & range);
                            score -= nm * umat + (run - nm) * umis;
                        } else {
                            const bool in_run = (range >> lane) & 1ull;
                            const uint32_t cm = lanemask_lt_count(mm & range) + (mt ? 1u : 0u);   // matches among the run's lanes up to this one
                            const uint32_t after = score - (cm * umat + (lane - base + 1u - cm) * umis);   // H after this lane's move
                            const uint64_t z = BALLOT(in_run && (int)after <= 0);
                            if (z) { run = (uint32_t)__builtin_ctzll(z) - base + 1u; done = true; }   // `while (score > 0)` stops there
                            score = (uint32_t)__builtin_amdgcn_readlane((int)after, base + run - 1u);
                        }
                        begin = (int)(j - (run - 1u));
                        if (lane >= base && lane < base + run && n_ops + (lane - base) < 4u * A.lds_words)
                            ops_b[n_ops + (lane - base)] = (uint8_t)SWMI_DIR_A;
                        n_ops += run; i -= run; j -= run; rho -= (int)run;
                        if (done || (int)score <= 0) break;
                        if (i == 0 || j == 0) { score = 0; break; }
                        if (rho < 0) break;                                         // continues in the strip above
                    }
                    // ---- the gap move that ended the run, if that cell was inspected ----
                    const uint32_t nxt = base + run;
                    if (run < 21u && ((vmask >> nxt) & 1ull)) {
                        begin = (int)j;
                        score -= ugap;
                        uint32_t op;
                        if ((imask >> nxt) & 1ull) { --i; --rho; op = SWMI_DIR_I; } else { --j; op = SWMI_DIR_D; }
                        if (lane == 0 && n_ops < 4u * A.lds_words) ops_b[n_ops] = (uint8_t)op;
                        n_ops += 1u;
                        if ((int)score <= 0) break;
                        if (i == 0 || j == 0) { score = 0; break; }
                        if (rho < 0) break;
                    }
                }
#endif
                if (TB_DBG) tk_walk += __builtin_amdgcn_s_memtime() - tw0;
            }
            if (TB_DBG) n_steps += n_ops;
            WAVE_SYNC();

            // ---- the record: a table entry + the payload (ops packed 2 bits each, 16 per dword [+ the two aligned strings]) ----
            const uint32_t opw = A.raw ? 0u : (n_ops + 15u) / 16u;          // (records with strings carry no ops)
            const uint32_t words = swmi_payload_words(n_ops, A.raw != nullptr);
            const SwmiReserve rsv = swmi_reserve_issue(A, lane, words, 1u);
            // while the reservation is on its way: this lane's first dword of packed ops, and the last 256 characters of
            // GetAlignment's two strings (SmithWaterman.java:418-431) from the caller's own bytes.  The walker's current direction
            // tile is free between two walks (a speculative span goes to the other one) and serves as scratch.
            const bool staged = n_ops <= 4u * A.lds_words;
            auto pack16 = [&](uint32_t w) {
                uint32_t packed = 0;
#pragma unroll
                for (uint32_t c = 0; c < 4; ++c) {
                    const uint32_t first = 16u * w + 4u * c;
                    uint32_t x = first < n_ops ? lds_ops[4u * w + c] : 0u;
                    if (first + 4u > n_ops && first < n_ops) x &= (1u << (8u * (n_ops - first))) - 1u;
                    const uint32_t b8 = (x & 3u) | ((x >> 6) & 0xCu) | ((x >> 12) & 0x30u) | ((x >> 18) & 0xC0u);
                    packed |= b8 << (8u * c);
                }
                return packed;
            };
            const uint32_t packed0 = (staged && lane < opw) ? pack16(lane) : 0u;
            SwmiStrings<SwmiOpsPerByte> strs(SwmiOpsPerByte{ops_b}, staged ? n_ops : 0u, ci, cj, raw_ref, raw_read, lane, tile_cur);
            const uint32_t sw = strs.words(), ctop = strs.n_chunks();
            uint32_t wr0 = 0, wq0 = 0;
            if (A.raw && staged) strs.chunk(ctop - 1u, wr0, wq0);
            unsigned long long off;
            uint32_t rslot;
            if (swmi_reserve_finish(A, rsv, words, 1u, off, rslot) && staged) {
                uint32_t *dst = A.arena + off;
                if (lane == 0) swmi_write_rec(A, rslot, pd.out_id, crank, begin, ci, cj, n_ops, off);
                if (lane < opw) dst[lane] = packed0;
                for (uint32_t w = lane + WAVE; w < opw; w += WAVE) dst[w] = pack16(w);
                if (A.raw) {
                    const uint32_t w = 64u * (ctop - 1u) + lane;
                    if (w < sw) { dst[w] = wr0; dst[sw + w] = wq0; }
                    strs.store_from(dst, ctop - 1u);
                }
            } else if (lane == 0) {
                atomicOr(&A.out[pd.out_id].flags, SWMI_F_ARENA_OVF);
                if (A.ovf_host) *A.ovf_host = 1u;
            }
            WAVE_SYNC();
        }
    }
    if (COOP && lane == 0)                                                         // releases this team's helpers
        __hip_atomic_store(const_cast<uint32_t *>(&shared[7u + 4u * slot]), 0xFFFFFFFFu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (TB_DBG && lane == 0 && slot == 0) {
        A.dbg[4 * pd.out_id] = __builtin_amdgcn_s_memtime() - tk0;
        A.dbg[4 * pd.out_id + 1] = tk_walk;
        A.dbg[4 * pd.out_id + 2] = n_steps | (tk_stage << 16);       // (steps < 65536 in the diagnostics runs)
        A.dbg[4 * pd.out_id + 3] = n_iters;
    }
}

// The helper side of COOP: wave `wave` >= nw serves team (wave - nw) % nw as its window number 1 + (wave - nw) / nw.
template <int R>
__device__ __forceinline__ void coop_helper(const TraceArgs &A, const PairDesc pd, const uint32_t lane, const uint32_t wave,
                                            const uint32_t nw, const uint32_t ts, const uint32_t n_waves_all,
                                            uint32_t *__restrict__ tiles, volatile uint32_t *__restrict__ shared) {
    const SeqDesc rd = A.refs[pd.ref_id];
    const SeqDesc qd = A.reads[pd.read_id];
    const uint32_t n = rd.len, m = qd.len;
    const uint32_t *__restrict__ refw = A.seqw + rd.boff;
    const uint32_t *__restrict__ readw = A.seqw + qd.boff;
    const StripGeom G = strip_geom<R>(m, n, 1u);
    const bool acgt = rd.acgt && qd.acgt && SWMI_SCORES_FIT(A);
    const uint32_t team = (wave - nw) % nw, q = 1u + (wave - nw) / nw;
    uint32_t last = 0;
    for (;;) {
        uint32_t seq;
        while ((seq = __hip_atomic_load(const_cast<uint32_t *>(&shared[7u + 4u * team]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) == last)
            __builtin_amdgcn_s_sleep(SWMI_HELPER_SLEEP);
        if (seq == 0xFFFFFFFFu) break;
        last = seq;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        // request: flags bit 0 = the team's second buffer, bit 1 = speculative (the walker takes no window: helper q takes window
        // q-1), bits 2.. = the strip
        const uint32_t flags = shared[4u + 4u * team], wlo = shared[5u + 4u * team], nq = shared[6u + 4u * team];
        const uint32_t s = flags >> 2;
        const uint32_t win = (flags & 2u) ? q - 1u : q;
        if (q < ts && win < nq) {
            uint32_t *__restrict__ base = ((flags & 1u) ? tiles + n_waves_all * (SWMI_CK_BLOCKS * SWMI_RMAX * WAVE) : tiles) +
                                          team * ts * (SWMI_CK_BLOCKS * SWMI_RMAX * WAVE);
            (void)replay_any<R, false, false>(A, pd, n, m, acgt, refw, readw, G, s, wlo + win * SWMI_CK_BLOCKS, lane,
                                       base + win * SWMI_CK_BLOCKS * R * WAVE, 0, 0u, nullptr, 0u);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) atomicAdd(const_cast<uint32_t *>(&shared[20u + team]), 1u);
        }
    }
}

// 4 wavefronts per workgroup (one pair each, like the sweep) so that the walks of slot 0 -- one per pair -- land
// one per SIMD; the extra slots' few walks fall where they may.
template <int TMODE>
__device__ __forceinline__ void traceback_entry(const TraceArgs &A, uint32_t *lds_all) {
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t pair = blockIdx.x * FILL_WAVES + wave;
    if (pair >= A.n_pairs) return;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t slot = blockIdx.y;
    uint32_t *tb_lds = lds_all + wave * (A.lds_words + A.lds_read_words + SWMI_TB_REFWIN_WORDS +
                                         (TMODE == 0 ? SWMI_TB_BLOCKS : SWMI_CK_BLOCKS) * SWMI_RMAX * WAVE);
    uint32_t *tile = tb_lds + A.lds_words + A.lds_read_words + SWMI_TB_REFWIN_WORDS;
    const PairDesc pd = A.pairs[pair];
    const PairOut po = A.out[pd.out_id];
    if (A.out_host && slot == 0 && lane == 0) A.out_host[pd.out_id] = po;      // result straight into pinned host memory
    if (po.flags & (SWMI_F_DEGENERATE | SWMI_F_CELL_OVF)) return;
    if (po.n_cells <= slot) return;
    const uint32_t R = swmi_rows_per_lane(A.reads[pd.read_id].len);
    if (R == 1)      traceback_pair<1, TMODE, false>(A, pd, po, lane, slot, SWMI_TB_SLOTS, tb_lds, tile, nullptr, 1u);
    else if (R == 2) traceback_pair<2, TMODE, false>(A, pd, po, lane, slot, SWMI_TB_SLOTS, tb_lds, tile, nullptr, 1u);
    else if (R == 3) traceback_pair<3, TMODE, false>(A, pd, po, lane, slot, SWMI_TB_SLOTS, tb_lds, tile, nullptr, 1u);
    else             traceback_pair<4, TMODE, false>(A, pd, po, lane, slot, SWMI_TB_SLOTS, tb_lds, tile, nullptr, 1u);
}

// mode 1, before the walks: the pair's maximum cells are listed, and the first span of each walk is prepared at the
// same time.  The sweep left one maximum per checkpoint window; a window whose maximum equals the pair's is a candidate.
//   * one candidate (the usual case): wave 0 re-sweeps it with the cell test on, the other waves re-sweep the windows
//     below it, so the span a single alignment's walk starts in is complete when the cell list is;
//   * 2..4 candidates: wave w re-sweeps candidate w (cells into its own quarter of the pair's cell list), the remaining
//     waves the windows below their team's candidate.  If every candidate holds exactly ONE maximum cell the quarters
//     are compacted and walker w starts in its own window; otherwise the generic path below runs;
//   * generic: wave 0 re-sweeps all candidates one after the other (detect_cells).
// On return the workgroup has passed a barrier, shared[0] = number of cells, shared[3] = 1 if every walker's first span
// is staged; the return value is this wave's first staged block (~0: none).  shared[24..27] / [28..31]: per candidate
// cell count / first staged block.
template <int R>
__device__ __forceinline__ uint32_t winmax_detect(const TraceArgs &A, const PairDesc pd, PairOut &po, const uint32_t lane,
                                                  const uint32_t wave, const uint32_t n_waves, const uint32_t ccap,
                                                  uint32_t *__restrict__ tiles, uint32_t *__restrict__ walker_lds,
                                                  const uint32_t per_walker, volatile uint32_t *__restrict__ shared) {
    const SeqDesc rd = A.refs[pd.ref_id];
    const SeqDesc qd = A.reads[pd.read_id];
    const uint32_t n = rd.len, m = qd.len;
    const uint32_t *__restrict__ refw = A.seqw + rd.boff;
    const uint32_t *__restrict__ readw = A.seqw + qd.boff;
    const bool acgt = rd.acgt && qd.acgt && SWMI_SCORES_FIT(A);
    const StripGeom G = strip_geom<R>(m, n, 1u);
    constexpr uint32_t WIN_WORDS = SWMI_CK_BLOCKS * SWMI_RMAX * WAVE;
    uint32_t ncand = 0, gc[SWMI_TB_SLOTS] = {0u, 0u, 0u, 0u};
    if (G.n_strips == 1u) {
        const uint32_t *__restrict__ wm = A.dir + pd.dir_off + G.wmax_off;
        for (uint32_t g0 = 0; g0 < G.n_ck; g0 += WAVE) {
            const uint32_t g = g0 + lane;
            const int wv = g < G.n_ck ? (int)wm[g] : -1;
            uint64_t cand = BALLOT(wv == po.score);
            while (cand) {
                const uint32_t gg = g0 + (uint32_t)__builtin_ctzll(cand);
                cand &= cand - 1ull;
                if (ncand == 0u) gc[0] = gg; else if (ncand == 1u) gc[1] = gg; else if (ncand == 2u) gc[2] = gg; else if (ncand == 3u) gc[3] = gg;
                ++ncand;
            }
        }
    }
    auto stage_ref = [&](uint32_t *__restrict__ dst, uint32_t wlo, uint32_t nwin) {
        const int clo = (int)(16u * wlo) - 63;
        const uint32_t cw0 = clo > 0 ? (uint32_t)clo >> 2 : 0u;
        const uint32_t cw1 = (16u * (wlo + nwin * SWMI_CK_BLOCKS) - 1u) >> 2;
        for (uint32_t x = cw0 + lane; x <= cw1 && x < (n + 3u) / 4u; x += WAVE) dst[x - cw0] = refw[x];
    };
    auto publish = [&](uint32_t cnt, uint32_t staged) {      // wave 0, after the cells are listed
        if (lane == 0) {
            po.n_cells = cnt;
            if (cnt > ccap) po.flags |= SWMI_F_CELL_OVF;
            A.out[pd.out_id] = po;
            if (A.out_host) A.out_host[pd.out_id] = po;
            shared[0] = cnt;
            shared[3] = staged;
        }
        if (lane < SWMI_TB_SLOTS) { shared[7u + 4u * lane] = 0u; shared[20u + lane] = 0u; }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the cell list has left the CU before the others read it
    };
    const uint64_t cbase = A.cells_off ? A.cells_off[pd.out_id] : (uint64_t)pd.out_id * A.cell_cap;
    uint2 *__restrict__ cells = const_cast<uint2 *>(A.cells) + cbase;

    const uint32_t seg = ccap / SWMI_TB_SLOTS;
    if (ncand >= 2u && ncand <= SWMI_TB_SLOTS && ncand <= n_waves && seg >= 1u) {
        const uint32_t nw = ncand, ts = n_waves / nw;
        if (wave < nw) {
            const uint32_t g = wave == 0u ? gc[0] : wave == 1u ? gc[1] : wave == 2u ? gc[2] : gc[3];
            const uint32_t nq0 = g + 1u < ts ? g + 1u : ts;
            const uint32_t wlo0 = (g + 1u - nq0) * SWMI_CK_BLOCKS;
            uint32_t *my = walker_lds + wave * per_walker;
            for (uint32_t w = lane; w < (m + 3u) / 4u; w += WAVE) my[A.lds_words + w] = readw[w];
            stage_ref(my + A.lds_words + A.lds_read_words, wlo0, nq0);
            const uint32_t c = replay_any<R, true, false>(A, pd, n, m, acgt, refw, readw, G, 0u, g * SWMI_CK_BLOCKS, lane,
                                                   tiles + wave * ts * WIN_WORDS + (nq0 - 1u) * SWMI_CK_BLOCKS * R * WAVE,
                                                   po.score, 0u, cells + wave * seg, seg);
            if (lane == 0) { shared[24u + wave] = c; shared[28u + wave] = wlo0; }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            const uint32_t team = (wave - nw) % nw, q = 1u + (wave - nw) / nw;
            const uint32_t g = team == 0u ? gc[0] : team == 1u ? gc[1] : team == 2u ? gc[2] : gc[3];
            const uint32_t nq0 = g + 1u < ts ? g + 1u : ts;
            if (q < nq0)
                (void)replay_any<R, false, false>(A, pd, n, m, acgt, refw, readw, G, 0u, (g - q) * SWMI_CK_BLOCKS, lane,
                                           tiles + team * ts * WIN_WORDS + (nq0 - 1u - q) * SWMI_CK_BLOCKS * R * WAVE,
                                           0, 0u, nullptr, 0u);
        }
        __syncthreads();
        bool one_each = true;
        for (uint32_t w = 0; w < nw; ++w) one_each = one_each && shared[24u + w] == 1u;
        if (one_each) {
            if (wave == 0) {
                uint2 c = make_uint2(0, 0);
                if (lane < nw) { c.x = ld_l2(&cells[lane * seg].x); c.y = ld_l2(&cells[lane * seg].y); }
                if (lane < nw) cells[lane] = c;
                publish(nw, 1u);
            }
            __syncthreads();
            return wave < nw ? shared[28u + wave] : 0xFFFFFFFFu;
        }
        __syncthreads();                                     // everybody has read the counts before wave 0 reuses shared[]
    }

    const bool pre = ncand == 1u;
    const uint32_t nq0 = gc[0] + 1u < n_waves ? gc[0] + 1u : n_waves;             // windows of the span that ends with window gc[0]
    const uint32_t wlo0 = (gc[0] + 1u - nq0) * SWMI_CK_BLOCKS;
    if (wave == 0) {
        if (n_waves == 1u) {                                 // nobody else to do it
            for (uint32_t w = lane; w < (m + 3u) / 4u; w += WAVE) walker_lds[A.lds_words + w] = readw[w];
            if (pre) stage_ref(walker_lds + A.lds_words + A.lds_read_words, wlo0, nq0);
        }
        const uint32_t cnt = detect_cells<R, false>(A, pd, po, lane, pre ? tiles + (nq0 - 1u) * SWMI_CK_BLOCKS * R * WAVE : tiles);
        publish(cnt, (pre && cnt == 1u) ? 1u : 0u);       // (the candidate's window doubles as the first window of the walk)
    } else {
        // the read's codes for the walkers: wave w fills walker w's copy, the last wave also walker 0's
        if (wave < SWMI_TB_SLOTS)
            for (uint32_t w = lane; w < (m + 3u) / 4u; w += WAVE) walker_lds[wave * per_walker + A.lds_words + w] = readw[w];
        if (wave == n_waves - 1u)
            for (uint32_t w = lane; w < (m + 3u) / 4u; w += WAVE) walker_lds[A.lds_words + w] = readw[w];
        if (pre) {
            if (wave < nq0)
                (void)replay_any<R, false, false>(A, pd, n, m, acgt, refw, readw, G, 0u, wlo0 + (wave - 1u) * SWMI_CK_BLOCKS, lane,
                                           tiles + (wave - 1u) * SWMI_CK_BLOCKS * R * WAVE, 0, 0u, nullptr, 0u);
            if (wave == n_waves - 1u) stage_ref(walker_lds + A.lds_words + A.lds_read_words, wlo0, nq0);
        }
    }
    __syncthreads();
    return (pre && wave == 0) ? wlo0 : 0xFFFFFFFFu;
}

// mode 1: one workgroup of SWMI_TB_WAVES waves = ONE pair.  Wave 0 first lists the maximum cells (detect_cells),
// the workgroup meets at a barrier, then min(cells, 4) waves walk the alignments and the others help them.
// (4 waves per SIMD = at most 128 VGPRs: at the headline the 1000 workgroups of four waves must ALL be resident, or the launch
//  grows a second round of workgroups -- at 130 VGPRs it took 0.092 ms instead of 0.073)
extern "C" __global__ void __launch_bounds__(WAVE * SWMI_TB_WAVES) __attribute__((amdgpu_waves_per_eu(4, 4)))
sw_traceback_winmax_kernel(const TraceArgs A) {
    extern __shared__ uint32_t wm_lds[];
    const uint32_t pair = blockIdx.x;
    if (pair >= A.n_pairs) return;
    // (rotating the walker role over the hardware waves, in case wave w always landed on SIMD w, changes nothing)
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const PairDesc pd = A.pairs[pair];
    PairOut po = A.out[pd.out_id];
    if (finish_pair(A, pd, po) && wave == 0 && lane == 0) A.out[pd.out_id] = po;
    if (po.flags & (SWMI_F_DEGENERATE | SWMI_F_DONE)) {      // (DONE: sw_resident_pairs_kernel did the whole pair) same decision in every wave: nobody waits at the barrier
        if (A.out_host && wave == 0 && lane == 0) A.out_host[pd.out_id] = po;
        return;
    }
    // LDS: [32 shared words][tiles: one window per wave][per walker: ops staging, read codes, reference window]
    const uint32_t n_waves = blockDim.x >> 6;
    const uint32_t per_walker = A.lds_words + A.lds_read_words + SWMI_TB_REFWIN_WORDS;
    constexpr uint32_t WIN_WORDS = SWMI_CK_BLOCKS * SWMI_RMAX * WAVE;
    uint32_t *shared = wm_lds;
    uint32_t *tiles = wm_lds + 32;
    const uint32_t R = swmi_rows_per_lane(A.reads[pd.read_id].len);
    const uint32_t ccap = A.cells_cap ? A.cells_cap[pd.out_id] : A.cell_cap;
    // (A.pad2: the launcher reserved a SECOND set of window tiles for speculative staging -- traceback_pair)
    const uint32_t tile_sets = A.pad2 ? 2u : 1u;
    uint32_t *tiles2 = A.pad2 ? tiles + n_waves * WIN_WORDS : nullptr;
    uint32_t *walker_lds0 = tiles + tile_sets * n_waves * WIN_WORDS;
    if (!swmi_common_pair(A, A.refs[pd.ref_id], A.reads[pd.read_id])) {
        // a byte alphabet, the DistributedSW tie order or a read of several strips: the plain scheme -- wave 0 lists the
        // cells, then up to four independent walkers, one window at a time -- with every kernel variant available
        if (wave == 0) {
            uint32_t c;
            if (R == 1)      c = detect_cells<1>(A, pd, po, lane, tiles);
            else if (R == 2) c = detect_cells<2>(A, pd, po, lane, tiles);
            else if (R == 3) c = detect_cells<3>(A, pd, po, lane, tiles);
            else             c = detect_cells<4>(A, pd, po, lane, tiles);
            if (lane == 0) {
                po.n_cells = c;
                if (c > ccap) po.flags |= SWMI_F_CELL_OVF;
                A.out[pd.out_id] = po;
                if (A.out_host) A.out_host[pd.out_id] = po;
                shared[0] = c;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the cell list has left the CU before the others read it
        }
        __syncthreads();
        const uint32_t c = shared[0];
        uint32_t nwr = c < SWMI_TB_SLOTS ? c : SWMI_TB_SLOTS;
        if (nwr > n_waves) nwr = n_waves;
        if (c > ccap || wave >= nwr) return;
        po.n_cells = c;
        uint32_t *lds = walker_lds0 + wave * per_walker;
        uint32_t *tile = tiles + wave * WIN_WORDS;
        if (R == 1)      traceback_pair<1, 1, false>(A, pd, po, lane, wave, nwr, lds, tile, nullptr, 1u);
        else if (R == 2) traceback_pair<2, 1, false>(A, pd, po, lane, wave, nwr, lds, tile, nullptr, 1u);
        else if (R == 3) traceback_pair<3, 1, false>(A, pd, po, lane, wave, nwr, lds, tile, nullptr, 1u);
        else             traceback_pair<4, 1, false>(A, pd, po, lane, wave, nwr, lds, tile, nullptr, 1u);
        return;
    }
    uint32_t pre_wlo;
    if (R == 1)      pre_wlo = winmax_detect<1>(A, pd, po, lane, wave, n_waves, ccap, tiles, walker_lds0, per_walker, shared);
    else if (R == 2) pre_wlo = winmax_detect<2>(A, pd, po, lane, wave, n_waves, ccap, tiles, walker_lds0, per_walker, shared);
    else if (R == 3) pre_wlo = winmax_detect<3>(A, pd, po, lane, wave, n_waves, ccap, tiles, walker_lds0, per_walker, shared);
    else             pre_wlo = winmax_detect<4>(A, pd, po, lane, wave, n_waves, ccap, tiles, walker_lds0, per_walker, shared);
    const uint32_t cnt = shared[0];                                    // (winmax_detect ends with a barrier)
    if (shared[3] != 1u) pre_wlo = 0xFFFFFFFFu;                        // no staged first spans
    if (cnt > ccap || cnt == 0u) return;
    po.n_cells = cnt;
    uint32_t nw = cnt < SWMI_TB_SLOTS ? cnt : SWMI_TB_SLOTS;           // walkers = teams
    if (nw > n_waves) nw = n_waves;                                    // (a one-wave workgroup walks its alignments one after the other)
    const uint32_t ts = n_waves / nw;                                  // waves (= windows per round) per team
    if (ts == 1u) {
        // no helpers to share the re-sweeps with: the walkers run independently, one window at a time
        if (wave >= nw) return;
        uint32_t *lds = walker_lds0 + wave * per_walker;
        uint32_t *tile = tiles + wave * WIN_WORDS;
        if (R == 1)      traceback_pair<1, 1, false, false>(A, pd, po, lane, wave, nw, lds, tile, nullptr, 1u, pre_wlo, true);
        else if (R == 2) traceback_pair<2, 1, false, false>(A, pd, po, lane, wave, nw, lds, tile, nullptr, 1u, pre_wlo, true);
        else if (R == 3) traceback_pair<3, 1, false, false>(A, pd, po, lane, wave, nw, lds, tile, nullptr, 1u, pre_wlo, true);
        else             traceback_pair<4, 1, false, false>(A, pd, po, lane, wave, nw, lds, tile, nullptr, 1u, pre_wlo, true);
        return;
    }
    if (wave < nw) {
        uint32_t *lds = walker_lds0 + wave * per_walker;
        uint32_t *tile = tiles + wave * ts * WIN_WORDS;
        uint32_t *tile2 = tiles2 ? tiles2 + wave * ts * WIN_WORDS : nullptr;
        if (R == 1)      traceback_pair<1, 1, true, false>(A, pd, po, lane, wave, nw, lds, tile, shared, ts, pre_wlo, true, nullptr, tile2);
        else if (R == 2) traceback_pair<2, 1, true, false>(A, pd, po, lane, wave, nw, lds, tile, shared, ts, pre_wlo, true, nullptr, tile2);
        else if (R == 3) traceback_pair<3, 1, true, false>(A, pd, po, lane, wave, nw, lds, tile, shared, ts, pre_wlo, true, nullptr, tile2);
        else             traceback_pair<4, 1, true, false>(A, pd, po, lane, wave, nw, lds, tile, shared, ts, pre_wlo, true, nullptr, tile2);
    } else {
        if (R == 1)      coop_helper<1>(A, pd, lane, wave, nw, ts, n_waves, tiles, shared);
        else if (R == 2) coop_helper<2>(A, pd, lane, wave, nw, ts, n_waves, tiles, shared);
        else if (R == 3) coop_helper<3>(A, pd, lane, wave, nw, ts, n_waves, tiles, shared);
        else             coop_helper<4>(A, pd, lane, wave, nw, ts, n_waves, tiles, shared);
    }
}

extern "C" __global__ void __launch_bounds__(WAVE * FILL_WAVES)
sw_traceback_kernel(const TraceArgs A) {
    extern __shared__ uint32_t tb_lds[];
    traceback_entry<0>(A, tb_lds);
}

extern "C" __global__ void __launch_bounds__(WAVE * FILL_WAVES)
sw_traceback_replay_kernel(const TraceArgs A) {
    extern __shared__ uint32_t tb_lds[];
    traceback_entry<1>(A, tb_lds);
}

// ------------------------------------------------------------------------------------------------
// split traceback (mode 1): for batches whose pairs carry many tied maxima (periodic references: the reference's own
// EngineerData sets, one tied maximum per period) or that have few pairs, one workgroup per pair is the wrong grain --
// a 128 kbp periodic reference against one read is ONE pair with 1600 alignments.  Here the grain is the window and
// the alignment:
//   sw_detect_windows_kernel  one wavefront per checkpoint window of every pair: a window whose maximum equals the
//                             pair's is re-swept with the cell test on; its cells go to the pair's list (slots reserved
//                             by one atomicAdd, any order: the host orders a pair's records by cell) and one walk item
//                             per cell to a global queue;
//   sw_walk_items_kernel      a fixed grid of wavefronts shares the queue (item w, w + W, ...): one alignment per
//                             wavefront, windows re-swept one at a time.
// ------------------------------------------------------------------------------------------------
#define SWMI_SPLIT_WAVES 4u

extern "C" __global__ void __launch_bounds__(WAVE * SWMI_SPLIT_WAVES)
sw_detect_windows_kernel(const TraceArgs A) {
    extern __shared__ uint32_t dw_lds[];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t item = blockIdx.x * SWMI_SPLIT_WAVES + wave;
    const uint32_t n_items = A.win_off[A.n_pairs];
    if (item >= n_items) return;
    // the pair this window belongs to: last p with win_off[p] <= item
    uint32_t lo = 0, hi = A.n_pairs;
    while (hi - lo > 1u) {
        const uint32_t mid = (lo + hi) >> 1;
        if (A.win_off[mid] <= item) lo = mid; else hi = mid;
    }
    const uint32_t pair = lo, wloc = item - A.win_off[pair];
    const PairDesc pd = A.pairs[pair];
    PairOut po = A.out[pd.out_id];
    if (finish_pair(A, pd, po) && wloc == 0u && lane == 0) A.out[pd.out_id] = po;     // (window 0's wave completes the record)
    if (po.flags & (SWMI_F_DEGENERATE | SWMI_F_DONE)) return;                          // (DONE: sw_resident_pairs_kernel did the whole pair)
    const SeqDesc rd = A.refs[pd.ref_id];
    const SeqDesc qd = A.reads[pd.read_id];
    const uint32_t n = rd.len, m = qd.len;
    const uint32_t R = swmi_rows_per_lane(m);
    const uint64_t wblocks = ((uint64_t)n + 63u + 15u) / 16u;
    const uint32_t n_ck = (uint32_t)((wblocks + SWMI_CK_BLOCKS - 1u) / SWMI_CK_BLOCKS);
    const uint32_t strip = wloc / n_ck, g = wloc - strip * n_ck;
    const uint64_t wmax_off = (uint64_t)n_ck * (R + 2) * WAVE;
    const uint64_t strip_words = wmax_off + (uint64_t)((n_ck + 63u) & ~63u);
    if ((int)A.dir[pd.dir_off + strip * strip_words + wmax_off + g] != po.score) return;

    // this wave's scratch: the re-swept window (direction bits nobody reads here) and the cells it finds
    constexpr uint32_t WIN_WORDS = SWMI_CK_BLOCKS * SWMI_RMAX * WAVE;
    uint32_t *tile = dw_lds + wave * (WIN_WORDS + 2u * SWMI_DETECT_LCAP);
    uint2 *found = reinterpret_cast<uint2 *>(tile + WIN_WORDS);
    const uint32_t *__restrict__ refw = A.seqw + rd.boff;
    const uint32_t *__restrict__ readw = A.seqw + qd.boff;
    const bool acgt = rd.acgt && qd.acgt && SWMI_SCORES_FIT(A);
    uint32_t cnt;
    if (R == 1)      { const StripGeom G = strip_geom<1>(m, n, 1u); cnt = replay_any<1, true>(A, pd, n, m, acgt, refw, readw, G, strip, g * SWMI_CK_BLOCKS, lane, tile, po.score, 0u, found, SWMI_DETECT_LCAP); }
    else if (R == 2) { const StripGeom G = strip_geom<2>(m, n, 1u); cnt = replay_any<2, true>(A, pd, n, m, acgt, refw, readw, G, strip, g * SWMI_CK_BLOCKS, lane, tile, po.score, 0u, found, SWMI_DETECT_LCAP); }
    else if (R == 3) { const StripGeom G = strip_geom<3>(m, n, 1u); cnt = replay_any<3, true>(A, pd, n, m, acgt, refw, readw, G, strip, g * SWMI_CK_BLOCKS, lane, tile, po.score, 0u, found, SWMI_DETECT_LCAP); }
    else             { const StripGeom G = strip_geom<4>(m, n, 1u); cnt = replay_any<4, true>(A, pd, n, m, acgt, refw, readw, G, strip, g * SWMI_CK_BLOCKS, lane, tile, po.score, 0u, found, SWMI_DETECT_LCAP); }
    if (cnt == 0u) return;                                   // (pad rows can make a window's maximum a value no real cell holds)
    WAVE_SYNC();
    const uint64_t cbase = A.cells_off ? A.cells_off[pd.out_id] : (uint64_t)pd.out_id * A.cell_cap;
    const uint32_t ccap = A.cells_cap ? A.cells_cap[pd.out_id] : A.cell_cap;
    unsigned long long base = 0;
    uint32_t qbase = 0;
    const uint32_t have = cnt < SWMI_DETECT_LCAP ? cnt : SWMI_DETECT_LCAP;
    if (lane == 0) {
        base = atomicAdd((unsigned long long *)&A.out[pd.out_id].n_cells, (unsigned long long)cnt);
        const bool fits = cnt <= SWMI_DETECT_LCAP && base + cnt <= (unsigned long long)ccap;
        if (fits) {
            qbase = atomicAdd(A.q_count, have);
            if (qbase + have > A.q_cap) qbase = 0xFFFFFFFFu;
        } else {
            qbase = 0xFFFFFFFFu;
        }
        if (qbase == 0xFFFFFFFFu) atomicOr(&A.out[pd.out_id].flags, SWMI_F_CELL_OVF);     // the host re-runs the pair with exact sizes
    }
    base = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(base >> 32)) << 32) |
           (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)base);
    qbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)qbase);
    if (qbase == 0xFFFFFFFFu) return;
    uint2 *__restrict__ cells = const_cast<uint2 *>(A.cells) + cbase;
    for (uint32_t c = lane; c < have; c += WAVE) {
        const uint2 cell = found[c];
        cells[base + c] = cell;
        A.q_items[qbase + c] = make_uint4(pair, cell.x, cell.y, 0u);
    }
}

extern "C" __global__ void __launch_bounds__(WAVE * SWMI_SPLIT_WAVES)
sw_walk_items_kernel(const TraceArgs A) {
    extern __shared__ uint32_t wi_lds[];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t gw = blockIdx.x * SWMI_SPLIT_WAVES + wave, n_waves = gridDim.x * SWMI_SPLIT_WAVES;
    // zero-copy results: the pair outputs are final (the detect kernel ended one launch ago)
    if (A.out_host)
        for (uint32_t p = gw * WAVE + lane; p < A.n_pairs; p += n_waves * WAVE) A.out_host[p] = A.out[p];
    uint32_t n_items = *A.q_count;
    if (n_items > A.q_cap) n_items = A.q_cap;
    constexpr uint32_t WIN_WORDS = SWMI_CK_BLOCKS * SWMI_RMAX * WAVE;
    const uint32_t per_wave = A.lds_words + A.lds_read_words + SWMI_TB_REFWIN_WORDS + WIN_WORDS;
    uint32_t *lds = wi_lds + wave * per_wave;
    uint32_t *tile = lds + A.lds_words + A.lds_read_words + SWMI_TB_REFWIN_WORDS;
    for (uint32_t it = gw; it < n_items; it += n_waves) {
        const uint4 item = A.q_items[it];
        const PairDesc pd = A.pairs[item.x];
        PairOut po = A.out[pd.out_id];
        if (po.flags & (SWMI_F_DEGENERATE | SWMI_F_CELL_OVF)) continue;         // (an overflowed pair is re-run as a whole)
        const uint32_t R = swmi_rows_per_lane(A.reads[pd.read_id].len);
        if (R == 1)      traceback_pair<1, 1, false>(A, pd, po, lane, 0u, 1u, lds, tile, nullptr, 1u, 0xFFFFFFFFu, false, &item);
        else if (R == 2) traceback_pair<2, 1, false>(A, pd, po, lane, 0u, 1u, lds, tile, nullptr, 1u, 0xFFFFFFFFu, false, &item);
        else if (R == 3) traceback_pair<3, 1, false>(A, pd, po, lane, 0u, 1u, lds, tile, nullptr, 1u, 0xFFFFFFFFu, false, &item);
        else             traceback_pair<4, 1, false>(A, pd, po, lane, 0u, 1u, lds, tile, nullptr, 1u, 0xFFFFFFFFu, false, &item);
        WAVE_SYNC();
    }
}

// ------------------------------------------------------------------------------------------------
// resident pairs (mode 1): a pair whose WHOLE 2-bit direction field fits a wavefront's share of LDS -- the reference's
// own benchmark shapes, 80 bp reads against 400 bp references (EngineerData.java:51-224) -- is handled start to finish
// by one wavefront, with nothing but its result leaving the CU:
//   A  score sweep (the 3-VALU cells of sweep_fast), one maximum per 32-step window kept in LDS -> the pair's maximum;
//   B  second sweep with direction bits (the same cell stream the replay uses) into LDS, with the cell test switched on
//      in the windows whose maximum equals the pair's -> the tied maximum cells, listed in LDS;
//   C  ALL alignments walked at once, one LANE each: a lane chases its own path through the field in LDS (three LDS
//      reads and ~35 VALU per step for up to 64 alignments together), tracking the score like SmithWaterman.java:380-409,
//      packing its ops into its own scratch row; the wave then reserves arena space with one atomicAdd and the lanes
//      copy their records out.  Periodic references give every pair a handful of alignments: they cost one walk, not five.
// No checkpoints, no HBM workspace.  The host orders a pair's records by cell (SWMI_RANK_BY_CELL).
// LDS per wavefront (dwords): field [wblocks*R*64] | window maxima [n_ck] | cells [2*cell_cap] | ops [64*ops_words] |
//                             reference codes [(n+3)/4+1] | read codes [(m+3)/4+1] | string scratch [128]
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t wave_scan_add_u32(uint32_t v) {          // inclusive prefix sum over the 64 lanes (DPP)
#define SWMI_DPP_ADD(ctrl, rmask, bmask)                                                     \
    { const uint32_t o_ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, rmask, bmask, true); v += o_; }
    SWMI_DPP_ADD(0x111, 0xf, 0xf)   // row_shr:1
    SWMI_DPP_ADD(0x112, 0xf, 0xf)   // row_shr:2
    SWMI_DPP_ADD(0x114, 0xf, 0xf)   // row_shr:4
    SWMI_DPP_ADD(0x118, 0xf, 0xf)   // row_shr:8  -> inclusive scan inside every row of 16
    SWMI_DPP_ADD(0x142, 0xa, 0xf)   // row_bcast:15 -> rows 1 and 3 add the total of the row before
    SWMI_DPP_ADD(0x143, 0xc, 0xf)   // row_bcast:31 -> rows 2 and 3 add the total of rows 0-1
#undef SWMI_DPP_ADD
    return v;
}

template <int R, bool STRICT>
__device__ __forceinline__ void resident_pair(const TraceArgs &A, const ResidentArgs &X, const PairDesc pd, const uint32_t lane, uint32_t *__restrict__ lds) {
    const SeqDesc rd = A.refs[pd.ref_id];
    const SeqDesc qd = A.reads[pd.read_id];
    const uint32_t n = rd.len, m = qd.len;
    const uint32_t *__restrict__ refw = A.seqw + rd.boff;
    const uint32_t *__restrict__ readw = A.seqw + qd.boff;
    const uint32_t lact = (m + R - 1) / R;                                  // one strip: m <= 64 * R
    const uint32_t lane_eff = lane < lact ? lane : 0x40000000u;
    const uint32_t T = n + lact - 1, nblk = (T + 15u) / 16u, n_ck = (nblk + SWMI_CK_BLOCKS - 1u) / SWMI_CK_BLOCKS;
    uint32_t *field = lds;                                                  // [nblk][R][64]
    uint32_t *wmaxs = field + nblk * R * WAVE;
    uint2 *cells = reinterpret_cast<uint2 *>(wmaxs + ((n_ck + 1u) & ~1u));
    uint32_t *opsb = reinterpret_cast<uint32_t *>(cells + X.res_cell_cap);
    uint32_t *refc = opsb + WAVE * X.res_ops_words;
    uint32_t *readc = refc + (n + 3u) / 4u + 1u;
    uint32_t *scratch = readc + (m + 3u) / 4u + 1u;                         // [SWMI_EMIT_SCRATCH_WORDS] 256 characters of both strings
    for (uint32_t w = lane; w < (n + 3u) / 4u; w += WAVE) refc[w] = refw[w];
    for (uint32_t w = lane; w < (m + 3u) / 4u; w += WAVE) readc[w] = readw[w];
    const uint4 *__restrict__ refq = reinterpret_cast<const uint4 *>(refw);

    // ---- A: scores only ---------------------------------------------------------------------------------------------
    int pair_max = 0;
    {
        const uint32_t gm = (uint32_t)(-(int64_t)A.gap);
        const int one = 1;
        SweepState<R> S;
        build_profiles<R>(S.q, readw, lane * R, m, A.match, A.mismatch);
#pragma unroll
        for (int k = 0; k < R; ++k) { S.h[k] = 0; S.g[k] = 0; S.hp[k] = 0; }
        S.lmax = -1;
        uint4 wnext = refq[0];
        S.rby = 0;
        S.rbx = wave_shr1((int)(1u << (wnext.x & 31u)), 0);
        for (uint32_t tb = 0; tb < nblk; ++tb) {
            const uint4 w = wnext;
            wnext = refq[tb + 1];
            if ((tb % SWMI_CK_BLOCKS) == 0u && tb > 0u) {
                const int wm = wave_max_i32(lane < lact ? S.lmax : -1);
                if (lane == 0) wmaxs[tb / SWMI_CK_BLOCKS - 1u] = (uint32_t)wm;
                pair_max = pair_max > wm ? pair_max : wm;
                S.lmax = -1;
            }
            const uint32_t t0 = 16u * tb;
#ifndef SWMI_NO_ASM
            if (t0 + 15u < n) {
                SweepStep4Asm<R>::run(S.h, S.g, S.hp, S.q, S.rbx, S.rby, w.x, w.y, one, gm, S.lmax);
                SweepStep4Asm<R>::run(S.h, S.g, S.hp, S.q, S.rbx, S.rby, w.y, w.z, one, gm, S.lmax);
                SweepStep4Asm<R>::run(S.h, S.g, S.hp, S.q, S.rbx, S.rby, w.z, w.w, one, gm, S.lmax);
                SweepStep4Asm<R>::run(S.h, S.g, S.hp, S.q, S.rbx, S.rby, w.w, wnext.x, one, gm, S.lmax);
            } else
#endif
            {
                sweep_tail_block<R>(S, w, wnext.x, t0, lane_eff, n, one, gm);
            }
        }
        const int wm = wave_max_i32(lane < lact ? S.lmax : -1);
        if (lane == 0) wmaxs[(nblk - 1u) / SWMI_CK_BLOCKS] = (uint32_t)wm;
        pair_max = pair_max > wm ? pair_max : wm;
    }
    PairOut po;
    if (pair_max <= 0) {                                                   // every cell ties at 0: SmithWaterman.java:154,182-185
        po.score = 0; po.flags = SWMI_F_DEGENERATE | SWMI_F_DONE; po.n_cells = (uint64_t)m * n;
        if (lane == 0) { A.out[pd.out_id] = po; if (A.out_host) A.out_host[pd.out_id] = po; }
        return;
    }
    WAVE_SYNC();

    // ---- B: direction bits into LDS, the cells equal to the maximum listed on the way ----------------------------------
    uint32_t ncell;
    {
        FillState<R> S;
        S.thr = pair_max; S.cnt = 0; S.ev_prev = 0; S.events = 0; S.dbg_skip = false; S.lmax = -1;
        setup_rows<R, true>(S, readw, lane * R, m, A.match, A.mismatch);
        uint4 wnext = refq[0];
        for (uint32_t tb = 0; tb < nblk; ++tb) {
            const uint4 w = wnext;
            wnext = refq[tb + 1];
            const uint32_t t0 = 16u * tb;
            const bool steady = (t0 + 1u >= lact) && (t0 + 15u < n);
            const bool hot = (int)wmaxs[tb / SWMI_CK_BLOCKS] == pair_max;       // (wave-uniform: an LDS word)
            if (hot) {
                if (steady) fill_block16<R, true, STRICT, false, false, SWMI_MODE_DETECT>(S, w, t0, lane, lane_eff, n, m, lane * R, A.gap, A.match, A.mismatch,
                                                                                          0, false, false, nullptr, cells, X.res_cell_cap);
                else        fill_block16<R, true, STRICT, false, true, SWMI_MODE_DETECT>(S, w, t0, lane, lane_eff, n, m, lane * R, A.gap, A.match, A.mismatch,
                                                                                         0, false, false, nullptr, cells, X.res_cell_cap);
            } else {
                if (steady) fill_block16<R, true, STRICT, false, false, SWMI_MODE_REPLAY>(S, w, t0, lane, lane_eff, n, m, lane * R, A.gap, A.match, A.mismatch,
                                                                                          0, false, false, nullptr, nullptr, 0u);
                else        fill_block16<R, true, STRICT, false, true, SWMI_MODE_REPLAY>(S, w, t0, lane, lane_eff, n, m, lane * R, A.gap, A.match, A.mismatch,
                                                                                         0, false, false, nullptr, nullptr, 0u);
            }
            const int miss = (int)(t0 + 15u) - ((int)(lane + n) - 1);           // a lane past its last column still owes the missing shifts
#pragma unroll
            for (int k = 0; k < R; ++k) {
                uint32_t v = S.acc[k];
                if (miss > 0 && miss < 16) v <<= 2 * miss;
                field[(tb * R + k) * WAVE + lane] = v;
            }
        }
        ncell = S.cnt;
    }
    po.score = pair_max; po.flags = SWMI_F_DONE | (ncell > X.res_cell_cap ? SWMI_F_CELL_OVF : 0u); po.n_cells = ncell;
    if (lane == 0) { A.out[pd.out_id] = po; if (A.out_host) A.out_host[pd.out_id] = po; }
    if (ncell > X.res_cell_cap || ncell == 0u) return;            // (too many: the host re-runs the pair through the ordinary path)
    WAVE_SYNC();

    // ---- C: one lane per alignment ----------------------------------------------------------------------------------
    const uint8_t *ref_b = reinterpret_cast<const uint8_t *>(refc);
    const uint8_t *read_b = reinterpret_cast<const uint8_t *>(readc);
    const uint32_t umat = (uint32_t)A.match, umis = (uint32_t)A.mismatch, ugap = (uint32_t)A.gap;
    const uint32_t max_ops = 16u * X.res_ops_words;
    for (uint32_t base = 0; base < ncell; base += WAVE) {
        const bool mine = base + lane < ncell;
        const uint2 c0 = mine ? cells[base + lane] : make_uint2(0u, 0u);
        uint32_t i = c0.x, j = c0.y, score = (uint32_t)pair_max, nops = 0, cur = 0;
        int begin = 0;
        bool active = mine;
        uint32_t *my_ops = opsb + lane * X.res_ops_words;
        while (BALLOT(active)) {
            if (active) {
                const uint32_t rho = i - 1u, l = rho / R, k = rho - l * R;
                const uint32_t t = j - 1u + l;
                const uint32_t dw = field[((t >> 4) * R + k) * WAVE + l];
                const uint32_t rc = ref_b[j - 1u], qc = read_b[i - 1u];
                const uint32_t d = (dw >> (2u * (15u - (t & 15u)))) & 3u;
                const bool isA = (d & 1u) != 0u, isI = d == 2u;
                begin = (int)j;                                                  // SmithWaterman.java:383
                score -= isA ? (rc == qc ? umat : umis) : ugap;                  // :388-406, H(pred) = H - delta
                const uint32_t op = isA ? SWMI_DIR_A : (isI ? SWMI_DIR_I : SWMI_DIR_D);
                cur |= op << (2u * (nops & 15u));
                ++nops;
                if ((nops & 15u) == 0u) { if (nops <= max_ops) my_ops[(nops >> 4) - 1u] = cur; cur = 0; }
                i -= (isA || isI) ? 1u : 0u;
                j -= (isA || !isI) ? 1u : 0u;
                active = (int)score > 0 && i != 0u && j != 0u;                   // `while (score > 0)` :380
            }
        }
        if ((nops & 15u) != 0u && nops <= max_ops) my_ops[nops >> 4] = cur;
        WAVE_SYNC();
        // records: table entries + payloads (packed ops [+ strings]), contiguous for the whole wave
        const uint32_t opw = A.raw ? 0u : (nops + 15u) / 16u;                // (records with strings carry no ops)
        const uint32_t words = mine ? swmi_payload_words(nops, A.raw != nullptr) : 0u;
        const uint32_t incl = wave_scan_add_u32(words);
        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        const uint32_t nhere = ncell - base < WAVE ? ncell - base : WAVE;
        unsigned long long off;
        uint32_t rslot;
        const bool fits = swmi_reserve(A, lane, total, nhere, off, rslot);
        const bool too_long = BALLOT(mine && nops > max_ops) != 0ull;
        if (fits && !too_long) {
            if (mine) {
                uint32_t *dst = A.arena + off + (incl - words);
                swmi_write_rec(A, rslot + lane, pd.out_id, SWMI_RANK_BY_CELL, begin, c0.x, c0.y, nops, off + (incl - words));
                for (uint32_t w = 0; w < opw; ++w) dst[w] = my_ops[w];
            }
            if (A.raw) {
                // the strings (SmithWaterman.java:418-431): one alignment after the other, the whole wavefront on each
                const uint8_t *__restrict__ raw_ref = A.raw + A.raw_off[pd.ref_id];
                const uint8_t *__restrict__ raw_read = A.raw + A.raw_off[A.raw_reads_at + pd.read_id];
                for (uint32_t a = 0; a < nhere; ++a) {
                    const uint32_t na = (uint32_t)__builtin_amdgcn_readlane((int)nops, (int)a);
                    const uint32_t at = (uint32_t)__builtin_amdgcn_readlane((int)(incl - words), (int)a);
                    const uint32_t ai = (uint32_t)__builtin_amdgcn_readlane((int)c0.x, (int)a);
                    const uint32_t aj = (uint32_t)__builtin_amdgcn_readlane((int)c0.y, (int)a);
                    swmi_emit_strings(A.arena + off + at, SwmiOpsPacked{opsb + a * X.res_ops_words},
                                      na, ai, aj, raw_ref, raw_read, lane, scratch);
                }
            }
        } else if (lane == 0) {
            atomicOr(&A.out[pd.out_id].flags, SWMI_F_ARENA_OVF);
            if (A.ovf_host) *A.ovf_host = 1u;
        }
        WAVE_SYNC();
    }
}

extern "C" __global__ void __launch_bounds__(WAVE * FILL_WAVES)
sw_resident_pairs_kernel(const TraceArgs A, const ResidentArgs X) {
    extern __shared__ uint32_t rp_lds[];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t item = blockIdx.x * FILL_WAVES + wave;
    if (item >= X.n_res) return;
    // (the arena header was reset by sw_sweep_winmax_kernel, one launch earlier; the traceback kernels append after this one)
    const PairDesc pd = A.pairs[X.res_items[item]];
    uint32_t *lds = rp_lds + wave * X.res_lds_words;
    const uint32_t R = swmi_rows_per_lane(A.reads[pd.read_id].len);
    if (A.strict) {
        if (R == 1)      resident_pair<1, true>(A, X, pd, lane, lds);
        else if (R == 2) resident_pair<2, true>(A, X, pd, lane, lds);
        else if (R == 3) resident_pair<3, true>(A, X, pd, lane, lds);
        else             resident_pair<4, true>(A, X, pd, lane, lds);
    } else {
        if (R == 1)      resident_pair<1, false>(A, X, pd, lane, lds);
        else if (R == 2) resident_pair<2, false>(A, X, pd, lane, lds);
        else if (R == 3) resident_pair<3, false>(A, X, pd, lane, lds);
        else             resident_pair<4, false>(A, X, pd, lane, lds);
    }
}

extern "C" hipError_t swmi_launch_resident(const TraceArgs *a, const ResidentArgs *x, hipStream_t st) {
    if (x->n_res == 0) return hipSuccess;
    static const bool attr = [] { (void)hipFuncSetAttribute(reinterpret_cast<const void *>(sw_resident_pairs_kernel),
                                                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); return true; }();
    (void)attr;
    hipLaunchKernelGGL(sw_resident_pairs_kernel, dim3((x->n_res + FILL_WAVES - 1) / FILL_WAVES), dim3(WAVE * FILL_WAVES),
                       (size_t)FILL_WAVES * x->res_lds_words * sizeof(uint32_t), st, *a, *x);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// host-callable launchers (the runtime in swmi_api.cpp is plain C++)
// ------------------------------------------------------------------------------------------------
// Small launches: the dispatcher may stack several workgroups on one CU while other CUs stay empty (measured: 250
// workgroups of the 67-VGPR column-chunk kernel ran two to a CU, each wave sharing its SIMD, 1.5x slower per step).  A
// dynamic-LDS request nobody uses caps the workgroups a CU can hold at what an even spread needs, so the launch is dealt
// over all 256 CUs.  SWMI_LDS_SPREAD=0 switches it off.
static size_t spread_lds(uint32_t n_groups) {
    static const int on = getenv("SWMI_LDS_SPREAD") ? atoi(getenv("SWMI_LDS_SPREAD")) : 1;
    if (!on || n_groups == 0 || n_groups > 4u * 256u) return 0;
    const uint32_t per_cu = (n_groups + 255u) / 256u;                      // workgroups a CU must take
    // per_cu fit, per_cu + 1 do not -- and no more than that needs: the rest of the CU's LDS stays free for the kernels of
    // ANOTHER batch in flight on the same GPU (a sweep that reserved the whole 160 KB kept the other batch's traceback
    // workgroups off its CU: 0.130 ms per step with two batches in flight, 0.119 without the reservation)
    return ((size_t)(160u * 1024u) / (per_cu + 1u) / 1024u + 1u) * 1024u;
}
template <class K>
static void allow_big_lds(K kernel) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

// ev_start / ev_stop (both or none): the launch is ONE kernel and the events take its start and stop times from the dispatch
// itself (hipExtLaunchKernelGGL) -- no marker packets between the kernels of a run, which hipEventRecord would put there
extern "C" hipError_t swmi_launch_fill(const FillArgs *a, hipStream_t st, hipEvent_t ev_start, hipEvent_t ev_stop) {
    if (a->n_pairs == 0) return hipSuccess;
    static const bool attrs = [] {
        allow_big_lds(sw_fill_kernel); allow_big_lds(sw_fill_score_kernel); allow_big_lds(sw_sweep_winmax_kernel);
        allow_big_lds(sw_sweep_winmax_strips_kernel); allow_big_lds(sw_sweep_winmax_cols_kernel);
        return true;
    }();
    (void)attrs;
    const dim3 grid((a->n_pairs + FILL_WAVES - 1) / FILL_WAVES), block(WAVE * FILL_WAVES);
    const size_t lds = spread_lds(grid.x);
    if (a->mode == 0)      hipLaunchKernelGGL(sw_fill_kernel, grid, block, lds, st, *a);
    else if (a->mode == 1) {
        if (ev_start && ev_stop && !(a->skip_multi && a->n_strip_items) && !a->n_col_items) {
            hipExtLaunchKernelGGL(sw_sweep_winmax_kernel, grid, block, (uint32_t)lds, st, ev_start, ev_stop, 0u, *a);
            return hipGetLastError();
        }
        hipLaunchKernelGGL(sw_sweep_winmax_kernel, grid, block, a->n_col_items || a->n_strip_items ? 0 : lds, st, *a);
        if (a->skip_multi && a->n_strip_items) {
            const uint32_t g = (a->n_strip_items + FILL_WAVES - 1) / FILL_WAVES;
            hipLaunchKernelGGL(sw_sweep_winmax_strips_kernel, dim3(g), block, spread_lds(g), st, *a);
        }
        if (a->n_col_items) {
            const uint32_t g = (a->n_col_items + FILL_WAVES - 1) / FILL_WAVES;
            hipLaunchKernelGGL(sw_sweep_winmax_cols_kernel, dim3(g), block, spread_lds(g), st, *a);
        }
    }
    else                   hipLaunchKernelGGL(sw_fill_score_kernel, grid, block, lds, st, *a);
    return hipGetLastError();
}

extern "C" hipError_t swmi_launch_traceback_split(const TraceArgs *a, uint32_t n_windows, hipStream_t st) {
    if (a->n_pairs == 0) return hipSuccess;
    const size_t win = (size_t)SWMI_CK_BLOCKS * SWMI_RMAX * WAVE;
    const size_t det_words = SWMI_SPLIT_WAVES * (win + 2u * SWMI_DETECT_LCAP);
    if (n_windows)
        hipLaunchKernelGGL(sw_detect_windows_kernel, dim3((n_windows + SWMI_SPLIT_WAVES - 1) / SWMI_SPLIT_WAVES), dim3(WAVE * SWMI_SPLIT_WAVES),
                           det_words * sizeof(uint32_t), st, *a);
    const size_t per_wave = (size_t)a->lds_words + a->lds_read_words + SWMI_TB_REFWIN_WORDS + win;
    // enough wavefronts to fill the chip several times over, but no more workgroups than there can be items
    uint32_t groups = (a->q_cap + SWMI_SPLIT_WAVES - 1) / SWMI_SPLIT_WAVES;
    if (groups > 2048u) groups = 2048u;
    if (groups < 1u) groups = 1u;
    hipLaunchKernelGGL(sw_walk_items_kernel, dim3(groups), dim3(WAVE * SWMI_SPLIT_WAVES), SWMI_SPLIT_WAVES * per_wave * sizeof(uint32_t), st, *a);
    return hipGetLastError();
}

extern "C" hipError_t swmi_launch_traceback(const TraceArgs *a, hipStream_t st, hipEvent_t ev_start, hipEvent_t ev_stop) {
    if (a->n_pairs == 0) return hipSuccess;
    const size_t tile = (size_t)(a->mode == 0 ? SWMI_TB_BLOCKS : SWMI_CK_BLOCKS) * SWMI_RMAX * WAVE;
    const size_t per_wave = (size_t)a->lds_words + a->lds_read_words + SWMI_TB_REFWIN_WORDS + tile;
    const dim3 block(WAVE * FILL_WAVES);
    if (a->mode == 1) {
        // 8 waves per pair (bigger teams, shorter critical path) while every workgroup of the launch can be resident
        // at once (4 waves per SIMD at this kernel's register count), else 4.  Big batches are throughput-bound: there a
        // helper wave that mostly waits only takes a slot another pair's walker could use, so every pair gets ONE wave
        // (it lists the cells, then walks the alignments one after the other).
        static int forced = getenv("SWMI_TB_WAVES") ? atoi(getenv("SWMI_TB_WAVES")) : 0;
        static int big = getenv("SWMI_TB_BIG") ? atoi(getenv("SWMI_TB_BIG")) : 10000;   // measured crossover: between 8 k and 16 k pairs
        const uint32_t n_waves = forced ? (uint32_t)forced : (a->n_pairs <= 512u ? SWMI_TB_WAVES : a->n_pairs <= (uint32_t)big ? 4u : 1u);
        const size_t n_walkers = n_waves < SWMI_TB_SLOTS ? n_waves : SWMI_TB_SLOTS;
        const size_t tiles = (size_t)n_waves * SWMI_CK_BLOCKS * SWMI_RMAX * WAVE;
        size_t words = 32 + tiles + n_walkers * ((size_t)a->lds_words + a->lds_read_words + SWMI_TB_REFWIN_WORDS);
        // speculative staging (traceback_pair): a second set of tiles, while teams have helpers and the block still fits.
        // Measured (profiles/r03/ab_spec_staging.txt): 250 pairs 0.0554 -> 0.0491 ms, 500 pairs 0.0600 -> 0.0593, 1000 pairs
        // 0.0735 -> 0.0768 -- from about 500 pairs on the SIMDs are shared by the waves of several pairs and the helpers'
        // extra re-sweeps (a span whose walk ends early is wasted) cost other pairs' walkers more than the waits they save.
        static const int spec_opt = getenv("SWMI_TB_SPEC") ? atoi(getenv("SWMI_TB_SPEC")) : 1;      // 0 never, 1 automatic, 2 always
        TraceArgs t = *a;
        t.pad2 = (spec_opt && (spec_opt == 2 || a->n_pairs <= 384u) && n_waves > 1u && (words + tiles) * sizeof(uint32_t) <= 160u * 1024u) ? 1u : 0u;
        if (t.pad2) words += tiles;
        if (ev_start && ev_stop)
            hipExtLaunchKernelGGL(sw_traceback_winmax_kernel, dim3(a->n_pairs), dim3(WAVE * n_waves), (uint32_t)(words * sizeof(uint32_t)), st, ev_start, ev_stop, 0u, t);
        else
            hipLaunchKernelGGL(sw_traceback_winmax_kernel, dim3(a->n_pairs), dim3(WAVE * n_waves), words * sizeof(uint32_t), st, t);
    } else {
        const dim3 grid((a->n_pairs + FILL_WAVES - 1) / FILL_WAVES, SWMI_TB_SLOTS);
        if (a->mode == 0) hipLaunchKernelGGL(sw_traceback_kernel, grid, block, per_wave * FILL_WAVES * sizeof(uint32_t), st, *a);
        else              hipLaunchKernelGGL(sw_traceback_replay_kernel, grid, block, per_wave * FILL_WAVES * sizeof(uint32_t), st, *a);
    }
    return hipGetLastError();
}
