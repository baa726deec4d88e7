// swmi_tfused.hip -- gfx950 kernel: sweep AND traceback of a pair in ONE launch, in the TRANSPOSED layout (option "tfused").
//
// Same results as the sweep + traceback kernels of swmi_kernels.hip (ScoreMatrix.call src/sw/SmithWaterman.java:129-190,
// GetCellScore.call :217-252 / DistributedSW.java:305-330, GetAlignment.call :354-436) for the usual pair: both sequences of
// fast symbols, int4 scores, gap < 0, a read of at most 256 bases, a reference of at most 64 * SWMI_TF_BMAX bases.
// Measurements and the decision to leave it opt-in: DESIGN.md 4.4.
//
// Why transposed.  The sweep of swmi_kernels.hip puts the READ's rows on the lanes (R = 3 rows per lane at 150 bp) and
// streams the reference through: n + 63 steps of 3 R cell instructions + ~8 of per-step overhead (neighbour exchange, symbol
// feed, window maximum, checkpoints).  At 150 x 2000 that is 2050 x 17 = 34.8 k instructions per pair, half of them overhead,
// and one wavefront per SIMD issues one instruction per ~5 cycles whatever it is (tools/ubench_occ.hip): what counts is the
// count.  Here the lanes own the REFERENCE's columns, B = ceil(n / 64) consecutive columns each (B = 32 at 2 kbp), and the
// read's rows stream through: m + 62 steps of 3.5 B + ~20 instructions = 212 x 133 = 28.2 k.  The per-step overhead is paid
// 212 times instead of 2050 times; what the pipeline fill costs more (62 of 212 steps instead of 50 of 2050) is less.
//
//   A  sweep: lane l, step t, row i = t - l, columns B*l .. B*l+B-1.  Per cell v_dot8_i32_i4 (one-hot read symbol . the
//      column's 8 x int4 score profile + the diagonal), v_max3_i32, v_sub_u32 clamp (hp = max(H + gap, 0)), and half a
//      v_max3 for the lane's running maximum.  N comes from the lane's own registers, W from the cell before in the same
//      step, NW/W of a lane's first column from lane l-1 by DPP.  The read symbol moves down the lanes by one DPP shift per
//      step.  Every step each lane stores the H of its LAST column: 256 coalesced bytes -- column checkpoints
//      ck[t][l] = H(t - l, B*(l+1) - 1), (m + 62) x 256 B per pair (54 KB at 150 bp, against 83 KB of lane-state checkpoints).
//      Lanes that have not started compute zeros from zeros, rows past the read's end and padding columns see a zero one-hot /
//      zero profile and cannot exceed the true maximum: nothing is masked but the lanes' maxima in the last L - 1 steps.
//   B  the lanes whose maximum equals the pair's name the candidate stripes.  A BLOCK of 64 * TF_BR (256) columns, right-aligned
//      on a candidate stripe, left edge on a checkpointed column, is re-swept with TF_BR columns per lane, the left column fed
//      from the checkpoints, rows streaming from a zero top row: scores are kept x 4 with the move in the two low bits (alignment
//      2 > insertion 1 > deletion 0, or the reverse for the strict mode), so that ONE v_max3 yields score and direction with the
//      reference's tie order; the two bits are shifted into a direction word per column (v_alignbit) and every 16 steps a lane
//      stores its words to LDS.  Cells of the stripe equal to the maximum are listed on the way.
//   C  an alignment is walked by a whole wavefront, a run of "alignment" moves per iteration (tf_walk); a walk that leaves its
//      block re-sweeps the block on its left.  Its record goes straight to the arena (format of swmi_device.h: AlnRec).
// The four sweeping wavefronts of a workgroup and two helpers share B and C through two queues in LDS: block tasks and walk
// items (below).  Nothing overflows: a block with more tied cells than its list holds is taken in passes.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "swmi_device.h"
#include "swmi_emit.h"

#define WAVE 64
#define TF_WAVES 4                       // wavefronts of a workgroup that sweep a pair each
#define TF_HELPERS SWMI_TF_HELPERS       // ... and wavefronts that only take tasks (at most)
#define TF_BR SWMI_TF_BR                 // columns per lane of a re-swept block
#define TF_BW (64u * TF_BR)              // ... and its width
#define TF_ACC (TF_BR >= 5u ? 2u : 1u)   // candidate stripes a block takes cells from, counted from its right edge
#define BALLOT(pred) __builtin_amdgcn_ballot_w64(pred)
#define WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)

namespace {

__device__ __forceinline__ int tf_shr1(int old, int src) { return __builtin_amdgcn_update_dpp(old, src, 0x138 /*wave_shr:1*/, 0xf, 0xf, false); }
__device__ __forceinline__ int tf_shr1_zero(int src) { return __builtin_amdgcn_update_dpp(0, src, 0x138, 0xf, 0xf, true); }
__device__ __forceinline__ int tf_ror1(int src) { return __builtin_amdgcn_update_dpp(src, src, 0x13C /*wave_ror:1*/, 0xf, 0xf, false); }
__device__ __forceinline__ int tf_max3(int a, int b, int c) { const int t = a > b ? a : b; return t > c ? t : c; }
__device__ __forceinline__ int tf_subsat(int a, uint32_t b) { return (int)__builtin_elementwise_sub_sat((uint32_t)a, b); }

__device__ __forceinline__ int tf_wave_max(int v) {
#define TF_DPP_MAX(ctrl, rmask) { int o_ = __builtin_amdgcn_update_dpp(v, v, ctrl, rmask, 0xf, false); v = v > o_ ? v : o_; }
    TF_DPP_MAX(0x111, 0xf) TF_DPP_MAX(0x112, 0xf) TF_DPP_MAX(0x114, 0xf) TF_DPP_MAX(0x118, 0xf) TF_DPP_MAX(0x142, 0xa) TF_DPP_MAX(0x143, 0xc)
#undef TF_DPP_MAX
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ uint32_t tf_scan_add(uint32_t v) {          // inclusive prefix sum over the 64 lanes
#define TF_DPP_ADD(ctrl, rmask) { const uint32_t o_ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, rmask, 0xf, true); v += o_; }
    TF_DPP_ADD(0x111, 0xf) TF_DPP_ADD(0x112, 0xf) TF_DPP_ADD(0x114, 0xf) TF_DPP_ADD(0x118, 0xf) TF_DPP_ADD(0x142, 0xa) TF_DPP_ADD(0x143, 0xc)
#undef TF_DPP_ADD
    return v;
}
__device__ __forceinline__ uint32_t tf_lanes_below(uint64_t mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}
__device__ __forceinline__ uint32_t tf_ld_l2(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint32_t tf_uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

// 8 x int4 score profile of one reference column: nibble c/4 = score of the column's base against read symbol c
__device__ __forceinline__ int tf_profile(uint32_t code, bool inside, int match, int mismatch) {
    uint32_t p = (uint32_t)(mismatch & 0xF) * 0x11111111u;
    const uint32_t sh = code & 28u;
    p = (p & ~(0xFu << sh)) | ((uint32_t)(match & 0xF) << sh);
    return inside ? (int)p : 0;
}

struct TfPair {
    uint32_t n, m, B, L;             // reference / read length, columns per lane of the sweep, lanes that hold columns
    const uint8_t *ref_b, *read_b;   // LDS copies of the byte images
    const uint8_t *raw_ref, *raw_read;   // the caller's own bytes (for the aligned strings), or null
    uint32_t *ck;                    // column checkpoints [step][lane]
    int match, mismatch;
    uint32_t g;                      // -gap > 0
};

// ---- A: the score sweep --------------------------------------------------------------------------------------------
template <int B>
__device__ __forceinline__ int tf_sweep(const TfPair &P, const uint32_t lane, int &lane_max) {
    int H[B], hp[B], q[B];
#pragma unroll
    for (int k = 0; k < B; ++k) {
        const uint32_t c = (uint32_t)B * lane + (uint32_t)k;
        q[k] = tf_profile(P.ref_b[c < P.n ? c : 0u], c < P.n, P.match, P.mismatch);
        H[k] = 0; hp[k] = 0;
    }
    int nwL = 0, M = 0, oh = 0;
    const uint32_t T = tf_uni(P.m + P.L - 1u);
    uint32_t *__restrict__ ckrow = P.ck;                                  // (wave-uniform: the store takes it as its scalar base)
    for (uint32_t t0 = 0; t0 < T; t0 += 64u) {
        const uint32_t idx = t0 + 63u - lane;
        int Q = idx < P.m ? (int)(1u << (P.read_b[idx] & 28u)) : 0;       // lane 63 holds the symbol of step t0, lane 62 of t0 + 1, ...
        const uint32_t tend = T - t0 < 64u ? T - t0 : 64u;
        for (uint32_t r = 0; r < tend; ++r, ckrow += WAVE) {
            Q = tf_ror1(Q);
            oh = tf_shr1(Q, oh);                                       // the symbol moves one lane down; lane 0 takes the next one
            const int nw_next = tf_shr1_zero(H[B - 1]);                // lane l-1 finished this row one step ago: W's H now, NW next step
            const int hpL = tf_shr1_zero(hp[B - 1]);
#pragma unroll
            for (int k = B - 1; k >= 1; --k) H[k] = __builtin_amdgcn_sdot8(oh, q[k], H[k - 1], true);    // NW + s, in place
            H[0] = __builtin_amdgcn_sdot8(oh, q[0], nwL, true);
            nwL = nw_next;
            int w = hpL;
#pragma unroll
            for (int k = 0; k < B; ++k) {
                const int x = tf_max3(H[k], hp[k], w);                  // SmithWaterman.java:223-249 with hp = max(H + gap, 0)
                H[k] = x;
                w = tf_subsat(x, P.g);
                hp[k] = w;
            }
            int Mn = M;
#pragma unroll
            for (int k = 0; k + 1 < B; k += 2) Mn = tf_max3(Mn, H[k], H[k + 1]);
            if (B & 1) Mn = Mn > H[B - 1] ? Mn : H[B - 1];
            // rows past the read's end copy the row above diagonally (zero one-hot): they must not carry the maximum into
            // the stripes to the right, which would then be re-swept for nothing.  Only the last L - 1 steps have such rows.
            if (t0 + r < P.m) M = Mn;
            else              M = (t0 + r - lane < P.m) ? Mn : M;
            ckrow[lane] = (uint32_t)H[B - 1];
        }
    }
    lane_max = M;
    return tf_wave_max(M);
}

// ---- B: one block of TF_BW columns with directions ---------------------------------------------------------------------
// Block = columns c_lo .. c_lo + TF_BW - 1, c_lo = B*k.  Lane la owns columns c_lo + TF_BR la + kk.  tile[((t/16)*TF_BR + kk)*64 + la]
// holds the 16 steps t = 16 q .. 16 q + 15 of column kk of lane la, step t at bits 2*(t%16); step t of lane la is row t - la.
// Cells equal to `pmax` inside columns [acc_lo, acc_hi) are counted, and those numbered base .. base + cell_cap - 1 (in the order
// they are met) listed in cells[] (1-based i, j); returns the count.
struct TfReplayState {
    int Hs[TF_BR], hpN[TF_BR];
    uint32_t dw[TF_BR];
    int nwL, wlast, oh, Q, Bq, gmax;
};

// 16 steps.  LIST = false: the lane's maximum over the group is kept (2 instructions per step); LIST = true: every step tests
// its four cells and appends the hits -- run only for a group whose maximum reached the pair's, from the saved state.
template <bool STRICT, bool LIST>
__device__ __forceinline__ uint32_t tf_replay_group(TfReplayState &S, const int (&q)[TF_BR], const uint32_t subW, const uint32_t subN,
                                                    const uint32_t tg, const uint32_t lane, const TfPair &P, const uint32_t c_lo,
                                                    const int target, const uint32_t acc_lo, const uint32_t acc_hi,
                                                    uint2 *__restrict__ cells, const uint32_t cell_cap, const uint32_t base, uint32_t cnt) {
    constexpr int TAGH = STRICT ? 0 : 2;
#pragma unroll
    for (uint32_t rr = 0; rr < 16u; ++rr) {
        S.Q = tf_ror1(S.Q);
        S.oh = tf_shr1(S.Q, S.oh);
        S.Bq = tf_ror1(S.Bq);
        const int nw_next = tf_shr1(S.Bq, S.Hs[TF_BR - 1]);                  // lane 0: the boundary column
        const int hpL = tf_shr1(tf_subsat(S.Bq, subW), S.wlast);
        int a[TF_BR];
#pragma unroll
        for (int kk = TF_BR - 1; kk >= 1; --kk) a[kk] = __builtin_amdgcn_sdot8(S.oh, q[kk], S.Hs[kk - 1], true);
        a[0] = __builtin_amdgcn_sdot8(S.oh, q[0], S.nwL, true);
        S.nwL = nw_next;
        int w = hpL;
#pragma unroll
        for (int kk = 0; kk < TF_BR; ++kk) {
            const int x = tf_max3(a[kk], S.hpN[kk], w);              // the low two bits name the winner, ties by the reference's order
            S.dw[kk] = __builtin_amdgcn_alignbit((uint32_t)x, S.dw[kk], 2u);
            S.Hs[kk] = (x & ~3) | TAGH;
            w = tf_subsat(S.Hs[kk], subW);
            S.hpN[kk] = tf_subsat(S.Hs[kk], subN);
        }
        S.wlast = w;
        if (!LIST) {
#pragma unroll
            for (int kk = 0; kk + 1 < TF_BR; kk += 2) S.gmax = tf_max3(S.gmax, S.Hs[kk], S.Hs[kk + 1]);
            if (TF_BR & 1) S.gmax = S.gmax > S.Hs[TF_BR - 1] ? S.gmax : S.Hs[TF_BR - 1];
        } else {
            const int row = (int)(tg + rr) - (int)lane;
#pragma unroll
            for (int kk = 0; kk < TF_BR; ++kk) {
                const uint32_t c = c_lo + TF_BR * lane + (uint32_t)kk;
                const bool ok = S.Hs[kk] == target && row >= 0 && row < (int)P.m && c < P.n && c >= acc_lo && c < acc_hi;
                const uint64_t mk = BALLOT(ok);
                if (mk) {
                    const uint32_t pos = cnt + tf_lanes_below(mk);
                    if (ok && pos >= base && pos - base < cell_cap) cells[pos - base] = make_uint2((uint32_t)row + 1u, c + 1u);
                    cnt += (uint32_t)__builtin_popcountll(mk);
                }
            }
        }
    }
    return cnt;
}

template <bool STRICT>
__device__ __forceinline__ uint32_t tf_replay(const TfPair &P, const uint32_t lane, const uint32_t k_stripe, uint32_t *__restrict__ tile,
                                              const bool detect, const int pmax, const uint32_t acc_lo, const uint32_t acc_hi,
                                              uint2 *__restrict__ cells, const uint32_t cell_cap, const uint32_t base) {
    constexpr int TAGH = STRICT ? 0 : 2;
    uint32_t cnt = 0;
    const uint32_t c_lo = P.B * k_stripe;
    const uint32_t subW = 4u * P.g + (STRICT ? (uint32_t)-2 : 2u), subN = 4u * P.g + (STRICT ? (uint32_t)-1 : 1u);
    const int target = 4 * pmax + TAGH;
    int q[TF_BR];
    TfReplayState S;
#pragma unroll
    for (int kk = 0; kk < TF_BR; ++kk) {
        const uint32_t c = c_lo + TF_BR * lane + (uint32_t)kk;
        q[kk] = tf_profile(P.ref_b[c < P.n ? c : 0u], c < P.n, P.match, P.mismatch);
        S.Hs[kk] = TAGH; S.hpN[kk] = 0; S.dw[kk] = 0;
    }
    S.nwL = TAGH; S.wlast = 0; S.oh = 0; S.gmax = 0;
    const uint32_t T = (P.m + 63u + 15u) & ~15u;
    const uint32_t *__restrict__ ckcol = P.ck + (k_stripe ? k_stripe - 1u : 0u);
    for (uint32_t t0 = 0; t0 < T; t0 += 64u) {
        const uint32_t idx = t0 + 63u - lane;
        S.Q = idx < P.m ? (int)(4u << (P.read_b[idx] & 28u)) : 0;        // one-hot nibble 4: the dot product yields 4 s
        S.Bq = TAGH;                                                     // left boundary column, H(i, c_lo - 1) in stored form
        if (k_stripe && idx < P.m) S.Bq = (int)(4u * tf_ld_l2(ckcol + (size_t)(idx + k_stripe - 1u) * WAVE)) + TAGH;
        const uint32_t tend = T - t0 < 64u ? T - t0 : 64u;               // (a multiple of 16)
        for (uint32_t r0 = 0; r0 < tend; r0 += 16u) {
            const TfReplayState S0 = S;
            S.gmax = 0;
            cnt = tf_replay_group<STRICT, false>(S, q, subW, subN, t0 + r0, lane, P, c_lo, target, acc_lo, acc_hi, cells, cell_cap, base, cnt);
            if (detect && BALLOT(S.gmax == target)) {                    // rare: a maximum cell in this group -- once more, listing
                S = S0;
                cnt = tf_replay_group<STRICT, true>(S, q, subW, subN, t0 + r0, lane, P, c_lo, target, acc_lo, acc_hi, cells, cell_cap, base, cnt);
            }
#pragma unroll
            for (int kk = 0; kk < TF_BR; ++kk) tile[(((t0 + r0) >> 4) * TF_BR + (uint32_t)kk) * WAVE + lane] = S.dw[kk];
        }
    }
    return cnt;
}

__device__ __forceinline__ int tf_sweep_dispatch(const TfPair &P, uint32_t lane, int &lm) {
    switch (P.B) {
#define TF_CASE(b) case b: return tf_sweep<b>(P, lane, lm);
        TF_CASE(2) TF_CASE(4) TF_CASE(6) TF_CASE(8) TF_CASE(10) TF_CASE(12) TF_CASE(14) TF_CASE(16) TF_CASE(18) TF_CASE(20)
        TF_CASE(22) TF_CASE(24) TF_CASE(26) TF_CASE(28) TF_CASE(30) TF_CASE(32) TF_CASE(34) TF_CASE(36) TF_CASE(38) TF_CASE(40)
#undef TF_CASE
    }
    lm = 0;
    return 0;
}

// ---- workgroup-shared state (LDS) ------------------------------------------------------------------------------------
// The first 4 wavefronts of a workgroup sweep 4 pairs, one each, and go on with the rightmost candidate block of their pair.
// Everything else comes in TASKS any wavefront of the workgroup may take -- the sweepers once their own block is through, and up
// to two helper wavefronts that do nothing else:
//   block task: one more candidate block of a pair -- re-sweep it into the taker's tile, list its maximum cells, walk the first
//               alignment; a sweeper hands the other alignments out as walk items, a helper walks them itself;
//   walk item:  one alignment, walked through the tile of the wavefront that listed it (LDS is shared), into a record.
// With one wavefront per SIMD the launch lasts as long as its slowest pair: a pair with tied maxima in several blocks, or
// three alignments to walk, must not be the work of one wavefront.
#define TF_QCAP 96u
struct TfSlot {                      // one per sweeper = per pair of the workgroup
    uint32_t n, m, out_id, pmax;
    uint32_t ck_lo, ck_hi;           // the pair's column checkpoints
    uint32_t tasks_total, tasks_done, cells;
    uint32_t ref_id, read_id;        // (for the caller's own bytes: the aligned strings)
    uint32_t pad[5];
};
struct TfQueue {
    uint32_t n, taken, lock, pad;
    uint4 e[TF_QCAP];
};
struct TfShared {
    uint32_t owners_done, pad[3];
    uint32_t tile_users[8];          // walk items still reading this wavefront's tile
    TfQueue qb;                      // block tasks {slot, first stripe of the block, accepted columns lo, hi}
    TfQueue qw;                      // walk items {slot | tile owner << 8, first column of the tile, i, j}
    TfSlot slot[TF_WAVES];
};
// Who waits for whom: a wavefront waits only for the walk items that read ITS tile, before it overwrites the tile.  Helpers hand
// no walk items out, so nobody reads their tiles and they never wait: every walk item is taken and finished in the end, and
// every wait of a sweeper ends.  Every wait is bounded all the same (a give-up raises a host-visible error code).


__device__ __forceinline__ uint32_t tf_lds_add(uint32_t *p, uint32_t v) {
    return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ uint32_t tf_lds_load(const uint32_t *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// lane 0 appends to a queue of the workgroup: lock (returns the count), write the entries behind it, publish the new count
__device__ __forceinline__ uint32_t tf_queue_lock(const TraceArgs &A, TfQueue *q) {
    for (uint32_t spins = 0; __hip_atomic_exchange(&q->lock, 1u, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != 0u; ++spins) {
        if (spins > (1u << 22)) { if (A.ovf_host) A.ovf_host[1] = 0xDEAD0001u; break; }          // (never seen: a wavefront must not hang)
        __builtin_amdgcn_s_sleep(1);
    }
    return tf_lds_load(&q->n);
}
__device__ __forceinline__ void tf_queue_publish(TfQueue *q, const uint32_t new_n) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __hip_atomic_store(&q->n, new_n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);      // (the count moves after the entries are written)
    __hip_atomic_store(&q->lock, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// The region of a wavefront (dwords): tile | cells [2 cap] | staged ops of one alignment, one per byte | reference codes | read codes
struct TfRegion {
    uint32_t *tile;
    uint2 *cells;
    uint32_t *stage;
};
__device__ __forceinline__ TfRegion tf_region(const TFusedArgs &X, uint32_t *lds) {
    TfRegion R;
    R.tile = lds;
    R.cells = reinterpret_cast<uint2 *>(lds + X.tile_words);
    R.stage = lds + X.tile_words + 2u * X.cell_cap;
    return R;
}

// takes the next entry of a queue if there is one (lane 0 decides, all lanes get the index; 0xFFFFFFFF: none)
__device__ __forceinline__ uint32_t tf_queue_try_take(TfQueue *q, const uint32_t lane) {
    uint32_t idx = 0xFFFFFFFFu;
    if (lane == 0) {
        for (;;) {
            uint32_t t = tf_lds_load(&q->taken);
            if (t >= tf_lds_load(&q->n)) break;
            if (__hip_atomic_compare_exchange_strong(&q->taken, &t, t + 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) { idx = t; break; }
        }
    }
    return tf_uni(idx);
}

// before a wavefront overwrites its own tile: the walk items reading it must be through
__device__ __forceinline__ void tf_wait_tile(const TraceArgs &A, TfShared *sh, const uint32_t me) {
    for (uint32_t spins = 0; tf_lds_load(&sh->tile_users[me]) != 0u; ++spins) {
        if (spins > (1u << 22)) { if (A.ovf_host) A.ovf_host[1] = 0xDEAD0003u; break; }
        __builtin_amdgcn_s_sleep(4);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// One alignment, the whole wavefront on it: lane k looks at cell (i - k, j - k), the diagonal up-left of the current cell.
// The leading lanes whose cell says "alignment" are one run of the path; a prefix sum of their score deltas finds where
// `while (score > 0)` (SmithWaterman.java:380) ends it.  The lane behind the run holds the gap move that follows: one
// iteration per gap of the alignment instead of one per step.  The tile read is `tile_w` (columns t_lo ..), the walker's own
// or the one of the wavefront that listed the cell (`foreign`: given back as soon as the walk leaves it); a walk that leaves
// its block on the left re-sweeps the block it needs into the walker's own tile.  The record goes straight to the arena.
template <bool STRICT>
__device__ __forceinline__ bool tf_walk(const TraceArgs &A, const TFusedArgs &X, TfShared *sh, const TfPair &P, const uint32_t out_id,
                                        const int pmax, const uint32_t ci, const uint32_t cj, const uint32_t *tile_w, uint32_t t_lo,
                                        int foreign, const uint32_t me, const uint32_t lane, const TfRegion &R) {
    const uint32_t umat = (uint32_t)A.match, umis = (uint32_t)A.mismatch, ugap = (uint32_t)A.gap;
    uint8_t *stage_b = reinterpret_cast<uint8_t *>(R.stage);
    const uint32_t stage_cap = 4u * (X.stage_words - SWMI_EMIT_SCRATCH_WORDS);   // (the tail of the staging area: scratch of swmi_emit_strings)
    uint32_t i = ci, j = cj, score = (uint32_t)pmax, nops = 0;
    int begin = 0;
    bool bad = false, moved = false;
    for (uint32_t guard = 0;; ++guard) {
        if (guard > 8192u) { bad = true; break; }                        // (a corrupted workspace must not hang the wavefront)
        if (j - 1u < t_lo || j - 1u >= t_lo + TF_BW) {                    // the block that holds column j - 1, right-aligned on it
            const uint32_t k2 = j > TF_BW ? (j - TF_BW + P.B - 1u) / P.B : 0u;
            if (foreign >= 0) {
                if (lane == 0) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); tf_lds_add(&sh->tile_users[foreign], 0xFFFFFFFFu); }
                foreign = -1;
            }
            tf_wait_tile(A, sh, me);
            WAVE_SYNC();
            (void)tf_replay<STRICT>(P, lane, k2, R.tile, false, pmax, 0u, 0u, R.cells, 0u, 0u);
            WAVE_SYNC();
            tile_w = R.tile;
            t_lo = P.B * k2;
            moved = true;
        }
        const bool valid = lane < i && lane < j && j - 1u - lane >= t_lo;
        const uint32_t col = valid ? j - 1u - lane : t_lo, row = valid ? i - 1u - lane : 0u;
        const uint32_t c = col - t_lo, la = c / TF_BR, kk = c - la * TF_BR, t = row + la;
        uint32_t dwv = tile_w[((t >> 4) * TF_BR + kk) * WAVE + la];
        uint32_t rc = P.ref_b[col], qc = P.read_b[row];
        asm volatile("" : "+v"(dwv), "+v"(rc), "+v"(qc));              // (the three LDS reads in flight together: one wait, not two)
        const uint32_t tag = (dwv >> (2u * (t & 15u))) & 3u;
        const uint32_t op = STRICT ? 2u - tag : tag;                     // SWMI_DIR_D 0, SWMI_DIR_I 1, SWMI_DIR_A 2
        const bool isA = valid && op == SWMI_DIR_A;
        const uint32_t dlt = rc == qc ? umat : umis;                     // SmithWaterman.java:388-406, H(pred) = H - delta
        const uint32_t cum = tf_scan_add(isA ? dlt : 0u);
        const uint64_t runm = ~BALLOT(isA);
        const uint32_t r = runm ? (uint32_t)__builtin_ctzll(runm) : 64u;
        const uint64_t inrun = r >= 64u ? ~0ull : ((1ull << r) - 1ull);
        const uint64_t stopm = BALLOT((int)(score - cum) <= 0) & inrun;
        const uint32_t r_eff = stopm ? (uint32_t)__builtin_ctzll(stopm) + 1u : r;
        if (lane < r_eff && nops + lane < stage_cap) stage_b[nops + lane] = (uint8_t)SWMI_DIR_A;
        if (r_eff) {
            score -= (uint32_t)__builtin_amdgcn_readlane((int)cum, (int)(r_eff - 1u));
            nops += r_eff; i -= r_eff; j -= r_eff;
            begin = (int)(j + 1u);                                       // SmithWaterman.java:383: the column of the last visited cell
        }
        if (stopm || (int)score <= 0 || i == 0u || j == 0u) break;       // `while (score > 0)` :380
        if (r < 64u && __builtin_amdgcn_readlane((int)valid, (int)r)) {  // the gap move behind the run
            const uint32_t opg = (uint32_t)__builtin_amdgcn_readlane((int)op, (int)r);
            if (lane == 0 && nops < stage_cap) stage_b[nops] = (uint8_t)opg;
            ++nops;
            begin = (int)j;
            score -= ugap;
            if (opg == SWMI_DIR_I) --i; else --j;
            if ((int)score <= 0 || i == 0u || j == 0u) break;
        }
    }
    if (foreign >= 0 && lane == 0) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); tf_lds_add(&sh->tile_users[foreign], 0xFFFFFFFFu); }
    WAVE_SYNC();
    // the record: a table entry + the payload -- the staged ops (one per byte) packed 16 per dword [+ the two strings]
    const uint32_t opw = A.raw ? 0u : (nops + 15u) / 16u, words = swmi_payload_words(nops, A.raw != nullptr);      // (strings, or ops)
    unsigned long long off;
    uint32_t rslot;
    if (swmi_reserve(A, lane, words, 1u, off, rslot) && !bad && nops <= stage_cap) {
        uint32_t *dst = A.arena + off;
        if (lane == 0) swmi_write_rec(A, rslot, out_id, SWMI_RANK_BY_CELL, begin, ci, cj, nops, off);
        for (uint32_t w = lane; w < opw; w += WAVE) {
            uint32_t v = 0;
#pragma unroll
            for (uint32_t x = 0; x < 4u; ++x) {
                const uint32_t by = R.stage[4u * w + x];
                v |= ((by & 3u) | ((by >> 6) & 0xCu) | ((by >> 12) & 0x30u) | ((by >> 18) & 0xC0u)) << (8u * x);
            }
            const uint32_t rem = nops - 16u * w;
            if (rem < 16u) v &= (1u << (2u * rem)) - 1u;
            dst[w] = v;
        }
        if (A.raw)                                                        // the two strings GetAlignment returns (SmithWaterman.java:418-431)
            swmi_emit_strings(dst, SwmiOpsPerByte{stage_b}, nops, ci, cj, P.raw_ref, P.raw_read, lane,
                              R.stage + (X.stage_words - SWMI_EMIT_SCRATCH_WORDS));
    } else if (lane == 0) {
        atomicOr(&A.out[out_id].flags, SWMI_F_ARENA_OVF);
        if (A.ovf_host) *A.ovf_host = 1u;
    }
    WAVE_SYNC();
    return moved;
}

// a task of pair `slot` is through: whoever finishes the pair's last task writes its output
__device__ __forceinline__ void tf_task_done(const TraceArgs &A, TfSlot *slot, const uint32_t cells, const uint32_t lane) {
    if (lane != 0) return;
    if (cells) tf_lds_add(&slot->cells, cells);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    const uint32_t done = tf_lds_add(&slot->tasks_done, 1u) + 1u;
    if (done == tf_lds_load(&slot->tasks_total)) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const uint32_t out_id = slot->out_id;
        PairOut po;
        po.score = (int)slot->pmax; po.n_cells = tf_lds_load(&slot->cells);
        po.flags = SWMI_F_DONE | (atomicOr(&A.out[out_id].flags, 0u) & SWMI_F_ARENA_OVF);
        A.out[out_id].score = po.score; A.out[out_id].n_cells = po.n_cells; atomicOr(&A.out[out_id].flags, SWMI_F_DONE);
        if (A.out_host) A.out_host[out_id] = po;
    }
}

// One block task, by wavefront `me` of the workgroup, in its own LDS region; the pair's codes are in the region of the
// wavefront that swept it.
template <bool STRICT>
__device__ __forceinline__ void tf_block_task(const TraceArgs &A, const TFusedArgs &X, TfShared *sh, const uint32_t slot_id, const TfPair &P,
                                              const uint32_t kd, const uint32_t acc_lo, const uint32_t acc_hi,
                                              const uint32_t me, const uint32_t lane, const TfRegion &R, const bool allow_push) {
    TfSlot *slot = &sh->slot[slot_id];
    const int pmax = (int)slot->pmax;
    const uint32_t out_id = slot->out_id;
    const uint32_t cap = X.cell_cap;
    uint32_t total = 0;
    // a block with more maximum cells than the list holds is taken `cap` cells at a time (each pass re-sweeps it)
    for (uint32_t base = 0;; base += cap) {
        tf_wait_tile(A, sh, me);
        WAVE_SYNC();
        const uint32_t found = tf_replay<STRICT>(P, lane, kd, R.tile, true, pmax, acc_lo, acc_hi, R.cells, cap, base);
        WAVE_SYNC();
        total = found;
        const uint32_t here = found > base ? (found - base < cap ? found - base : cap) : 0u;
        // the alignments after the first go to the walk queue: their walkers read this tile while this wavefront walks the first
        uint32_t pushed = 0;
        if (here > 1u && allow_push) {
            if (lane == 0) {
                uint32_t at = tf_queue_lock(A, &sh->qw);
                for (; pushed + 1u < here && at < TF_QCAP; ++pushed, ++at) {
                    const uint2 c = R.cells[here - 1u - pushed];          // (the last `pushed` cells of the list)
                    sh->qw.e[at] = make_uint4(slot_id | (me << 8), P.B * kd, c.x, c.y);
                }
                tf_lds_add(&slot->tasks_total, pushed);                   // (before anybody can take, let alone finish, one of them)
                tf_lds_add(&sh->tile_users[me], pushed);
                tf_queue_publish(&sh->qw, at);
            }
            pushed = tf_uni(pushed);
        }
        for (uint32_t a = 0; a + pushed < here; ++a) {
            const uint2 c0 = R.cells[a];
            const bool moved = tf_walk<STRICT>(A, X, sh, P, out_id, pmax, tf_uni(c0.x), tf_uni(c0.y), R.tile, P.B * kd, -1, me, lane, R);
            if (moved && a + 1u + pushed < here) {                         // the walk left the block and re-swept another one into this tile
                WAVE_SYNC();                                               // (no walk item can be reading it: the walk waited for them)
                (void)tf_replay<STRICT>(P, lane, kd, R.tile, false, pmax, 0u, 0u, R.cells, 0u, 0u);
                WAVE_SYNC();
            }
        }
        if (found <= base + cap) break;
    }
    tf_task_done(A, slot, total, lane);
}

template <bool STRICT>
__device__ __forceinline__ void tf_workgroup(const TraceArgs &A, const TFusedArgs &X, const uint32_t wave, const uint32_t lane,
                                             uint32_t *__restrict__ lds_all) {
    TfShared *sh = reinterpret_cast<TfShared *>(lds_all);
    uint32_t *regions = lds_all + (sizeof(TfShared) + 3u) / 4u;
    uint32_t *lds = regions + wave * X.lds_words;
    const uint32_t region_codes = X.tile_words + 2u * X.cell_cap + X.stage_words;   // reference codes | read codes
    for (uint32_t w = threadIdx.x; w < sizeof(TfShared) / 4u; w += blockDim.x) lds_all[w] = 0u;
    __syncthreads();
#define TF_MARK(v) do { if (X.debug_marks && A.ovf_host && lane == 0 && blockIdx.x == 0) *(volatile uint32_t *)&A.ovf_host[2u + wave] = (v); } while (0)
    TF_MARK(0x100u);
    if (wave >= TF_WAVES + X.n_helpers) return;                           // (no LDS region for this helper)
    const uint32_t item = wave < TF_WAVES ? blockIdx.x * TF_WAVES + wave : 0xFFFFFFFFu;
    TfSlot *myslot = &sh->slot[wave < TF_WAVES ? wave : 0u];
    const uint32_t g = (uint32_t)(-(int64_t)A.gap);
    const TfRegion R = tf_region(X, lds);
    const unsigned long long tk_start = A.dbg ? __builtin_amdgcn_s_memtime() : 0ull;       // diagnostics (SWMI_DEBUG_FILL)
    unsigned long long tk_sweep = 0, tk_pro = 0, tk_task0 = 0, tk_idle = 0, n_taken = 0;
    uint32_t dbg_id = 0xFFFFFFFFu;
    bool counted = false;                                                  // this sweeper has announced that its tasks are in the queue

    // ---- A: this wavefront's own pair -------------------------------------------------------------------------------------
    if (item < X.n_items) {
        const PairDesc pd = A.pairs[X.items[item]];
        const SeqDesc rd = A.refs[pd.ref_id];
        const SeqDesc qd = A.reads[pd.read_id];
        const uint32_t n = rd.len, m = qd.len;
        const uint32_t *__restrict__ refw = A.seqw + rd.boff;
        const uint32_t *__restrict__ readw = A.seqw + qd.boff;
        uint32_t *refc = lds + region_codes;
        uint32_t *readc = refc + X.ref_words;
        for (uint32_t w = lane; w < (n + 3u) / 4u; w += WAVE) refc[w] = refw[w];
        for (uint32_t w = lane; w < (m + 3u) / 4u; w += WAVE) readc[w] = readw[w];
        WAVE_SYNC();
        dbg_id = pd.out_id;
        if (A.dbg) tk_pro = __builtin_amdgcn_s_memtime() - tk_start;
        TfPair P;
        P.n = n; P.m = m;
        P.B = swmi_tf_cols_per_lane(n);
        P.L = (n + P.B - 1u) / P.B;
        P.ref_b = reinterpret_cast<const uint8_t *>(refc);
        P.read_b = reinterpret_cast<const uint8_t *>(readc);
        P.ck = const_cast<uint32_t *>(A.dir) + pd.dir_off;
        P.match = A.match; P.mismatch = A.mismatch;
        P.g = g;
        P.raw_ref = A.raw ? A.raw + A.raw_off[pd.ref_id] : nullptr;
        P.raw_read = A.raw ? A.raw + A.raw_off[A.raw_reads_at + pd.read_id] : nullptr;
        int lane_max = 0;
        const int pmax = tf_sweep_dispatch(P, lane, lane_max);
        if (A.dbg) tk_sweep = __builtin_amdgcn_s_memtime() - tk_start - tk_pro;
        TF_MARK(0x200u);
        if (pmax <= 0) {                                                   // every cell ties at 0: SmithWaterman.java:154,182-185
            PairOut po;
            po.score = 0; po.flags = SWMI_F_DEGENERATE | SWMI_F_DONE; po.n_cells = (uint64_t)m * n;
            if (lane == 0) { A.out[pd.out_id] = po; if (A.out_host) A.out_host[pd.out_id] = po; }
        } else {
            // candidate stripes, right to left, grouped into blocks: block tasks for the workgroup's queue
            uint64_t cand = BALLOT(lane_max == pmax && lane < P.L);
            // the checkpoints are read back by other lanes and wavefronts of this workgroup: stores done, loads through L2
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            // A block is right-aligned on a candidate stripe and takes the cells of that stripe and the one before it: cells
            // further left would have little of the block to their left, and their walks would leave it at once -- they get a
            // block of their own (which another wavefront can take).
            const uint32_t own_ls = 63u - (uint32_t)__builtin_clzll(cand);
            const uint32_t own_end = P.B * (own_ls + 1u);
            const uint32_t own_kd = own_end > TF_BW ? (own_end - TF_BW + P.B - 1u) / P.B : 0u;
            const uint32_t own_hi = own_end < n ? own_end : n;
            const uint32_t own_first = own_ls + 1u >= own_kd + TF_ACC ? own_ls + 1u - TF_ACC : own_kd;      // first stripe whose cells this block takes
            uint64_t left_rest = 0;
            uint32_t left_top = 0;
            if (lane == 0) {
                PairOut po; po.score = pmax; po.flags = 0u; po.n_cells = 0; A.out[pd.out_id] = po;
                myslot->n = n; myslot->m = m; myslot->out_id = pd.out_id; myslot->pmax = (uint32_t)pmax;
                myslot->ref_id = pd.ref_id; myslot->read_id = pd.read_id;
                const unsigned long long cka = (unsigned long long)(uintptr_t)P.ck;
                myslot->ck_lo = (uint32_t)cka; myslot->ck_hi = (uint32_t)(cka >> 32);
                // the rightmost block is this wavefront's own next piece of work; the others go to the queue
                uint32_t acc_top = P.B * own_first, ntask = 1;
                uint64_t rest = own_first ? cand & ((1ull << own_first) - 1ull) : 0ull;
                if (rest) {
                    uint32_t at = tf_queue_lock(A, &sh->qb);
                    while (rest) {
                        const uint32_t ls = 63u - (uint32_t)__builtin_clzll(rest);
                        const uint32_t s_end = P.B * (ls + 1u);
                        const uint32_t kd = s_end > TF_BW ? (s_end - TF_BW + P.B - 1u) / P.B : 0u;
                        const uint32_t first = ls + 1u >= kd + TF_ACC ? ls + 1u - TF_ACC : kd;
                        if (at < TF_QCAP) sh->qb.e[at++] = make_uint4(wave, kd, P.B * first, s_end < acc_top ? s_end : acc_top);
                        else if (!left_rest) { left_rest = rest; left_top = acc_top; }     // (queue full: this wavefront does the rest itself)
                        ++ntask;
                        acc_top = P.B * first;
                        rest &= first ? ((1ull << first) - 1ull) : 0ull;
                    }
                    myslot->tasks_total = ntask;
                    tf_queue_publish(&sh->qb, at);
                } else myslot->tasks_total = ntask;
            }
            left_rest = ((uint64_t)tf_uni((uint32_t)(left_rest >> 32)) << 32) | tf_uni((uint32_t)left_rest);
            left_top = tf_uni(left_top);
            if (lane == 0) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); tf_lds_add(&sh->owners_done, 1u); }
            counted = true;
            const unsigned long long tt0 = A.dbg ? __builtin_amdgcn_s_memtime() : 0ull;
            tf_block_task<STRICT>(A, X, sh, wave, P, own_kd, P.B * own_first, own_hi, wave, lane, R, X.n_helpers != 0u);
            if (A.dbg) { tk_task0 = __builtin_amdgcn_s_memtime() - tt0; ++n_taken; }
            while (left_rest) {                                            // blocks that did not fit the queue
                const uint32_t ls = 63u - (uint32_t)__builtin_clzll(left_rest);
                const uint32_t s_end = P.B * (ls + 1u);
                const uint32_t kd = s_end > TF_BW ? (s_end - TF_BW + P.B - 1u) / P.B : 0u;
                const uint32_t first = ls + 1u >= kd + TF_ACC ? ls + 1u - TF_ACC : kd;
                tf_block_task<STRICT>(A, X, sh, wave, P, kd, P.B * first, s_end < left_top ? s_end : left_top, wave, lane, R, X.n_helpers != 0u);
                left_top = P.B * first;
                left_rest &= first ? ((1ull << first) - 1ull) : 0ull;
            }
        }
    }
    TF_MARK(0x300u);
    if (lane == 0 && wave < TF_WAVES && !counted) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        tf_lds_add(&sh->owners_done, 1u);
    }

    // ---- B + C: tasks, any pair of the workgroup ------------------------------------------------------------------------------
    auto pair_of = [&](const uint32_t sw, TfPair &P) {
        TfSlot *slot = &sh->slot[sw];
        P.n = tf_uni(slot->n); P.m = tf_uni(slot->m);
        P.B = swmi_tf_cols_per_lane(P.n);
        P.L = (P.n + P.B - 1u) / P.B;
        uint32_t *codes = regions + sw * X.lds_words + region_codes;
        P.ref_b = reinterpret_cast<const uint8_t *>(codes);
        P.read_b = reinterpret_cast<const uint8_t *>(codes + X.ref_words);
        P.ck = reinterpret_cast<uint32_t *>((uintptr_t)(((unsigned long long)tf_uni(slot->ck_hi) << 32) | tf_uni(slot->ck_lo)));
        P.match = A.match; P.mismatch = A.mismatch;
        P.g = g;
        P.raw_ref = A.raw ? A.raw + A.raw_off[tf_uni(slot->ref_id)] : nullptr;
        P.raw_read = A.raw ? A.raw + A.raw_off[A.raw_reads_at + tf_uni(slot->read_id)] : nullptr;
    };
    // everybody: block tasks first, then walk items, until every pair of the workgroup is through.  A helper lists cells
    // without handing any out (allow_push false): nobody ever reads a helper's tile, so a helper never waits.
    for (uint32_t spins = 0;;) {
        const unsigned long long ti0 = A.dbg ? __builtin_amdgcn_s_memtime() : 0ull;
        uint32_t idx = tf_queue_try_take(&sh->qb, lane);
        if (idx != 0xFFFFFFFFu) {
            TF_MARK(0x500u | idx);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            const uint4 e = sh->qb.e[idx];
            const uint32_t sw = tf_uni(e.x) & 0xFFu;
            TfPair P;
            pair_of(sw, P);
            tf_block_task<STRICT>(A, X, sh, sw, P, tf_uni(e.y), tf_uni(e.z), tf_uni(e.w), wave, lane, R, wave < TF_WAVES && X.n_helpers != 0u);
            ++n_taken;
            spins = 0;
            continue;
        }
        idx = X.n_helpers ? tf_queue_try_take(&sh->qw, lane) : 0xFFFFFFFFu;
        if (idx != 0xFFFFFFFFu) {
            TF_MARK(0x800u | idx);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            const uint4 e = sh->qw.e[idx];
            const uint32_t e0 = tf_uni(e.x), sw = e0 & 0xFFu, owner = (e0 >> 8) & 0xFFu;
            TfSlot *slot = &sh->slot[sw];
            TfPair P;
            pair_of(sw, P);
            const uint32_t *otile = regions + owner * X.lds_words;
            (void)tf_walk<STRICT>(A, X, sh, P, tf_uni(slot->out_id), (int)tf_uni(slot->pmax), tf_uni(e.z), tf_uni(e.w), otile, tf_uni(e.y),
                                  (int)owner, wave, lane, R);         // (its own tile too is given back before it may be overwritten)
            tf_task_done(A, slot, 0u, lane);
            spins = 0;
            continue;
        }
        // nothing more can come once every sweeper has queued its tasks and every pair's tasks are done
        if (tf_lds_load(&sh->owners_done) >= TF_WAVES) {
            bool open = false;
            for (uint32_t w = 0; w < TF_WAVES; ++w) open = open || tf_lds_load(&sh->slot[w].tasks_done) != tf_lds_load(&sh->slot[w].tasks_total);
            if (!open) break;
        }
        if (++spins > (1u << 22)) { if (lane == 0 && A.ovf_host) A.ovf_host[1] = 0xDEAD0002u; break; }    // (never seen: a wavefront must not hang)
        __builtin_amdgcn_s_sleep(8);
        if (A.dbg) tk_idle += __builtin_amdgcn_s_memtime() - ti0;
    }
    TF_MARK(0x700u);
    if (A.dbg && lane == 0 && dbg_id != 0xFFFFFFFFu) {   // {wave lifetime, sweep | prologue << 32, first block task << 16, block tasks taken | wait << 32}
        A.dbg[4ull * dbg_id + 0] = __builtin_amdgcn_s_memtime() - tk_start;
        A.dbg[4ull * dbg_id + 1] = (tk_sweep & 0xFFFFFFFFull) | (tk_pro << 32);
        A.dbg[4ull * dbg_id + 2] = tk_task0 << 16;
        A.dbg[4ull * dbg_id + 3] = n_taken | (tk_idle << 32);
    }
}

}  // namespace

extern "C" __global__ void __launch_bounds__(WAVE * (TF_WAVES + TF_HELPERS))
sw_tfused_kernel(const TraceArgs A, const TFusedArgs X) {
    extern __shared__ uint32_t tf_lds[];
    const uint32_t wave = tf_uni(threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (A.strict) tf_workgroup<true>(A, X, wave, lane, tf_lds);
    else          tf_workgroup<false>(A, X, wave, lane, tf_lds);
}

extern "C" hipError_t swmi_launch_tfused(const TraceArgs *a, const TFusedArgs *x, hipStream_t st) {
    if (x->n_items == 0) return hipSuccess;
    static const bool attr = [] { (void)hipFuncSetAttribute(reinterpret_cast<const void *>(sw_tfused_kernel),
                                                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); return true; }();
    (void)attr;
    const uint32_t groups = (x->n_items + TF_WAVES - 1) / TF_WAVES;
    size_t lds = (size_t)(TF_WAVES + x->n_helpers) * x->lds_words * sizeof(uint32_t) + ((sizeof(TfShared) + 3u) / 4u) * 4u;
    // a launch that fits the chip once: an LDS request that keeps the dispatcher from stacking workgroups on some CUs while
    // others stay empty (swmi_kernels.hip, spread_lds)
    static const int spread = getenv("SWMI_LDS_SPREAD") ? atoi(getenv("SWMI_LDS_SPREAD")) : 1;
    if (spread && groups <= 4u * 256u) {
        const size_t even = ((size_t)(160u * 1024u) / ((groups + 255u) / 256u)) & ~(size_t)1023;
        if (even > lds) lds = even;
    }
    hipLaunchKernelGGL(sw_tfused_kernel, dim3(groups), dim3(WAVE * (TF_WAVES + TF_HELPERS)), lds, st, *a, *x);
    return hipGetLastError();
}
