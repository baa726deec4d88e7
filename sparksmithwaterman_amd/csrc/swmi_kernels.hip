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
#include <stdint.h>
#include "swmi_device.h"

#define WAVE 64

// ------------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------------
// v_mov_b32_dpp wave_shr:1 : lane l receives lane l-1's value, lane 0 keeps `old`.
__device__ __forceinline__ int wave_shr1(int old, int src) {
    return __builtin_amdgcn_update_dpp(old, src, 0x138 /*wave_shr:1*/, 0xf, 0xf, false);
}

__device__ __forceinline__ int wave_max_i32(int v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        int o = __shfl_xor(v, off, WAVE);
        v = v > o ? v : o;
    }
    return v;
}

__device__ __forceinline__ uint32_t lanemask_lt_count(uint64_t mask) {
    // number of set bits of `mask` below this lane
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// acc = 2*acc + bit, the bit coming straight from a compare's lane mask (v_cmp + v_addc_co_u32).
__device__ __forceinline__ uint32_t push_bit(uint32_t acc, bool bit) {
    return acc + acc + (bit ? 1u : 0u);
}

// a value every lane holds identically -> scalar register (keeps the per-step base extraction on the SALU)
__device__ __forceinline__ uint32_t uniform_u32(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}

__device__ __forceinline__ uint32_t seq_code_bytes(const uint32_t *__restrict__ w, uint32_t pos) {
    return (w[pos >> 2] >> (8u * (pos & 3u))) & 0xFFu;
}
__device__ __forceinline__ uint32_t seq_code_packed(const uint32_t *__restrict__ w, uint32_t pos) {
    return (w[pos >> 4] >> (2u * (pos & 15u))) & 3u;
}

// ------------------------------------------------------------------------------------------------
// fill: one pair, one wavefront.  R rows per lane; ACGT = both sequences pure ACGT (profile lookup
// by v_bfe_i32 instead of compare+select); STRICT = DistributedSW tie order; MULTI = more than one
// strip of 64*R rows (seam row through memory).
// ------------------------------------------------------------------------------------------------
template <int R, bool ACGT, bool STRICT, bool MULTI>
__device__ __forceinline__ void fill_pair(const FillArgs &A, const PairDesc pd, const uint32_t lane) {
    const SeqDesc rd = A.refs[pd.ref_id];
    const SeqDesc qd = A.reads[pd.read_id];
    const uint32_t n = rd.len, m = qd.len;
    const uint32_t *__restrict__ refw = A.seqw + (ACGT ? rd.poff : rd.boff);
    const uint32_t *__restrict__ readw = A.seqw + (ACGT ? qd.poff : qd.boff);
    const int match = A.match, mismatch = A.mismatch, gap = A.gap;

    const uint32_t rps = WAVE * R;                       // rows per strip
    const uint32_t n_strips = (m + rps - 1) / rps;
    const uint32_t wblocks = (n + 63u + 15u) / 16u;      // 16-step blocks reserved per strip
    const uint64_t strip_words = (uint64_t)wblocks * R * WAVE;

    const uint64_t cbase = A.cells_off ? A.cells_off[pd.out_id] : (uint64_t)pd.out_id * A.cell_cap;
    const uint32_t ccap = A.cells_cap ? A.cells_cap[pd.out_id] : A.cell_cap;
    uint2 *__restrict__ cells = A.cells + cbase;

    int thr = 1;             // wave-uniform running maximum (a max of 0 never enters the list: degenerate case)
    uint32_t cnt = 0;        // wave-uniform number of cells equal to thr

    for (uint32_t s = 0; s < n_strips; ++s) {
        const uint32_t row0 = s * rps + lane * R;        // 0-based first row of this lane
        const uint32_t rows_left = m - s * rps;
        const uint32_t lact = rows_left >= rps ? WAVE : (rows_left + R - 1) / R;   // lanes holding rows
        const uint32_t T = n + lact - 1;                 // steps of this strip

        // read-side operands of this lane's rows
        int q[R];                                        // ACGT: 4 signed score bytes; else the base code
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const uint32_t row = row0 + k;
            if (ACGT) {
                uint32_t p = (uint32_t)(mismatch & 0xFF) * 0x01010101u;
                if (row < m) {
                    const uint32_t c = seq_code_packed(readw, row);
                    p = (p & ~(0xFFu << (8u * c))) | ((uint32_t)(match & 0xFF) << (8u * c));
                }
                q[k] = (int)p;
            } else {
                q[k] = row < m ? (int)seq_code_bytes(readw, row) : (int)SWMI_CODE_PAD;
            }
        }

        int h[R];
        uint32_t acc[R];
#pragma unroll
        for (int k = 0; k < R; ++k) { h[k] = 0; acc[k] = 0; }
        int nprev = 0;                                   // N received one step earlier = NW of this step
        int hb = 0;                                      // bottom-row H of the previous step (what lane l+1 reads)
        int rb = 0;                                      // reference base operand of this lane's current column
        int c0 = lane < lact ? -(int)lane - 1 : -(1 << 30);   // after the step's increment: c0 = j - 1

        uint32_t *__restrict__ dirp = A.dir + pd.dir_off + s * strip_words + lane;
        const int32_t *seam_in = nullptr;
        int32_t *seam_out = nullptr;
        if (MULTI) {
            int32_t *sb = A.seam + pd.seam_off;
            seam_in = sb + ((s + 1) & 1) * (uint64_t)(n + 1);    // written by strip s-1
            seam_out = sb + (s & 1) * (uint64_t)(n + 1);
        }
        const bool feeds_seam = MULTI && (s + 1 < n_strips);
        const bool reads_seam = MULTI && (s > 0);

        const uint32_t nblk = (T + 15u) / 16u;
        for (uint32_t tb = 0; tb < nblk; ++tb) {
            // reference codes of columns 16tb+1 .. 16tb+16 for lane 0 (wave-uniform, scalar registers)
            uint32_t rw0, rw1 = 0, rw2 = 0, rw3 = 0;
            if (ACGT) {
                rw0 = uniform_u32(refw[tb]);
            } else {
                rw0 = uniform_u32(refw[4 * tb]);     rw1 = uniform_u32(refw[4 * tb + 1]);
                rw2 = uniform_u32(refw[4 * tb + 2]); rw3 = uniform_u32(refw[4 * tb + 3]);
            }
            int seamv = 0;
            if (reads_seam) {
                // seam_in[16tb + 1 + lane] for lanes 0..15: N of lane 0 for the 16 steps of this block
                const uint32_t col = 16u * tb + 1u + (lane & 15u);
                seamv = col <= n ? __hip_atomic_load(seam_in + col, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
            }

#pragma unroll 4
            for (uint32_t sidx = 0; sidx < 16; ++sidx) {
                // ---- cross-lane part, all 64 lanes ------------------------------------------------
                uint32_t code;
                if (ACGT) {
                    code = ((rw0 >> (2u * sidx)) & 3u) * 8u;                    // bit offset into the profile
                } else {
                    const uint32_t wsel = (sidx >> 2) == 0 ? rw0 : (sidx >> 2) == 1 ? rw1 : (sidx >> 2) == 2 ? rw2 : rw3;
                    code = (wsel >> (8u * (sidx & 3u))) & 0xFFu;
                }
                rb = wave_shr1((int)code, rb);
                int topn = 0;
                if (reads_seam) topn = __builtin_amdgcn_readlane(seamv, sidx);
                const int nin = wave_shr1(topn, hb);
                c0 += 1;
                const bool active = (uint32_t)c0 < n;
                int mrow = -1;                 // stays -1 in lanes that are off their column range
                if (active) {
                    int diag = nprev, up = nin;
#pragma unroll
                    for (int k = 0; k < R; ++k) {
                        const int left = h[k];
                        int sc;
                        if (ACGT) sc = __builtin_amdgcn_sbfe(q[k], (unsigned)rb, 8u);
                        else      sc = (rb == q[k]) ? match : mismatch;
                        const int a = diag + sc;
                        const int t2 = (up > left ? up : left) + gap;
                        int hv = a > t2 ? a : t2;
                        hv = hv > 0 ? hv : 0;
                        const bool ba = STRICT ? (a > t2) : (a >= t2);
                        const bool bi = STRICT ? (up > left) : (up >= left);
                        acc[k] = push_bit(push_bit(acc[k], ba), bi);
                        diag = left;
                        up = hv;
                        h[k] = hv;
                    }
                    hb = h[R - 1];
                    mrow = h[0];
#pragma unroll
                    for (int k = 1; k < R; ++k) mrow = mrow > h[k] ? mrow : h[k];
                    if (feeds_seam && lane == WAVE - 1) seam_out[c0 + 1] = hb;
                }
                nprev = nin;

                // ---- rare: some lane reached the running maximum ---------------------------------
                // all 64 lanes vote, so thr/cnt below stay wave-uniform
                if (__builtin_expect(__ballot(mrow >= thr) != 0, 0)) {
                    int cand = -1;
#pragma unroll
                    for (int k = 0; k < R; ++k)
                        if (active && row0 + k < m) cand = cand > h[k] ? cand : h[k];
                    const int wmax = wave_max_i32(cand);
                    if (wmax >= thr) {
                        if (wmax > thr) { thr = wmax; cnt = 0; }
#pragma unroll
                        for (int k = 0; k < R; ++k) {
                            const bool hit = active && (row0 + k < m) && (h[k] == thr);
                            const uint64_t hm = __ballot(hit);
                            if (hm) {
                                const uint32_t pos = cnt + lanemask_lt_count(hm);
                                if (hit && pos < ccap) cells[pos] = make_uint2(row0 + k + 1, (uint32_t)c0 + 1u);
                                cnt += (uint32_t)__popcll(hm);
                            }
                        }
                    }
                }
            }

            // ---- end of a 16-step block: one coalesced 256 B store per row slot -------------------
            // lanes that finished their last column inside this block still owe the missing shifts
            const int t_end = (int)(16u * tb + 15u);
            const int t_last = (int)(lane + n) - 1;
            const int miss = t_end - t_last;
#pragma unroll
            for (int k = 0; k < R; ++k) {
                uint32_t v = acc[k];
                if (miss > 0 && miss < 16) v <<= 2 * miss;
                dirp[((uint64_t)tb * R + k) * WAVE] = v;
            }
        }
        if (MULTI) __threadfence();    // seam row of this strip visible before the next strip reads it
    }

    if (lane == 0) {
        PairOut o;
        if (cnt == 0) { o.score = 0; o.flags = SWMI_F_DEGENERATE; o.n_cells = (uint64_t)m * n; }
        else          { o.score = thr; o.flags = cnt > ccap ? SWMI_F_CELL_OVF : 0u; o.n_cells = cnt; }
        A.out[pd.out_id] = o;
    }
}

template <bool ACGT, bool STRICT>
__device__ __forceinline__ void fill_dispatch(const FillArgs &A, const PairDesc pd, uint32_t lane, uint32_t m) {
    const uint32_t R = swmi_rows_per_lane(m);
    if (R == 1)      fill_pair<1, ACGT, STRICT, false>(A, pd, lane);
    else if (R == 2) fill_pair<2, ACGT, STRICT, false>(A, pd, lane);
    else if (R == 3) fill_pair<3, ACGT, STRICT, false>(A, pd, lane);
    else if (m <= WAVE * SWMI_RMAX) fill_pair<SWMI_RMAX, ACGT, STRICT, false>(A, pd, lane);
    else             fill_pair<SWMI_RMAX, ACGT, STRICT, true>(A, pd, lane);
}

extern "C" __global__ void __launch_bounds__(WAVE)
sw_fill_kernel(const FillArgs A) {
    const uint32_t pair = blockIdx.x;
    if (pair >= A.n_pairs) return;
    const uint32_t lane = threadIdx.x;
    const PairDesc pd = A.pairs[pair];
    const SeqDesc rd = A.refs[pd.ref_id];
    const SeqDesc qd = A.reads[pd.read_id];
    // profile lookup needs both sequences pure ACGT and scores that fit a signed byte
    const bool acgt = rd.poff != SWMI_NO_PACKED && qd.poff != SWMI_NO_PACKED &&
                      A.match >= -128 && A.match <= 127 && A.mismatch >= -128 && A.mismatch <= 127;
    if (acgt) {
        if (A.strict) fill_dispatch<true, true>(A, pd, lane, qd.len);
        else          fill_dispatch<true, false>(A, pd, lane, qd.len);
    } else {
        if (A.strict) fill_dispatch<false, true>(A, pd, lane, qd.len);
        else          fill_dispatch<false, false>(A, pd, lane, qd.len);
    }
}

// ------------------------------------------------------------------------------------------------
// traceback: one wavefront per pair
// ------------------------------------------------------------------------------------------------
extern "C" __global__ void __launch_bounds__(WAVE)
sw_traceback_kernel(const TraceArgs A) {
    extern __shared__ uint32_t ops_lds[];
    const uint32_t pair = blockIdx.x;
    if (pair >= A.n_pairs) return;
    const uint32_t lane = threadIdx.x;
    const PairDesc pd = A.pairs[pair];
    const PairOut po = A.out[pd.out_id];
    if (po.flags & (SWMI_F_DEGENERATE | SWMI_F_CELL_OVF)) return;

    const SeqDesc rd = A.refs[pd.ref_id];
    const SeqDesc qd = A.reads[pd.read_id];
    const uint32_t n = rd.len, m = qd.len;
    const uint32_t *__restrict__ refw = A.seqw + rd.boff;
    const uint32_t *__restrict__ readw = A.seqw + qd.boff;
    const uint32_t R = swmi_rows_per_lane(m);
    const uint32_t rps = WAVE * R;
    const uint32_t wblocks = (n + 63u + 15u) / 16u;
    const uint64_t strip_words = (uint64_t)wblocks * R * WAVE;
    const uint32_t *__restrict__ dirp = A.dir + pd.dir_off;

    const uint64_t cbase = A.cells_off ? A.cells_off[pd.out_id] : (uint64_t)pd.out_id * A.cell_cap;
    const uint32_t ncell = (uint32_t)po.n_cells;
    const uint2 *__restrict__ cells = A.cells + cbase;

    // Process the tied cells in the order the reference lists them: row-major (SmithWaterman.java:157-185)
    // or per anti-diagonal with ascending j (DistributedSW.java:209-239).  Lists are short (<= cell_cap) in
    // the fast path; the re-run path may hand over long ones, so rank in chunks of 64.
    for (uint32_t base = 0; base < ncell; base += WAVE) {
        const uint32_t idx = base + lane;
        uint2 mine = make_uint2(0, 0);
        if (idx < ncell) mine = cells[idx];
        const uint64_t mykey = A.strict ? (((uint64_t)(mine.x + mine.y) << 32) | mine.y)
                                        : (((uint64_t)mine.x << 32) | mine.y);
        // rank of my cell among all cells of the pair
        uint32_t rank = 0;
        for (uint32_t o = 0; o < ncell; ++o) {
            const uint2 c = cells[o];
            const uint64_t k = A.strict ? (((uint64_t)(c.x + c.y) << 32) | c.y) : (((uint64_t)c.x << 32) | c.y);
            rank += (k < mykey) ? 1u : 0u;
        }
        const uint32_t nhere = ncell - base < WAVE ? ncell - base : WAVE;

        for (uint32_t a = 0; a < nhere; ++a) {
            const uint32_t ci = __builtin_amdgcn_readlane((int)mine.x, a);
            const uint32_t cj = __builtin_amdgcn_readlane((int)mine.y, a);
            const uint32_t crank = __builtin_amdgcn_readlane((int)rank, a);

            // ---- walk (lane 0), SmithWaterman.java:380-409 ----
            uint32_t n_ops = 0;
            int begin = 0;
            if (lane == 0) {
                uint32_t i = ci, j = cj;
                int score = po.score;
                uint32_t word = 0;
                while (score > 0) {
                    begin = (int)j;
                    const uint32_t r0 = i - 1;
                    const uint32_t s = r0 / rps, rl = r0 % rps;
                    const uint32_t l = rl / R, k = rl % R;
                    const uint32_t t = (j - 1) + l;
                    const uint32_t dw = dirp[s * strip_words + ((uint64_t)(t >> 4) * R + k) * WAVE + l];
                    const uint32_t d = (dw >> (2u * (15u - (t & 15u)))) & 3u;
                    uint32_t op;
                    if (d & 2u) {          // alignment: H(i-1,j-1) = H - s(ref[j-1], read[i-1])
                        const uint32_t rc = seq_code_bytes(refw, j - 1), qc = seq_code_bytes(readw, i - 1);
                        score = (int)((uint32_t)score - (uint32_t)(rc == qc ? A.match : A.mismatch));
                        --i; --j; op = SWMI_DIR_A;
                    } else if (d & 1u) {   // insertion: H(i-1,j) = H - gap
                        score = (int)((uint32_t)score - (uint32_t)A.gap);
                        --i; op = SWMI_DIR_I;
                    } else {               // deletion: H(i,j-1) = H - gap
                        score = (int)((uint32_t)score - (uint32_t)A.gap);
                        --j; op = SWMI_DIR_D;
                    }
                    word |= op << (2u * (n_ops & 15u));
                    if ((n_ops & 15u) == 15u) {
                        if ((n_ops >> 4) < A.lds_words) ops_lds[n_ops >> 4] = word;
                        word = 0;
                    }
                    ++n_ops;
                    if (i == 0 || j == 0) break;       // H is 0 on the border: the loop ends there too
                }
                if ((n_ops & 15u) != 0u && (n_ops >> 4) < A.lds_words) ops_lds[n_ops >> 4] = word;
            }
            n_ops = (uint32_t)__builtin_amdgcn_readfirstlane((int)n_ops);
            begin = __builtin_amdgcn_readfirstlane(begin);
            __syncthreads();

            // ---- append the record ----
            const uint32_t opw = (n_ops + 15u) / 16u;
            const uint32_t words = SWMI_ALNREC_WORDS + opw;
            unsigned long long off = 0;
            if (lane == 0) {
                off = atomicAdd(&A.hdr->used_words, (unsigned long long)words);
                atomicAdd(&A.hdr->n_records, 1ull);
            }
            off = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(off >> 32)) << 32) |
                  (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)off);
            if (off + words <= A.arena_cap_words && opw <= A.lds_words) {
                uint32_t *dst = A.arena + off;
                if (lane == 0) {
                    dst[0] = pd.out_id; dst[1] = crank; dst[2] = (uint32_t)begin;
                    dst[3] = ci; dst[4] = cj; dst[5] = n_ops;
                }
                for (uint32_t w = lane; w < opw; w += WAVE) dst[SWMI_ALNREC_WORDS + w] = ops_lds[w];
            } else if (lane == 0) {
                atomicOr(&A.out[pd.out_id].flags, SWMI_F_ARENA_OVF);
            }
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------------------------------------
// host-callable launchers (the runtime in swmi_api.cpp is plain C++)
// ------------------------------------------------------------------------------------------------
extern "C" hipError_t swmi_launch_fill(const FillArgs *a, hipStream_t st) {
    if (a->n_pairs == 0) return hipSuccess;
    hipLaunchKernelGGL(sw_fill_kernel, dim3(a->n_pairs), dim3(WAVE), 0, st, *a);
    return hipGetLastError();
}

extern "C" hipError_t swmi_launch_traceback(const TraceArgs *a, hipStream_t st) {
    if (a->n_pairs == 0) return hipSuccess;
    hipLaunchKernelGGL(sw_traceback_kernel, dim3(a->n_pairs), dim3(WAVE), a->lds_words * sizeof(uint32_t), st, *a);
    return hipGetLastError();
}
