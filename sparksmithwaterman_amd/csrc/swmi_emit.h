// swmi_emit.h -- device side of GetAlignment's output (src/sw/SmithWaterman.java:388-406, 418-431): the two aligned strings
// of one alignment, written by a whole wavefront into the alignment's payload in the arena (swmi_device.h: AlnRec).
//
// The walk leaves the ops of a path from the maximum cell backwards (op 0 = the maximum cell).  The reference pushes one
// {refChar, readChar} pair per step on a stack and pops it into the strings, so character p of both strings belongs to op
// t = n_ops - 1 - p.  Op t sits on cell (i_t, j_t) = (end_i - #{ops before t that consume a read base},
// end_j - #{... a reference base}): an alignment move takes ref[j-1] and read[i-1], an insertion '_' and read[i-1], a deletion
// ref[j-1] and '_' (:388-406).  Characters are the caller's ORIGINAL bytes (`raw`, as uploaded): case survives, as in the
// reference, which upper-cases only inside AlignmentScore.
//
// 64 ops per pass: lane l takes the op whose two characters land on string position base + 63 - l, so t rises with the lane and
// the two prefix counts are one ballot + mbcnt each.  Four passes fill 256 characters of both strings in an LDS scratch
// (byte stores), then every lane stores one dword of each string: 256-byte coalesced stores, also when the arena is pinned
// host memory.  Each string is followed by at least one NUL and padded to a dword: n_ops / 4 + 1 dwords.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "swmi_device.h"

#define SWMI_EMIT_SCRATCH_WORDS 128u      // LDS dwords emit_strings needs: 256 characters of each string

struct SwmiOpsPerByte {                   // ops staged one per byte (traceback_pair, tf_walk)
    const uint8_t *b;
    __device__ __forceinline__ uint32_t operator()(uint32_t t) const { return b[t]; }
};
struct SwmiOpsPacked {                    // ops packed 16 per dword, op t at bits 2 * (t % 16) (resident_pair)
    const uint32_t *w;
    __device__ __forceinline__ uint32_t operator()(uint32_t t) const { return (w[t >> 4] >> (2u * (t & 15u))) & 3u; }
};

// dwords of ONE string of an alignment of n_ops steps (characters + NUL, padded)
SWMI_HD static inline uint32_t swmi_str_words(uint32_t n_ops) { return n_ops / 4u + 1u; }

// dwords of an alignment's payload in the arena: both strings, or -- records without strings -- the packed ops
SWMI_HD static inline uint32_t swmi_payload_words(uint32_t n_ops, bool strings) {
    return strings ? 2u * swmi_str_words(n_ops) : (n_ops + 15u) / 16u;
}

// Lane 0 reserves `words` dwords of arena and `n_rec` table entries with ONE atomic (swmi_device.h: ArenaHdr).  Two halves,
// so that its round trip (device scope, ~1-2 us) overlaps what the wavefront can prepare without knowing where its record goes (packing the ops,
// the characters of the strings): swmi_reserve_issue returns the pending values, swmi_reserve_finish makes them wave-uniform
// {payload offset, first table slot}.  false: something did not fit -- the caller raises SWMI_F_ARENA_OVF and the host re-runs
// with the sizes the header then holds.
struct SwmiReserve { unsigned long long v; };
__device__ __forceinline__ SwmiReserve swmi_reserve_issue(const TraceArgs &A, const uint32_t lane, const uint32_t words, const uint32_t n_rec) {
    SwmiReserve r{0ull};
    if (lane == 0) r.v = atomicAdd(&A.hdr->reserved, ((unsigned long long)n_rec << SWMI_HDR_WORD_BITS) | (unsigned long long)words);
    return r;
}
__device__ __forceinline__ bool swmi_reserve_finish(const TraceArgs &A, const SwmiReserve r, const uint32_t words, const uint32_t n_rec,
                                                    unsigned long long &off, uint32_t &slot) {
    const unsigned long long v = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(r.v >> 32)) << 32) |
                                 (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)r.v);
    off = v & SWMI_HDR_WORD_MASK;
    slot = (uint32_t)(v >> SWMI_HDR_WORD_BITS);
    return off + words <= A.arena_cap_words && (unsigned long long)slot + n_rec <= (unsigned long long)A.rec_tab_cap;
}
__device__ __forceinline__ bool swmi_reserve(const TraceArgs &A, const uint32_t lane, const uint32_t words, const uint32_t n_rec,
                                             unsigned long long &off, uint32_t &slot) {
    return swmi_reserve_finish(A, swmi_reserve_issue(A, lane, words, n_rec), words, n_rec, off, slot);
}

__device__ __forceinline__ void swmi_write_rec(const TraceArgs &A, const uint32_t slot, const uint32_t out_id, const uint32_t rank,
                                               const int begin, const uint32_t end_i, const uint32_t end_j, const uint32_t n_ops,
                                               const unsigned long long off) {
    uint4 *e = reinterpret_cast<uint4 *>(A.rec_tab + slot);
    e[0] = make_uint4(out_id, rank, (uint32_t)begin, end_i);
    e[1] = make_uint4(end_j, n_ops, (uint32_t)off, (uint32_t)(off >> 32));
}

// The strings of one alignment, 256 characters (one dword of each string per lane) at a time, from the strings' END to their
// start -- t ascending, so the cell of the next op follows from the counts of the ops before it.  chunk(c) must be called
// for c = n_chunks() - 1, ..., 0; it needs no destination: the caller stores the lane's two dwords (word 64 c + lane of the
// reference-side string and of the read-side one) wherever the record went.
template <class OPS>
struct SwmiStrings {
    OPS ops;
    uint32_t n_ops, i, j, lane;
    const uint8_t *__restrict__ raw_ref, *__restrict__ raw_read;
    uint32_t *__restrict__ scratch;
    __device__ __forceinline__ SwmiStrings(const OPS o, uint32_t n, uint32_t end_i, uint32_t end_j, const uint8_t *rr, const uint8_t *rq,
                                           uint32_t l, uint32_t *sc)
        : ops(o), n_ops(n), i(end_i), j(end_j), lane(l), raw_ref(rr), raw_read(rq), scratch(sc) {}
    __device__ __forceinline__ uint32_t words() const { return swmi_str_words(n_ops); }
    __device__ __forceinline__ uint32_t n_chunks() const { return (words() + 63u) / 64u; }
    __device__ __forceinline__ void chunk(const uint32_t c, uint32_t &wr, uint32_t &wq) {
        uint8_t *sr = reinterpret_cast<uint8_t *>(scratch), *sq = sr + 256;
        uint32_t cr[4], cq[4];
#pragma unroll
        for (uint32_t q = 4u; q-- > 0u;) {
            cr[q] = 0u; cq[q] = 0u;
            // (a pass none of whose characters lands in a stored dword -- the strings end at n_ops, their padding before
            //  n_ops + 4 -- is skipped: two of the four passes for the 80-step alignments of the EngineerData shapes)
            if (256u * c + 64u * q >= n_ops + 4u) continue;
            const uint32_t p = 256u * c + 64u * q + 63u - lane;
            const bool valid = p < n_ops;
            const uint32_t op = valid ? ops(n_ops - 1u - p) : 3u;
            const bool ur = valid && op != SWMI_DIR_I, uq = valid && op != SWMI_DIR_D;
            const uint64_t mr = __builtin_amdgcn_ballot_w64(ur), mq = __builtin_amdgcn_ballot_w64(uq);
            const uint32_t jj = j - __builtin_amdgcn_mbcnt_hi((uint32_t)(mr >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mr, 0u));
            const uint32_t ii = i - __builtin_amdgcn_mbcnt_hi((uint32_t)(mq >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mq, 0u));
            cr[q] = valid ? (uint32_t)'_' : 0u;          // SmithWaterman.java:356
            cq[q] = cr[q];
            if (ur) cr[q] = raw_ref[jj - 1u];            // (the loads of the four passes are in flight together)
            if (uq) cq[q] = raw_read[ii - 1u];
            j -= (uint32_t)__builtin_popcountll(mr);
            i -= (uint32_t)__builtin_popcountll(mq);
        }
#pragma unroll
        for (uint32_t q = 0; q < 4u; ++q) {
            if (256u * c + 64u * q >= n_ops + 4u) continue;
            sr[64u * q + 63u - lane] = (uint8_t)cr[q];
            sq[64u * q + 63u - lane] = (uint8_t)cq[q];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        wr = scratch[lane];
        wq = scratch[64u + lane];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    // every chunk from `c_top` down, stored behind the packed ops at `dst`
    __device__ __forceinline__ void store_from(uint32_t *__restrict__ dst, uint32_t c_top) {
        const uint32_t sw = words();
        for (uint32_t c = c_top; c-- > 0u;) {
            uint32_t wr, wq;
            chunk(c, wr, wq);
            const uint32_t w = 64u * c + lane;
            if (w < sw) { dst[w] = wr; dst[sw + w] = wq; }
        }
    }
};

template <class OPS>
__device__ __forceinline__ void swmi_emit_strings(uint32_t *__restrict__ dst, const OPS ops, const uint32_t n_ops,
                                                  const uint32_t end_i, const uint32_t end_j,
                                                  const uint8_t *__restrict__ raw_ref, const uint8_t *__restrict__ raw_read,
                                                  const uint32_t lane, uint32_t *__restrict__ scratch) {
    SwmiStrings<OPS> S(ops, n_ops, end_i, end_j, raw_ref, raw_read, lane, scratch);
    S.store_from(dst, S.n_chunks());
}
