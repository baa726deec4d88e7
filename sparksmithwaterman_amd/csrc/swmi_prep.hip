// swmi_prep.hip -- gfx950 kernels that prepare a batch on the device (no DP arithmetic here).
//
// sw_encode_kernel: raw sequence bytes (as the caller / the FASTA reader hands them over) -> the canonical byte
// images the sweep reads (swmi_device.h: `seqw`), one wavefront per sequence.  This is the device half of what
// AlignmentScore's `Character.toUpperCase(x) == Character.toUpperCase(y)` (src/sw/SmithWaterman.java:309-318)
// needs: a 256-entry table maps a byte to its canonical code, so that code equality <=> that equality; the host only
// uploads the untouched bytes (H2D at PCIe rate) instead of encoding 10^9 bases on one core.
// It also derives SeqDesc.acgt (every base one of the eight fast symbols) with one ballot per 64 dwords.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "swmi_device.h"

#define WAVE 64
#define ENC_WAVES 4

// One wavefront per sequence.  raw_off[s] .. raw_off[s+1] are the sequence's bytes in `raw`; its image starts at dword
// desc[s].boff of `seqw` and is followed by zero pad dwords (the whole of seqw is zeroed by the host runtime before).
extern "C" __global__ void __launch_bounds__(WAVE * ENC_WAVES)
sw_encode_kernel(const uint8_t *__restrict__ raw, const uint64_t *__restrict__ raw_off, SeqDesc *__restrict__ desc,
                 uint32_t *__restrict__ seqw, const uint8_t *__restrict__ lut, uint32_t n_seq) {
    __shared__ uint8_t T[256];
    T[threadIdx.x] = lut[threadIdx.x];            // blockDim.x == 256
    __syncthreads();
    const uint32_t s = blockIdx.x * ENC_WAVES + (threadIdx.x >> 6);
    if (s >= n_seq) return;
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t b0 = raw_off[s];
    const uint32_t len = (uint32_t)(raw_off[s + 1] - b0);
    uint32_t *__restrict__ img = seqw + desc[s].boff;
    const uint8_t *__restrict__ src = raw + b0;
    const uint32_t nw = (len + 3u) >> 2;
    uint32_t bad = 0;
    for (uint32_t w = lane; w < nw; w += WAVE) {
        const uint32_t k = 4u * w;
        uint32_t v = 0;
        // (the image offset is 16-byte aligned, the raw offset is not: byte loads, served by L1)
        const uint32_t c0 = T[src[k]];
        const uint32_t c1 = k + 1u < len ? T[src[k + 1u]] : 0u;
        const uint32_t c2 = k + 2u < len ? T[src[k + 2u]] : 0u;
        const uint32_t c3 = k + 3u < len ? T[src[k + 3u]] : 0u;
        v = c0 | (c1 << 8) | (c2 << 16) | (c3 << 24);
        bad |= v & 0xE3E3E3E3u;                    // a fast symbol's code is 0, 4, ..., 28
        img[w] = v;
    }
    const uint64_t anybad = __builtin_amdgcn_ballot_w64(bad != 0u);
    if (lane == 0) desc[s].acgt = anybad ? 0u : 1u;
}

extern "C" hipError_t swmi_launch_encode(const uint8_t *raw, const uint64_t *raw_off, SeqDesc *desc, uint32_t *seqw,
                                         const uint8_t *lut, uint32_t n_seq, hipStream_t st) {
    if (n_seq == 0) return hipSuccess;
    hipLaunchKernelGGL(sw_encode_kernel, dim3((n_seq + ENC_WAVES - 1) / ENC_WAVES), dim3(WAVE * ENC_WAVES), 0, st,
                       raw, raw_off, desc, seqw, lut, n_seq);
    return hipGetLastError();
}
