// stepbench.hip -- cycles per anti-diagonal step of the fill kernel's steady-state block, in isolation.
//   hipcc --offload-arch=gfx950 -O3 -I sparksmithwaterman_amd/csrc tools/stepbench.hip -o /tmp/stepbench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
#include "swmi_cells_gen.inc"

__device__ __forceinline__ int wave_shr1(int old, int src) { return __builtin_amdgcn_update_dpp(old, src, 0x138, 0xf, 0xf, false); }
__device__ __forceinline__ int wave_shr1_zero(int src) { return __builtin_amdgcn_update_dpp(0, src, 0x138, 0xf, 0xf, true); }

// V: 0 = full step as compiled around the asm stream (feed + 2 dpp + cells + max3 + cmp + deferred branch)
//    1 = no tied-maximum check     2 = cells only (no dpp)
// DIRS: direction bits (mode 0 fill / mode 1 replay) or scores only (mode 1 fill)
template <int R, int V, bool DIRS>
__global__ void __launch_bounds__(256) k(int *out, const uint4 *words, int iters, long long *cyc, int gap, int thr_in) {
    int ha[R], hb[R]; uint32_t acc[R]; int q[R];
    for (int i = 0; i < R; ++i) { ha[i] = 0; hb[i] = 0; acc[i] = 0; q[i] = 0xFDFDFD05 ^ ((threadIdx.x * 7 + i) % 4 == 0 ? 0 : 0x08000000 >> (8 * ((threadIdx.x + i) % 3))); }
    int rb = 0, nprev = 0, thr = thr_in, ev = 0;
    unsigned long long ev_prev = 0;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        const uint4 w = words[it & 63];
#pragma unroll
        for (uint32_t s = 0; s < 16; ++s) {
            const int (&hin)[R] = (s & 1u) ? hb : ha;
            int (&hout)[R] = (s & 1u) ? ha : hb;
            const uint32_t wsel = s < 4 ? w.x : s < 8 ? w.y : s < 12 ? w.z : w.w;
            const int feed = (int)((wsel >> (8u * (s & 3u))) & 0x18u);
            int nin;
            if (V != 2) { rb = wave_shr1(feed, rb); nin = wave_shr1_zero(hin[R - 1]); } else { rb = feed; nin = nprev + 1; }
            CellsAsm<R, true, false, DIRS>::step(hin, hout, acc, q, rb, nprev, nin, gap, 5, -3);
            nprev = nin;
            if (V == 0) {
                int mrow = hout[0];
                for (int kk = 1; kk < R; ++kk) mrow = mrow > hout[kk] ? mrow : hout[kk];
                const unsigned long long e = __builtin_amdgcn_ballot_w64(mrow >= thr);
                if (__builtin_expect(ev_prev != 0, 0)) { ev++; thr += 1000; }
                ev_prev = e;
            }
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    int sum = ev + rb;
    for (int i = 0; i < R; ++i) sum += ha[i] + hb[i] + acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int R, int V, bool DIRS> void run(const char *name, int *dout, uint4 *dw, long long *dcyc) {
    const int iters = 128;
    for (int wps = 1; wps <= 4; wps *= 2) {
        int blocks = 256 * wps;
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        hipLaunchKernelGGL((k<R, V, DIRS>), dim3(blocks), dim3(256), 0, 0, dout, dw, 8, dcyc, -4, 1 << 30);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k<R, V, DIRS>), dim3(blocks), dim3(256), 0, 0, dout, dw, iters, dcyc, -4, 1 << 30);
        CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<long long> cyc(blocks);
        CK(hipMemcpy(cyc.data(), dcyc, blocks * sizeof(long long), hipMemcpyDeviceToHost));
        double mean = 0; for (auto c : cyc) mean += c; mean /= blocks;
        double steps = iters * 16.0;
        printf("R=%d %-34s waves/SIMD=%d  wall=%.3f ms  ticks/step=%.1f  ns/step(wall)=%.1f  chip GCUPS=%.0f\n", R, name, wps, ms,
               mean / steps, ms * 1e6 / steps, blocks * 4.0 * steps * R * 64 / (ms * 1e-3) / 1e9);
    }
}

int main() {
    int *dout; long long *dcyc; uint4 *dw;
    CK(hipMalloc(&dout, 256 * 8 * 256 * sizeof(int))); CK(hipMalloc(&dcyc, 256 * 8 * sizeof(long long)));
    std::vector<uint32_t> hw(64 * 4);
    for (size_t i = 0; i < hw.size(); ++i) hw[i] = (uint32_t)rand() * 2654435761u;
    CK(hipMalloc(&dw, hw.size() * 4)); CK(hipMemcpy(dw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    run<3, 0, true>("dirs: full step", dout, dw, dcyc);
    run<3, 2, true>("dirs: cells only", dout, dw, dcyc);
    run<3, 0, false>("score-only: full step", dout, dw, dcyc);
    run<3, 1, false>("score-only: no max check", dout, dw, dcyc);
    run<3, 2, false>("score-only: cells only", dout, dw, dcyc);
    run<2, 0, false>("score-only: full step", dout, dw, dcyc);
    run<4, 0, false>("score-only: full step", dout, dw, dcyc);
    return 0;
}
