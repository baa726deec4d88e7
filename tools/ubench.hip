// ubench.hip -- gfx950 micro-measurements that drive the kernel design (not part of the product).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench.hip -o gpurun_out/ubench && gpurun_out/ubench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

#define REP16(x) x x x x x x x x x x x x x x x x
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)

// variant 0: dependent v_add chain; 1: 4 independent chains; 2: dpp wave_shr chain; 3: cmp+addc pairs (with filler);
// 4: v_max3 chain; 5: v_bfe_i32 dep; 6: cmp+addc back-to-back with s_nop 1; 7: mix resembling one DP cell (9 instr)
template <int V>
__global__ void __launch_bounds__(256) k(int *out, int iters, long long *cyc) {
    int a = threadIdx.x, b = a * 3 + 1, c = a ^ 5, d = a + 7, e = 1, f = 2, g = 3, h = 4;
    unsigned acc = 0;
    int lm = 0, rb = a & 0x01010101, dg = 1, gp = -4, l0 = 0, l1 = 0, l2 = 0;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (V == 0) { asm volatile(REP64("v_add_u32 %0, %0, %1\n") : "+v"(a) : "v"(b)); }
        if (V == 1) { asm volatile(REP16("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n")
                                   : "+v"(a), "+v"(c), "+v"(d), "+v"(e) : "v"(b)); }
        if (V == 2) { asm volatile(REP64("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n") : "+v"(a)); }
        if (V == 3) { asm volatile(REP16("v_cmp_ge_i32 vcc, %1, %2\n v_add_u32 %3, %3, %2\n v_add_u32 %4, %4, %2\n v_addc_co_u32 %0, vcc, %0, %0, vcc\n")
                                   : "+v"(acc), "+v"(a) : "v"(b), "v"(c), "v"(d) : "vcc"); }
        if (V == 4) { asm volatile(REP64("v_max3_i32 %0, %0, %1, %2\n") : "+v"(a) : "v"(b), "v"(c)); }
        if (V == 5) { asm volatile(REP64("v_bfe_i32 %0, %1, %0, 8\n") : "+v"(a) : "v"(b)); }
        if (V == 6) { asm volatile(REP16("v_cmp_ge_i32 vcc, %1, %2\n s_nop 1\n v_addc_co_u32 %0, vcc, %0, %0, vcc\n v_add_u32 %1, %1, %2\n")
                                   : "+v"(acc), "+v"(a) : "v"(b) : "vcc"); }
        if (V == 7) {
            asm volatile(REP16(
                "v_bfe_i32 %4, %5, %6, 8\n"
                "v_add_u32 %4, %4, %1\n"
                "v_cmp_ge_i32 s[10:11], %2, %3\n"
                "v_max_i32 %7, %2, %3\n"
                "v_add_u32 %7, %7, %8\n"
                "v_cmp_ge_i32 s[12:13], %4, %7\n"
                "v_addc_co_u32 %0, vcc, %0, %0, s[10:11]\n"
                "v_max3_i32 %3, %4, %7, 0\n"
                "v_mov_b32 %1, %2\n"
                "v_addc_co_u32 %0, vcc, %0, %0, s[12:13]\n")
                : "+v"(acc), "+v"(a), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(b) : "vcc", "s10", "s11", "s12", "s13");
        }
        if (V == 8) { asm volatile(REP16("v_dot4_i32_i8 %0, %4, %5, %0\n v_dot4_i32_i8 %1, %4, %5, %1\n v_dot4_i32_i8 %2, %4, %5, %2\n v_dot4_i32_i8 %3, %4, %5, %3\n")
                                   : "+v"(a), "+v"(c), "+v"(d), "+v"(e) : "v"(b), "v"(f)); }
        if (V == 9) {      // score-only cell, dot4 form (3 rows interleaved as in the generated stream)
            asm volatile(REP16(
                "v_dot4_i32_i8 %0, %4, %5, %1\n"
                "v_max_i32 %3, %1, %2\n"
                "v_add_u32 %3, %3, %6\n"
                "s_nop 0\n"
                "v_max3_i32 %2, %0, %3, 0\n")
                : "+v"(a), "+v"(c), "+v"(d), "+v"(e) : "v"(b), "v"(f), "v"(g));
        }
        if (V == 10) {     // score-only cell, bfe + add form
            asm volatile(REP16(
                "v_bfe_i32 %0, %4, %5, 8\n"
                "v_add_u32 %0, %0, %1\n"
                "v_max_i32 %3, %1, %2\n"
                "v_add_u32 %3, %3, %6\n"
                "v_max3_i32 %2, %0, %3, 0\n")
                : "+v"(a), "+v"(c), "+v"(d), "+v"(e) : "v"(b), "v"(f), "v"(g));
        }
        if (V == 14) {     // one R=3 step of the mode-1 sweep as generated: chain dpp -> 3 x (max, add, max3)
            asm volatile(REP16(
                "v_mov_b32_dpp %[n], %[h2] wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                "v_dot4_i32_i8 %[a0], %[q], %[rb], %[d]\n"
                "v_max_i32 %[t], %[n], %[h0]\n"
                "v_dot4_i32_i8 %[a1], %[q], %[rb], %[h0]\n"
                "v_add_u32 %[t], %[gap], %[t]\n"
                "v_dot4_i32_i8 %[a2], %[q], %[rb], %[h1]\n"
                "v_max3_i32 %[h0], %[a0], %[t], 0\n"
                "v_max_i32 %[t], %[h0], %[h1]\n"
                "v_add_u32 %[t], %[gap], %[t]\n"
                "v_max3_i32 %[h1], %[a1], %[t], 0\n"
                "v_max_i32 %[t], %[h1], %[h2]\n"
                "v_add_u32 %[t], %[gap], %[t]\n"
                "v_max3_i32 %[h2], %[a2], %[t], 0\n"
                "v_max3_i32 %[lm], %[lm], %[h0], %[h1]\n"
                "v_mov_b32_dpp %[rb], %[rb] wave_shr:1 row_mask:0xf bank_mask:0xf\n"
                "v_mov_b32 %[d], %[n]\n")
                : [h0] "+v"(a), [h1] "+v"(c), [h2] "+v"(d), [n] "+v"(e), [t] "+v"(f), [a0] "+v"(g), [a1] "+v"(h), [a2] "+v"(acc), [lm] "+v"(lm), [rb] "+v"(rb), [d] "+v"(dg)
                : [q] "v"(b), [gap] "v"(gp));
        }
        if (V == 15) {     // same step, row chain shortened to (add, max) per row: a'' = max3(a, left + gap, 0) is off the chain
            asm volatile(REP16(
                "v_mov_b32_dpp %[n], %[h2] wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                "v_dot4_i32_i8 %[a0], %[q], %[rb], %[d]\n"
                "v_add_u32 %[l0], %[gap], %[h0]\n"
                "v_dot4_i32_i8 %[a1], %[q], %[rb], %[h0]\n"
                "v_add_u32 %[l1], %[gap], %[h1]\n"
                "v_dot4_i32_i8 %[a2], %[q], %[rb], %[h1]\n"
                "v_add_u32 %[l2], %[gap], %[h2]\n"
                "v_max3_i32 %[a0], %[a0], %[l0], 0\n"
                "v_add_u32 %[t], %[gap], %[n]\n"
                "v_max3_i32 %[a1], %[a1], %[l1], 0\n"
                "v_max_i32 %[h0], %[a0], %[t]\n"
                "v_max3_i32 %[a2], %[a2], %[l2], 0\n"
                "v_add_u32 %[t], %[gap], %[h0]\n"
                "v_max_i32 %[h1], %[a1], %[t]\n"
                "v_add_u32 %[t], %[gap], %[h1]\n"
                "v_max_i32 %[h2], %[a2], %[t]\n"
                "v_max3_i32 %[lm], %[lm], %[h0], %[h1]\n"
                "v_mov_b32_dpp %[rb], %[rb] wave_shr:1 row_mask:0xf bank_mask:0xf\n"
                "v_mov_b32 %[d], %[n]\n")
                : [h0] "+v"(a), [h1] "+v"(c), [h2] "+v"(d), [n] "+v"(e), [t] "+v"(f), [a0] "+v"(g), [a1] "+v"(h), [a2] "+v"(acc), [lm] "+v"(lm), [rb] "+v"(rb), [d] "+v"(dg),
                  [l0] "+v"(l0), [l1] "+v"(l1), [l2] "+v"(l2)
                : [q] "v"(b), [gap] "v"(gp));
        }
        if (V == 18) { asm volatile(REP16("v_dot8_i32_i4 %0, %4, %5, %0\n v_dot8_i32_i4 %1, %4, %5, %1\n v_dot8_i32_i4 %2, %4, %5, %2\n v_dot8_i32_i4 %3, %4, %5, %3\n")
                                    : "+v"(a), "+v"(c), "+v"(d), "+v"(e) : "v"(b), "v"(f)); }
        if (V == 16) {     // VOP3 sources all in ONE VGPR bank (index % 4 equal)
            asm volatile(REP16("v_max3_i32 v40, v44, v48, v52\n v_max3_i32 v41, v45, v49, v53\n v_dot4_i32_i8 v42, v46, v50, v54\n v_max3_i32 v43, v47, v51, v55\n")
                         ::: "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55");
        }
        if (V == 17) {     // ... in three different banks
            asm volatile(REP16("v_max3_i32 v40, v44, v49, v54\n v_max3_i32 v41, v45, v50, v55\n v_dot4_i32_i8 v42, v46, v51, v52\n v_max3_i32 v43, v47, v48, v53\n")
                         ::: "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55");
        }
        if (V == 11) { int sa = it; asm volatile(REP64("s_add_u32 %0, %0, 3\n") : "+s"(sa)); a += sa; }
        if (V == 12) { int sa = it; asm volatile(REP16("s_add_u32 %0, %0, 3\n v_add_u32 %1, %1, %2\n s_add_u32 %0, %0, 5\n v_add_u32 %1, %1, %2\n") : "+s"(sa), "+v"(a) : "v"(b)); a += sa; }
        if (V == 13) { int sa = it; asm volatile(REP16("s_cmp_lg_u32 %0, 77\n s_cbranch_scc0 1f\n1:\n v_add_u32 %1, %1, %2\n s_add_u32 %0, %0, 5\n") : "+s"(sa), "+v"(a) : "v"(b) : "scc"); a += sa; }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + c + d + e + (int)acc + f + g + h + lm + rb + dg + l0 + l1 + l2;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

__global__ void dpp_map(int *out) {
    int v = threadIdx.x + 100;
    int r = __builtin_amdgcn_update_dpp(-1, v, 0x138, 0xf, 0xf, false);
    int r2 = __builtin_amdgcn_update_dpp(-1, v, 0x138, 0xf, 0xf, true);
    out[threadIdx.x] = r; out[64 + threadIdx.x] = r2;
}

template <int V> void run(const char *name, int instr_per_iter, int *dout, long long *dcyc) {
    const int iters = 2000;
    for (int wps = 1; wps <= 8; wps *= 2) {       // waves per SIMD
        int blocks = 256 * wps;                   // 256 CUs x (4 waves = 1 per SIMD) x wps
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, dout, 10, dcyc);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, dout, iters, dcyc);
        CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<long long> cyc(blocks);
        CK(hipMemcpy(cyc.data(), dcyc, blocks * sizeof(long long), hipMemcpyDeviceToHost));
        double mean = 0; for (auto c : cyc) mean += c; mean /= blocks;
        double n = (double)iters * instr_per_iter;
        printf("%-28s waves/SIMD=%d  wall=%.3f ms  memtime-ticks/instr/wave=%.3f  wave-instr/s=%.3e  (chip: %.2f T lane-ops/s)\n",
               name, wps, ms, mean / n, blocks * 4.0 * n / (ms * 1e-3), blocks * 4.0 * n * 64 / (ms * 1e-3) / 1e12);
    }
}

int main() {
    int *dout; long long *dcyc;
    CK(hipMalloc(&dout, 256 * 8 * 256 * sizeof(int))); CK(hipMalloc(&dcyc, 256 * 8 * sizeof(long long)));
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    printf("device %s CUs=%d clock=%d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
    hipLaunchKernelGGL(dpp_map, dim3(1), dim3(64), 0, 0, dout);
    int h[128]; CK(hipMemcpy(h, dout, sizeof h, hipMemcpyDeviceToHost));
    printf("dpp wave_shr:1 bound_ctrl=0: lane0=%d lane1=%d lane15=%d lane16=%d lane31=%d lane32=%d lane63=%d\n", h[0], h[1], h[15], h[16], h[31], h[32], h[63]);
    printf("dpp wave_shr:1 bound_ctrl=1: lane0=%d lane1=%d lane16=%d lane32=%d\n", h[64], h[65], h[80], h[96]);
    run<0>("v_add dep chain", 64, dout, dcyc);
    run<1>("v_add 4 indep chains", 64, dout, dcyc);
    run<2>("v_mov_dpp wave_shr dep", 64, dout, dcyc);
    run<3>("cmp+2add+addc", 64, dout, dcyc);
    run<4>("v_max3 dep", 64, dout, dcyc);
    run<5>("v_bfe_i32 dep", 64, dout, dcyc);
    run<6>("cmp,s_nop1,addc,add", 64, dout, dcyc);
    run<7>("DP-cell mix (10 instr)", 160, dout, dcyc);
    run<8>("v_dot4_i32_i8 4 chains", 64, dout, dcyc);
    run<9>("cell dot4 (4 VALU + nop)", 80, dout, dcyc);
    run<10>("cell bfe+add (5 VALU)", 80, dout, dcyc);
    run<11>("s_add dep chain", 64, dout, dcyc);
    run<12>("s_add / v_add alternating", 64, dout, dcyc);
    run<13>("s_cmp, branch(not taken), v_add, s_add", 64, dout, dcyc);
    run<18>("v_dot8_i32_i4 4 chains", 64, dout, dcyc);
    run<16>("VOP3, 3 sources in one VGPR bank", 64, dout, dcyc);
    run<17>("VOP3, 3 sources in 3 VGPR banks", 64, dout, dcyc);
    run<14>("sweep step R=3 (16 instr, chain 10)", 16 * 16, dout, dcyc);
    run<15>("sweep step R=3 short chain (19 instr, chain 7)", 16 * 19, dout, dcyc);
    return 0;
}
