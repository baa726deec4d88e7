// ubench_banks.hip -- does the VGPR bank of a VOP3's sources matter for ONE wave per SIMD on gfx950?
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_banks.hip -o /tmp/ubench_banks && /tmp/ubench_banks
// Each variant issues 64 x 16 instructions per iteration on explicit physical registers (v40..v63), one wave per SIMD
// (256 workgroups of 256 threads) and one wave on the chip; prints cycles per instruction (s_memtime).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
#define REP16(x) x x x x x x x x x x x x x x x x
#define CLOB "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63"

template <int V>
__global__ void __launch_bounds__(256) k(int *out, int iters, long long *cyc) {
    asm volatile("v_mov_b32 v40, 1\n v_mov_b32 v41, 2\n v_mov_b32 v42, 3\n v_mov_b32 v43, 4\n v_mov_b32 v44, 5\n v_mov_b32 v45, 6\n v_mov_b32 v46, 7\n v_mov_b32 v47, 8\n"
                 "v_mov_b32 v48, 1\n v_mov_b32 v49, 2\n v_mov_b32 v50, 3\n v_mov_b32 v51, 4\n v_mov_b32 v52, 5\n v_mov_b32 v53, 6\n v_mov_b32 v54, 7\n v_mov_b32 v55, 8\n"
                 "v_mov_b32 v56, 1\n v_mov_b32 v57, 2\n v_mov_b32 v58, 3\n v_mov_b32 v59, 4\n v_mov_b32 v60, 5\n v_mov_b32 v61, 6\n v_mov_b32 v62, 7\n v_mov_b32 v63, 8\n" ::: CLOB);
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        // max3, independent destinations (4 of them round robin), sources in three different banks / all in one bank
        if (V == 0) asm volatile(REP16("v_max3_i32 v40, v45, v46, v47\n v_max3_i32 v41, v49, v50, v51\n v_max3_i32 v42, v53, v54, v55\n v_max3_i32 v43, v57, v58, v59\n") ::: CLOB);
        if (V == 1) asm volatile(REP16("v_max3_i32 v40, v44, v48, v52\n v_max3_i32 v41, v45, v49, v53\n v_max3_i32 v42, v46, v50, v54\n v_max3_i32 v43, v47, v51, v55\n") ::: CLOB);
        // dot8, same
        if (V == 2) asm volatile(REP16("v_dot8_i32_i4 v40, v45, v46, v47\n v_dot8_i32_i4 v41, v49, v50, v51\n v_dot8_i32_i4 v42, v53, v54, v55\n v_dot8_i32_i4 v43, v57, v58, v59\n") ::: CLOB);
        if (V == 3) asm volatile(REP16("v_dot8_i32_i4 v40, v44, v48, v52\n v_dot8_i32_i4 v41, v45, v49, v53\n v_dot8_i32_i4 v42, v46, v50, v54\n v_dot8_i32_i4 v43, v47, v51, v55\n") ::: CLOB);
        // two sources in one bank
        if (V == 4) asm volatile(REP16("v_max3_i32 v40, v44, v48, v53\n v_max3_i32 v41, v45, v49, v54\n v_max3_i32 v42, v46, v50, v55\n v_max3_i32 v43, v47, v51, v52\n") ::: CLOB);
        // VOP2 for reference
        if (V == 5) asm volatile(REP16("v_max_i32 v40, v45, v46\n v_max_i32 v41, v49, v50\n v_max_i32 v42, v53, v54\n v_max_i32 v43, v57, v58\n") ::: CLOB);
        if (V == 6) asm volatile(REP16("v_max_i32 v40, v44, v48\n v_max_i32 v41, v45, v49\n v_max_i32 v42, v46, v50\n v_max_i32 v43, v47, v51\n") ::: CLOB);
        // v_sub clamp (VOP3, one VGPR + one SGPR)
        if (V == 7) asm volatile(REP16("v_sub_u32_e64 v40, v45, s4 clamp\n v_sub_u32_e64 v41, v49, s4 clamp\n v_sub_u32_e64 v42, v53, s4 clamp\n v_sub_u32_e64 v43, v57, s4 clamp\n") ::: CLOB, "s4");
        // DPP forms
        if (V == 8) asm volatile(REP16("v_max_i32_dpp v40, v45, v46 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_add_u32_dpp v41, v49, v50 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
                                       "v_mov_b32_dpp v42, v53 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_lshlrev_b32_sdwa v43, v57, v58 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n") ::: CLOB);
        // dependent pairs: max3 -> sub clamp -> max3 ... (the row chain of the sweep)
        if (V == 9) asm volatile(REP16("v_max3_i32 v40, v45, v41, v46\n v_sub_u32_e64 v41, v40, s4 clamp\n v_max3_i32 v42, v49, v41, v50\n v_sub_u32_e64 v43, v42, s4 clamp\n") ::: CLOB, "s4");
        // code placement: the same 8-byte instructions starting at 0 mod 8 / at 4 mod 8 (one 4-byte s_nop in front)
        if (V == 10) asm volatile(".p2align 6\n" REP16("v_max3_i32 v40, v45, v46, v47\n v_max3_i32 v41, v49, v50, v51\n v_max3_i32 v42, v53, v54, v55\n v_max3_i32 v43, v57, v58, v59\n") ::: CLOB);
        if (V == 11) asm volatile(".p2align 6\n s_nop 0\n" REP16("v_max3_i32 v40, v45, v46, v47\n v_max3_i32 v41, v49, v50, v51\n v_max3_i32 v42, v53, v54, v55\n v_max3_i32 v43, v57, v58, v59\n") ::: CLOB);
        // one 4-byte instruction per 15 (the sweep's step at R = 3): the phase flips every step
        if (V == 12) asm volatile(".p2align 6\n" REP16("v_max3_i32 v40, v45, v46, v47\n v_max3_i32 v41, v49, v50, v51\n v_max3_i32 v42, v53, v54, v55\n v_max_i32_e32 v43, v57, v58\n") ::: CLOB);
        if (V == 13) asm volatile(".p2align 6\n" REP16("v_max3_i32 v40, v45, v46, v47\n v_max3_i32 v41, v49, v50, v51\n v_max3_i32 v42, v53, v54, v55\n v_max_i32_e64 v43, v57, v58\n") ::: CLOB);
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    int r;
    asm volatile("v_add_u32 %0, v40, v41\n v_add_u32 %0, %0, v42\n v_add_u32 %0, %0, v43" : "=v"(r) :: CLOB);
    if (r == 0x7fffffff) out[0] = r;
}

template <int V> void run(const char *name, int groups) {
    int *out; long long *cyc;
    CK(hipMalloc(&out, 64)); CK(hipMalloc(&cyc, 8 * 1024));
    const int iters = 2000;
    k<V><<<groups, 256>>>(out, 10, cyc);
    k<V><<<groups, 256>>>(out, iters, cyc);
    CK(hipDeviceSynchronize());
    long long h[1024];
    CK(hipMemcpy(h, cyc, 8 * groups, hipMemcpyDeviceToHost));
    double s = 0; for (int i = 0; i < groups; i++) s += h[i];
    printf("%-58s %4d workgroups: %.2f cycles per instruction\n", name, groups, s / groups / (iters * 64.0));
    CK(hipFree(out)); CK(hipFree(cyc));
}

int main() {
    for (int groups : {1, 256}) {
        run<0>("v_max3_i32, sources in 3 banks", groups);
        run<1>("v_max3_i32, sources in ONE bank", groups);
        run<4>("v_max3_i32, two sources in one bank", groups);
        run<2>("v_dot8_i32_i4, sources in 3 banks", groups);
        run<3>("v_dot8_i32_i4, sources in ONE bank", groups);
        run<5>("v_max_i32 (VOP2), 2 banks", groups);
        run<6>("v_max_i32 (VOP2), one bank", groups);
        run<7>("v_sub_u32 clamp (VGPR, SGPR)", groups);
        run<8>("DPP max / DPP add / DPP mov / SDWA shift mix", groups);
        run<9>("dependent chain max3 -> sub clamp -> max3 -> sub clamp", groups);
        run<10>("v_max3_i32 stream starting at 0 mod 64", groups);
        run<11>("v_max3_i32 stream starting at 4 mod 64", groups);
        run<12>("3 x v_max3 + 1 x v_max_i32_e32 (4 bytes): phase flips", groups);
        run<13>("3 x v_max3 + 1 x v_max_i32_e64 (8 bytes)", groups);
    }
    return 0;
}
