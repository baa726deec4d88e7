// ubench_occ.hip -- aggregate VALU issue rate of one SIMD against the number of waves resident on it (gfx950).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_occ.hip -o tools/_bin/ubench_occ && tools/_bin/ubench_occ
// The instruction mix is the sweep's row chain (v_dot8_i32_i4 / v_max3_i32 / v_sub_u32 clamp, dependent through the row);
// every wave runs the same stream; prints cycles per instruction per wave and per SIMD (s_memtime).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
#define REP16(x) x x x x x x x x x x x x x x x x
#define CLOB "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55"

template <int V>
__global__ void __launch_bounds__(1024) k(int *out, int iters, long long *cyc) {
    asm volatile("v_mov_b32 v40, 1\n v_mov_b32 v41, 2\n v_mov_b32 v42, 3\n v_mov_b32 v43, 4\n v_mov_b32 v44, 5\n v_mov_b32 v45, 6\n v_mov_b32 v46, 7\n v_mov_b32 v47, 8\n"
                 "v_mov_b32 v48, 1\n v_mov_b32 v49, 2\n v_mov_b32 v50, 3\n v_mov_b32 v51, 4\n v_mov_b32 v52, 5\n v_mov_b32 v53, 6\n v_mov_b32 v54, 7\n v_mov_b32 v55, 8\n" ::: CLOB);
    __syncthreads();
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (V == 0) asm volatile(REP16("v_dot8_i32_i4 v44, v50, v51, v48\n v_max3_i32 v40, v44, v41, v46\n v_sub_u32_e64 v41, v40, s4 clamp\n"
                           "v_dot8_i32_i4 v45, v50, v52, v49\n v_max3_i32 v42, v45, v41, v47\n v_sub_u32_e64 v43, v42, s4 clamp\n") ::: CLOB, "s4");
        // which instruction draws the power?  the same count of one kind each
        if (V == 1) asm volatile(REP16("v_max3_i32 v44, v50, v51, v48\n v_max3_i32 v40, v44, v41, v46\n v_max3_i32 v41, v40, v47, v49\n"
                           "v_max3_i32 v45, v50, v52, v49\n v_max3_i32 v42, v45, v41, v47\n v_max3_i32 v43, v42, v46, v48\n") ::: CLOB, "s4");
        if (V == 2) asm volatile(REP16("v_dot8_i32_i4 v44, v50, v51, v48\n v_dot8_i32_i4 v40, v44, v41, v46\n v_dot8_i32_i4 v41, v40, v47, v49\n"
                           "v_dot8_i32_i4 v45, v50, v52, v49\n v_dot8_i32_i4 v42, v45, v41, v47\n v_dot8_i32_i4 v43, v42, v46, v48\n") ::: CLOB, "s4");
        if (V == 3) asm volatile(REP16("v_add_u32_e32 v44, v50, v51\n v_add_u32_e32 v40, v44, v41\n v_add_u32_e32 v41, v40, v47\n"
                           "v_add_u32_e32 v45, v50, v52\n v_add_u32_e32 v42, v45, v41\n v_add_u32_e32 v43, v42, v46\n") ::: CLOB, "s4");
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
    int r;
    asm volatile("v_add_u32 %0, v40, v41\n v_add_u32 %0, %0, v42\n v_add_u32 %0, %0, v43" : "=v"(r) :: CLOB);
    if (r == 0x7fffffff) out[0] = r;
}

template <int V> void run(const char *name);
int main() {
    run<0>("the sweep's row chain: v_dot8_i32_i4 / v_max3_i32 / v_sub_u32 clamp");
    run<1>("v_max3_i32 only");
    run<2>("v_dot8_i32_i4 only");
    run<3>("v_add_u32 (VOP2) only");
    return 0;
}
template <int V> void run(const char *name) {
    printf("%s\n", name);
    int *out; long long *cyc;
    CK(hipMalloc(&out, 64)); CK(hipMalloc(&cyc, 8 * 16 * 1024));
    const int iters = 1000;
    struct { int groups, threads, per_simd; } cfg[] = {{256, 256, 1}, {256, 512, 2}, {256, 768, 3}, {256, 1024, 4}, {512, 768, 6}, {512, 1024, 8}};
    for (auto c : cfg) {
        k<V><<<c.groups, c.threads>>>(out, 10, cyc);
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0));
        k<V><<<c.groups, c.threads>>>(out, iters, cyc);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        static long long h[16 * 1024];
        CK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
        double s = 0; int nw = c.threads / 64;
        for (int g = 0; g < c.groups; g++) for (int w = 0; w < nw; w++) s += h[g * 16 + w];
        const double per_wave = s / (c.groups * nw) / (iters * 96.0);
        printf("  %d waves per SIMD: %.2f cycles per instruction per wave, %.2f per SIMD; kernel %.3f ms = %.2f G wave-instr/s per SIMD\n",
               c.per_simd, per_wave, per_wave / c.per_simd, ms, iters * 96.0 * c.per_simd / (ms * 1e-3) / 1e9);
    }
    CK(hipFree(out)); CK(hipFree(cyc));
}
