// ubench_dispatch.hip -- what a launch costs before and after its work: kernels that do (almost) nothing, in the shapes of the
// sweep (250 workgroups x 4 wavefronts, 133 VGPRs) and of the traceback (1000 x 4 wavefronts, 116 VGPRs, 19 KB of LDS each).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_dispatch.hip -o tools/_bin/ubench_dispatch && tools/_bin/ubench_dispatch
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int VGPRS>
__global__ void __launch_bounds__(256) nop_kernel(int *out, int spin) {
    extern __shared__ int lds[];
    if (VGPRS >= 116) asm volatile("v_mov_b32 v115, 0" ::: "v115");
    if (VGPRS >= 133) asm volatile("v_mov_b32 v132, 0" ::: "v132");
    int acc = 0;
    for (int i = 0; i < spin; ++i) asm volatile("v_add_u32 %0, %0, 1" : "+v"(acc));
    if (acc == 0x7fffffff) { lds[threadIdx.x] = acc; out[0] = lds[0]; }
}

template <int VGPRS> void run(const char *name, int groups, size_t lds, int spin) {
    int *out; CK(hipMalloc(&out, 64));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(nop_kernel<VGPRS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; i++) nop_kernel<VGPRS><<<groups, 256, lds>>>(out, spin);
    float sum = 0;
    for (int i = 0; i < 20; i++) {
        CK(hipEventRecord(e0));
        nop_kernel<VGPRS><<<groups, 256, lds>>>(out, spin);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); sum += ms;
    }
    printf("%-62s %5d workgroups, %6zu B LDS, %5d instructions of work: %.2f us per launch\n", name, groups, lds, spin, sum / 20 * 1e3);
    CK(hipFree(out));
}

int main() {
    run<32>("small kernel", 1, 0, 0);
    run<133>("shape of sw_sweep_winmax_kernel (133 VGPRs)", 250, 0, 0);
    run<116>("shape of sw_traceback_winmax_kernel (116 VGPRs, 19 KB)", 1000, 19 * 1024, 0);
    run<116>("  the same with 8 wavefronts' worth of workgroups less", 500, 19 * 1024, 0);
    run<116>("  the same, every wavefront 10000 instructions", 1000, 19 * 1024, 10000);
    run<116>("  500 workgroups, 10000 instructions", 500, 19 * 1024, 10000);
    run<133>("sweep shape, every wavefront 10000 instructions", 250, 0, 10000);
    return 0;
}
