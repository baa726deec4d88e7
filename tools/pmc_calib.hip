// pmc_calib.hip -- known-byte-count kernels in the fill kernel's access pattern, to calibrate the gfx950
// FETCH_SIZE / WRITE_SIZE counters (MI355X_MICROARCH.md: only 16 B/lane streams are calibrated there).
//   write_dwords: every wave stores `iters` x 256 contiguous bytes as one dword per lane (the direction-field store)
//   read_dwords : every wave loads  `iters` x 256 contiguous bytes as one dword per lane (the traceback's tile load)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

extern "C" __global__ void write_dwords(unsigned *p, int iters) {
    size_t w = (size_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
    unsigned *q = p + w * (size_t)iters * 64 + (threadIdx.x & 63);
    for (int i = 0; i < iters; ++i) q[(size_t)i * 64] = (unsigned)i * 2654435761u + threadIdx.x;
}
extern "C" __global__ void read_dwords(const unsigned *p, unsigned *out, int iters) {
    size_t w = (size_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
    const unsigned *q = p + w * (size_t)iters * 64 + (threadIdx.x & 63);
    unsigned acc = 0;
    for (int i = 0; i < iters; ++i) acc ^= q[(size_t)i * 64];
    if (acc == 0x12345678u) out[0] = acc;
}

int main() {
    const int blocks = 4096, threads = 256, iters = 384;           // 4096*4 waves * 384 * 256 B = 1.61 GB (> Infinity Cache)
    const size_t bytes = (size_t)blocks * (threads / 64) * iters * 256;
    unsigned *p, *out;
    CK(hipMalloc(&p, bytes)); CK(hipMalloc(&out, 4));
    hipLaunchKernelGGL(write_dwords, dim3(blocks), dim3(threads), 0, 0, p, iters);
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(read_dwords, dim3(blocks), dim3(threads), 0, 0, p, out, iters);
    CK(hipDeviceSynchronize());
    printf("calibration bytes per kernel: %zu\n", bytes);
    return 0;
}
