// Developer aid (GPU box): calibration of the rocprofv3 FETCH_SIZE / WRITE_SIZE counters on this stack with kernels that move a KNOWN
// number of bytes ("FETCH_SIZE can under-count wide streams 2x on gfx950 -- calibrate with a copy kernel").
//   k_copy16   every lane copies 16 bytes per step (dwordx4), grid-stride: N bytes read, N bytes written
//   k_read32   every lane reads 32 bytes per step (the classification kernel's access: one coefficient), writes 8 bytes per 32 read
// build: hipcc --offload-arch=gfx950 -O2 tools/micro/copy.hip -o tools/micro/copy ; run: rocprofv3 --pmc FETCH_SIZE --kernel-trace -- tools/micro/copy
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_copy16(const uint4* __restrict__ a, uint4* __restrict__ b, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
__global__ void k_read32(const ulonglong4* __restrict__ a, unsigned long long* __restrict__ b, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const ulonglong4 v = a[i];
        b[i] = v.x ^ v.y ^ v.z ^ v.w;
    }
}
int main() {
    const size_t N = (size_t)1 << 30;   // 1 GiB (beyond the 256 MiB Infinity Cache)
    void *a, *b;
    if (hipMalloc(&a, N) != hipSuccess || hipMalloc(&b, N) != hipSuccess) return 1;
    (void)hipMemset(a, 1, N); (void)hipMemset(b, 0, N);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_copy16, dim3(256 * 16), dim3(256), 0, 0, (const uint4*)a, (uint4*)b, N / 16);
        (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("k_copy16  %zu B read + %zu B written in %.3f ms = %.0f GB/s\n", N, N, ms, 2.0 * N / ms / 1e6);
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_read32, dim3(256 * 16), dim3(256), 0, 0, (const ulonglong4*)a, (unsigned long long*)b, N / 32);
        (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
        printf("k_read32  %zu B read + %zu B written in %.3f ms = %.0f GB/s\n", N, N / 4, ms, 1.25 * N / ms / 1e6);
    }
    return 0;
}
