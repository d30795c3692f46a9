// Developer aid (GPU box): is the per-process bimodality of k_classify_rows (46 us in some processes, 55-62 us in others)
// a property of the kernel or of the platform? A plain streaming kernel at the same scale -- 123 MB read at 32 B per lane, 31 MB
// written at 8 B per lane, the classification kernel's traffic -- timed the same way (median / best of 9 warm launches, HIP events),
// one line per process: run it a dozen times (tools/gp_bimodal.sh).
// build: hipcc --offload-arch=gfx950 -O2 tools/micro/stream154.hip -o tools/micro/stream154
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
__global__ __launch_bounds__(256) void k_read32(const ulonglong4* __restrict__ a, unsigned long long* __restrict__ b, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const ulonglong4 v = a[i]; b[i] = v.x ^ v.y ^ v.z ^ v.w; }
}
int main() {
    const size_t n = 3854208;                 // entries: 123.3 MB read, 30.8 MB written
    void *a, *b;
    if (hipMalloc(&a, n * 32) != hipSuccess || hipMalloc(&b, n * 8) != hipSuccess) return 1;
    (void)hipMemset(a, 1, n * 32); (void)hipMemset(b, 0, n * 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float t[9];
    for (int rep = 0; rep < 11; ++rep) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_read32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, (const ulonglong4*)a, (unsigned long long*)b, n);
        (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 2) t[rep - 2] = ms;
    }
    std::sort(t, t + 9);
    printf("stream 154 MB: median %.1f us best %.1f us -> %.0f GB/s (median), a at %p b at %p\n", t[4] * 1e3, t[0] * 1e3, n * 40.0 / t[4] / 1e6, a, b);
    return 0;
}
