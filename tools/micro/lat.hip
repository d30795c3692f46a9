// Developer microbenchmark (not part of the product): dependent-access latency of ONE wavefront on MI355X --
// LDS through ds_read and through a generic (flat) pointer, global memory L1- / L2-resident -- in shader
// clocks (s_memtime) and in ns (s_memrealtime, 100 MHz), with the rest of the chip idle or busy.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void spin(int iters, float* out) {   // keeps other CUs busy (clock ramp)
    float x = threadIdx.x;
    for (int i = 0; i < iters; ++i) x = x * 1.0001f + 0.5f;
    if (x == 123.f) *out = x;
}
template <int MODE>
__global__ void lat(const uint32_t* g, int hops, uint32_t n, unsigned long long* res) {
    __shared__ uint32_t s[8192];
    for (uint32_t i = threadIdx.x; i < 8192; i += blockDim.x) s[i] = (i * 1664525u + 1013904223u) & 8191u;
    __syncthreads();
    if (threadIdx.x >= 64) return;
    uint32_t i = threadIdx.x & 7;
    const uint32_t* fs = s;                          // generic pointer to LDS
    const uint32_t* volatile* hide = &fs; (void)hide;
    unsigned long long c0 = clock64(), t0 = wall_clock64();
    for (int h = 0; h < hops; ++h) {
        if (MODE == 0) i = s[i];
        if (MODE == 1) { const uint32_t* p = *hide; i = p[i]; }          // flat load, LDS aperture
        if (MODE == 2) i = g[i & (n - 1)];                                 // global (L1/L2 by n)
        if (MODE == 3) i = __hip_atomic_load(&g[i & (n - 1)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // past the L1
        if (MODE == 4) { s[(i + 1) & 8191] = i; __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); i = s[i]; }   // LDS store + fence + load
        if (MODE == 5) { ((uint32_t*)g)[(i + 64) & (n - 1)] = i; __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); i = g[i & (n - 1)]; }
    }
    unsigned long long c1 = clock64(), t1 = wall_clock64();
    if (threadIdx.x == 0) { res[0] = c1 - c0; res[1] = t1 - t0; res[2] = i; }
}
int main() {
    unsigned long long* dr; hipMalloc(&dr, 64);
    float* fo; hipMalloc(&fo, 4);
    const char* names[] = {"LDS ds_read", "LDS via flat pointer", "global plain", "global sc1 (agent)", "LDS store+fence+load", "global store+fence+load"};
    for (int busy = 0; busy < 2; ++busy) {
        hipStream_t s2; hipStreamCreate(&s2);
        for (uint32_t nkb : {16u, 1024u, 65536u}) {
            uint32_t n = nkb * 256;
            std::vector<uint32_t> h(n);
            for (uint32_t i = 0; i < n; ++i) h[i] = (uint32_t)((i * 1664525ull + 1013904223ull) & (n - 1));
            uint32_t* d; hipMalloc(&d, n * 4ull); hipMemcpy(d, h.data(), n * 4ull, hipMemcpyHostToDevice);
            for (int m = 0; m < 6; ++m) {
                if (m < 2 && nkb != 16) continue;
                if (m == 4 && nkb != 16) continue;
                int hops = 20000;
                for (int rep = 0; rep < 2; ++rep) {
                    if (busy) spin<<<1024, 256, 0, s2>>>(400000, fo);
                    switch (m) {
                        case 0: lat<0><<<1, 512>>>(d, hops, n, dr); break;
                        case 1: lat<1><<<1, 512>>>(d, hops, n, dr); break;
                        case 2: lat<2><<<1, 512>>>(d, hops, n, dr); break;
                        case 3: lat<3><<<1, 512>>>(d, hops, n, dr); break;
                        case 4: lat<4><<<1, 512>>>(d, hops, n, dr); break;
                        case 5: lat<5><<<1, 512>>>(d, hops, n, dr); break;
                    }
                    hipDeviceSynchronize();
                }
                unsigned long long r[3]; hipMemcpy(r, dr, 24, hipMemcpyDeviceToHost);
                printf("busy=%d %-26s span %6u KB: %7.1f clk/hop %7.1f ns/hop  (s_memtime/s_memrealtime = %.1f MHz)\n", busy, names[m], nkb,
                       (double)r[0] / hops, r[1] * 10.0 / hops, 100.0 * r[0] / r[1]);
            }
            hipFree(d);
        }
    }
    return 0;
}
