// Developer microbenchmark (not part of the product): dependent-load latency on MI355X.
// One lane chases a random cyclic permutation through buffers of several sizes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <numeric>
#include <algorithm>
#include <random>
__global__ void chase(const uint32_t* p, uint32_t start, int hops, uint32_t* out, unsigned long long* ticks, int mode) {
    uint32_t i = start;
    unsigned long long t0 = wall_clock64();
    for (int h = 0; h < hops; ++h) {
        if (mode == 0) i = p[i];
        else i = __hip_atomic_load(&p[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    unsigned long long t1 = wall_clock64();
    *out = i; *ticks = t1 - t0;
}
// barrier cost: 512 threads, N syncthreads
__global__ void bars(int n, unsigned long long* ticks) {
    unsigned long long t0 = wall_clock64();
    for (int i = 0; i < n; ++i) __syncthreads();
    unsigned long long t1 = wall_clock64();
    if (threadIdx.x == 0) *ticks = t1 - t0;
}
// atomicMin (no return) followed by an agent-scope load of another word + syncthreads
__global__ void atom(uint32_t* a, int n, unsigned long long* ticks) {
    unsigned long long t0 = wall_clock64();
    uint32_t acc = 0;
    for (int i = 0; i < n; ++i) {
        atomicMin(&a[(threadIdx.x * 977 + i * 131) & 0xFFFFF], (uint32_t)i);
        __syncthreads();
        acc += __hip_atomic_load(&a[(threadIdx.x * 613 + i * 17) & 0xFFFFF], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
    }
    unsigned long long t1 = wall_clock64();
    if (threadIdx.x == 0) { *ticks = t1 - t0; a[0] = acc; }
}
template <int MODE>
__global__ void phase(uint32_t* a, int n, unsigned long long* ticks) {
    unsigned long long t0 = wall_clock64();
    uint32_t acc = 0;
    for (int i = 0; i < n; ++i) {
        uint32_t* q = &a[(threadIdx.x * 977u + i * 131071u) & 0xFFFFFFu];
        if (MODE == 0) atomicMin(q, (uint32_t)i);
        if (MODE == 1) acc += __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (MODE == 2) *q = i;
        if (MODE == 3) acc += *q;
        if (MODE == 4) acc += atomicMin(q, (uint32_t)i);
        if (MODE == 5) { acc += *q; acc += a[(acc + threadIdx.x * 31u) & 0xFFFFFFu]; }   // two dependent loads
        __syncthreads();
    }
    unsigned long long t1 = wall_clock64();
    if (threadIdx.x == 0) { *ticks = t1 - t0; a[0] = acc; }
}
// gather throughput: every lane issues 32 INDEPENDENT scattered 4-byte loads per iteration
__global__ void gather(const uint32_t* a, int iters, uint32_t span_mask, unsigned long long* ticks, uint32_t* out) {
    uint32_t acc = 0, x = threadIdx.x * 2654435761u + 12345u;
    __syncthreads();
    unsigned long long t0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        uint32_t v[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) { x = x * 1664525u + 1013904223u; v[i] = a[(x >> 4) & span_mask]; }
#pragma unroll
        for (int i = 0; i < 32; ++i) acc += v[i];
    }
    __syncthreads();
    unsigned long long t1 = wall_clock64();
    if (threadIdx.x == 0) *ticks = t1 - t0;
    out[threadIdx.x] = acc;
}
int main() {
    unsigned long long* dt; uint32_t* dout;
    hipMalloc(&dt, 8); hipMalloc(&dout, 4);
    for (size_t mb : {1, 4, 32, 128, 512, 2048}) {
        size_t n = mb * 1024 * 1024 / 4;
        // stride-based cyclic walk with a large odd multiplier (cheap pseudo-random permutation cycle)
        std::vector<uint32_t> h(n);
        // single cycle: i -> (i + step) mod n with step coprime to n, scaled to jump far
        // full-period LCG over a power-of-two domain: a random-looking single cycle
        for (size_t i = 0; i < n; ++i) h[i] = (uint32_t)((i * 1664525ull + 1013904223ull) & (n - 1));
        uint32_t* d; hipMalloc(&d, n * 4); hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
        for (int mode = 0; mode < 2; ++mode) {
            int hops = 20000;
            chase<<<1, 1>>>(d, 0, hops, dout, dt, mode); hipDeviceSynchronize();
            chase<<<1, 1>>>(d, 7, hops, dout, dt, mode); hipDeviceSynchronize();
            unsigned long long t; hipMemcpy(&t, dt, 8, hipMemcpyDeviceToHost);
            printf("chase %4zu MB mode %d: %.1f ns/hop\n", mb, mode, t * 10.0 / hops);
        }
        hipFree(d);
    }
    bars<<<1, 512>>>(10000, dt); hipDeviceSynchronize();
    unsigned long long t; hipMemcpy(&t, dt, 8, hipMemcpyDeviceToHost);
    printf("syncthreads(512 threads): %.1f ns each\n", t * 10.0 / 10000);
    uint32_t* a; hipMalloc(&a, 4 << 20); hipMemset(a, 0xFF, 4 << 20);
    atom<<<1, 512>>>(a, 2000, dt); hipDeviceSynchronize();
    hipMemcpy(&t, dt, 8, hipMemcpyDeviceToHost);
    printf("atomicMin + sync + sc1 load + sync: %.1f ns per iteration\n", t * 10.0 / 2000);
    {
        uint32_t* g; hipMalloc(&g, 256 << 20); hipMemset(g, 1, 256 << 20);
        uint32_t* go; hipMalloc(&go, 4096);
        for (uint32_t span_mb : {1u, 64u, 256u})
            for (int threads : {64, 192, 512}) {
                int iters = 200;
                gather<<<1, threads>>>(g, iters, span_mb * 262144u - 1u, dt, go); hipDeviceSynchronize();
                gather<<<1, threads>>>(g, iters, span_mb * 262144u - 1u, dt, go); hipDeviceSynchronize();
                hipMemcpy(&t, dt, 8, hipMemcpyDeviceToHost);
                double ns = t * 10.0;
                printf("gather span %3u MB, %3d threads: %.1f ns per wave-load-instruction per wave (%.2f ns per lane-load overall)\n",
                       span_mb, threads, ns / (iters * 32.0), ns / (iters * 32.0 * threads));
            }
    }
    uint32_t* b; hipMalloc(&b, 64 << 20); hipMemset(b, 0xFF, 64 << 20);
    const char* names[] = {"atomicMin(no ret)", "sc1 load", "plain store", "plain load", "atomicMin(ret)", "2 dependent plain loads"};
    for (int threads : {64, 512}) {
    for (int m = 0; m < 6; ++m) {
        int n = 2000;
        switch (m) {
            case 0: phase<0><<<1, threads>>>(b, n, dt); break;
            case 1: phase<1><<<1, threads>>>(b, n, dt); break;
            case 2: phase<2><<<1, threads>>>(b, n, dt); break;
            case 3: phase<3><<<1, threads>>>(b, n, dt); break;
            case 4: phase<4><<<1, threads>>>(b, n, dt); break;
            case 5: phase<5><<<1, threads>>>(b, n, dt); break;
        }
        hipDeviceSynchronize();
        hipMemcpy(&t, dt, 8, hipMemcpyDeviceToHost);
        printf("%3d threads: %-26s + syncthreads: %.1f ns per iteration\n", threads, names[m], t * 10.0 / n);
    }}
    return 0;
}
