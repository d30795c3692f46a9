// Does a busy device clock lower? One workgroup runs a fixed dependent ALU chain and reads the shader clock (s_memtime) and the constant
// 100 MHz clock (s_memrealtime) around it -- alone, and next to 225 workgroups that spin on s_sleep + a poll of one word (what a team's helpers do).
// usage: ./clk
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ __launch_bounds__(512) void k_work(unsigned long long* out, volatile uint32_t* stop, int busy_kind) {
    if (blockIdx.x == 0) {
        if (threadIdx.x == 0) {
            const unsigned long long c0 = clock64(), w0 = wall_clock64();
            uint32_t x = 1;
            for (int i = 0; i < 4000000; ++i) x = x * 1664525u + 1013904223u;      // a dependent chain
            const unsigned long long c1 = clock64(), w1 = wall_clock64();
            out[0] = c1 - c0; out[1] = w1 - w0; out[2] = x;
            __hip_atomic_store((uint32_t*)stop, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    } else {
        // the others: polling helpers (busy_kind 1) or a tight ALU loop on every lane (busy_kind 2)
        if (busy_kind == 1) {
            if (threadIdx.x == 0) while (!__hip_atomic_load((uint32_t*)stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) __builtin_amdgcn_s_sleep(8);
        } else {
            uint32_t x = threadIdx.x;
            while (!__hip_atomic_load((uint32_t*)stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { for (int i = 0; i < 1000; ++i) x = x * 1664525u + 1013904223u; }
            if (x == 12345u) out[3] = x;
        }
    }
    __syncthreads();
}
int main() {
    unsigned long long* d; uint32_t* stop;
    CK(hipMalloc(&d, 64)); CK(hipMalloc(&stop, 4));
    struct { int grid, kind; const char* what; } cases[] = {{1, 0, "alone"}, {226, 1, "next to 225 polling workgroups"}, {226, 2, "next to 225 workgroups in an ALU loop"}};
    for (auto& c : cases) for (int rep = 0; rep < 2; ++rep) {
        CK(hipMemset(stop, 0, 4));
        hipLaunchKernelGGL(k_work, dim3(c.grid), dim3(512), 0, 0, d, (volatile uint32_t*)stop, c.kind);
        CK(hipDeviceSynchronize());
        unsigned long long h[3];
        CK(hipMemcpy(h, d, 24, hipMemcpyDeviceToHost));
        printf("%-40s chain: %8.3f ms, shader clock %7.1f MHz\n", c.what, h[1] / 1e5, (double)h[0] / ((double)h[1] / 100.0));
    }
    return 0;
}
