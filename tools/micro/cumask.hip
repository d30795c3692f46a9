// Which XCDs / CUs does a stream made with hipExtStreamCreateWithCUMask run on?  For each of a few masks: 512 workgroups of 512 threads
// that stay a little while (so that they spread), histogram of HW_REG_XCC_ID and of (SE, CU) from HW_REG_HW_ID.
// usage: ./cumask
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_where(uint32_t* out) {
    if (threadIdx.x == 0) {
        const uint32_t xcc = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20) & 15u;
        const uint32_t hw = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 4);      // HW_REG_HW_ID
        out[blockIdx.x] = (xcc << 28) | (hw & 0x0FFFFFFFu);
        const unsigned long long t0 = wall_clock64();
        while (wall_clock64() - t0 < 20000ull) __builtin_amdgcn_s_sleep(8);      // 0.2 ms
    }
    __syncthreads();
}
int main() {
    const int NB = 512;
    uint32_t* d;
    CK(hipMalloc(&d, 4 * NB));
    std::vector<uint32_t> h(NB);
    struct M { const char* name; uint32_t w[8]; } masks[] = {
        {"words 0 (bits 0-31)", {0xFFFFFFFFu, 0, 0, 0, 0, 0, 0, 0}},
        {"word 7 (bits 224-255)", {0, 0, 0, 0, 0, 0, 0, 0xFFFFFFFFu}},
        {"every 8th bit", {0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u}},
        {"bits with index % 8 != 7", {0x7F7F7F7Fu, 0x7F7F7F7Fu, 0x7F7F7F7Fu, 0x7F7F7F7Fu, 0x7F7F7F7Fu, 0x7F7F7F7Fu, 0x7F7F7F7Fu, 0x7F7F7F7Fu}},
        {"all", {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}},
    };
    for (auto& m : masks) {
        hipStream_t s;
        hipError_t e = hipExtStreamCreateWithCUMask(&s, 8, m.w);
        if (e != hipSuccess) { printf("%s: create failed: %s\n", m.name, hipGetErrorString(e)); (void)hipGetLastError(); continue; }
        CK(hipMemsetAsync(d, 0xFF, 4 * NB, s));
        hipLaunchKernelGGL(k_where, dim3(NB), dim3(512), 0, s, d);
        CK(hipStreamSynchronize(s));
        CK(hipMemcpy(h.data(), d, 4 * NB, hipMemcpyDeviceToHost));
        int xc[16] = {0};
        std::vector<int> seen(1 << 16, 0);
        int distinct = 0;
        for (int i = 0; i < NB; ++i) { xc[h[i] >> 28]++; const uint32_t key = ((h[i] >> 28) << 12) | ((h[i] >> 8) & 0xFFFu); if (!seen[key]++) ++distinct; }
        printf("%-28s XCC histogram:", m.name);
        for (int i = 0; i < 8; ++i) printf(" %d", xc[i]);
        printf("   distinct (xcc, hw_id[19:8]) places: %d\n", distinct);
        printf("    XCC of workgroups 0..39:");
        for (int i = 0; i < 40; ++i) printf(" %u", h[i] >> 28);
        printf("\n");
        CK(hipStreamDestroy(s));
    }
    return 0;
}
