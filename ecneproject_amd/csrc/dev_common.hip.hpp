// dev_common.hip.hpp — device-side helpers shared by every kernel: lane / wave ids, 256-bit loads and shuffles, agent-scope loads.
// Part of the gfx950 device code of libecne_hip (see kernels.hip.hpp for the overview).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "engine_types.hpp"
#define ECNE_FINE_TICKS 1   // in-kernel phase clocks (measured: no effect on the solve time)
#include "fp256.hpp"

namespace ecne {

#define ECNE_WG 512
#define ECNE_NWAVES (ECNE_WG / 64)

// dynamic LDS of k_solve: mutable state of single-workgroup jobs (k_solve "LDS residency", chain.hip.hpp)
extern __shared__ __align__(16) unsigned char ecne_dyn_lds[];

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }
__device__ __forceinline__ uint64_t lanes_below() { return (1ull << lane_id()) - 1ull; }
__device__ __forceinline__ void wg_fence() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); }

__device__ __forceinline__ fp::u256 ld256(const uint64_t* p) { return fp::make(p[0], p[1], p[2], p[3]); }
__device__ __forceinline__ void st256(uint64_t* p, const fp::u256& v) {
    p[0] = v.w[0]; p[1] = v.w[1]; p[2] = v.w[2]; p[3] = v.w[3];
}
__device__ __forceinline__ fp::u256 shfl256(const fp::u256& v, int src) {
    fp::u256 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        unsigned lo = __shfl((unsigned)(v.w[i] & 0xffffffffu), src, 64);
        unsigned hi = __shfl((unsigned)(v.w[i] >> 32), src, 64);
        r.w[i] = ((uint64_t)hi << 32) | lo;
    }
    return r;
}

// abs(flip_coeffs(x)) of rule R7 (:1245-1259): values above the literal threshold are taken as
// negative numbers. The literal is NOT p-1 (it is ~1e75 below it) and is kept exactly.
__device__ __forceinline__ fp::u256 r7_abs(const fp::u256& c) {
    const fp::u256 thr = fp::make(0x43e1f593f0000000ULL, 0x9c41be16bb2a8891ULL, 0x045fcd3eea44076aULL,
                                  0x2e2e53955f6f1dfeULL);
    if (fp::cmp(c, thr) > 0) {
        fp::u256 t;
        fp::sub_raw(t, fp::modulus(), c);
        return t;
    }
    return c;
}

// exclusive prefix sum over the 64 lanes of a wavefront; *total = the sum
__device__ __forceinline__ uint32_t wave_excl_scan(uint32_t x, uint32_t* total) {
    uint32_t incl = x;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(incl, d, 64); if (lane_id() >= d) incl += y; }
    *total = __shfl(incl, 63, 64);
    return incl - x;
}

__device__ __forceinline__ uint32_t ld_agent(const uint32_t* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace ecne
