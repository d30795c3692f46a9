// abstract.hip.hpp — device side of abstraction() (reference :237-395): where in the main file could a trusted function sit?
// Part of the gfx950 device code of libecne_hip (see kernels.hip.hpp for the overview).
//
// The reference hashes every row (hash_r1cs_equation :228-235: the sorted non-zero coefficients of a, b, c) and slides
// the trusted function's hash list over the main file, comparing the first len-1 hashes (:259-270); windows that pass are
// verified exactly (per-part multisets, variable bijection) and replaced greedily. The O(rows) part of that -- one pass over
// every coefficient of the main file, then one comparison per start row -- runs here:
//   k_abs_fingerprint   one lane per row: an order-insensitive 64-bit fingerprint of the three per-part multisets of
//                       non-zero coefficients (a commutative sum of per-coefficient hashes, so nothing is sorted).
//                       Streams 32 B per coefficient + 24 B of row pointers per row: HBM-bound.
//   k_abs_weighted_scan / k_abs_scan_tops / k_abs_candidates
//                       a polynomial window hash over the fingerprints: with v_t = f_t * r^t (mod 2^64) and the exclusive
//                       prefix sums P, the window [i, i + m) matches the pattern's T = sum_j g_j * r^j iff
//                       P[i + m] - P[i] == T * r^i.  One scan and one comparison per row instead of m per row.
// Equal windows always produce equal values, so no occurrence is missed; what passes is only a CANDIDATE and is verified
// exactly on the host (host_model.hpp, unchanged), which also keeps the reference's greedy left-to-right replacement with
// its stuck cursor (:368-388). The result is therefore the host path's, window for window.
#pragma once
#include "dev_common.hip.hpp"

namespace ecne {

#define ECNE_ABS_R 0x9E3779B97F4A7C15ull      // odd: invertible mod 2^64, so distinct positions get distinct weights

__device__ __forceinline__ uint64_t abs_mix(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}
__device__ __forceinline__ uint64_t abs_pow(uint64_t t) {      // r^t mod 2^64
    uint64_t b = ECNE_ABS_R, acc = 1;
    while (t) { if (t & 1) acc *= b; b *= b; t >>= 1; }
    return acc;
}

struct AbsRows {      // rows in dictionary order as the reader left them (host_model.hpp Rows), coefficients only
    const uint64_t* ptr[3];      // nC + 1 offsets per part
    const uint64_t* coef[3];     // 4 limbs per entry
    uint64_t n;
};

// A workgroup takes 256 consecutive rows. Their coefficients are one contiguous range per part, so the range is streamed
// entry by entry -- lane = entry: a wavefront reads 64 x 32 contiguous bytes per load instruction, whatever the row lengths
// -- each entry is hashed where it lands and only the 8-byte hash is staged in LDS; then lane = row sums its own range out of
// LDS. A row longer than the tile (a 1 025-term sum) simply spans several tiles.
#define ECNE_ABS_TILE 2048
__global__ __launch_bounds__(256) void k_abs_fingerprint(AbsRows R, uint64_t* __restrict__ out) {
    __shared__ uint64_t s_h[ECNE_ABS_TILE];
    __shared__ uint64_t s_lo, s_hi;
    const uint64_t nblk = (R.n + 255) / 256;
    for (uint64_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const uint64_t r0 = blk * 256, r = r0 + threadIdx.x;
        const bool mine = r < R.n;
        uint64_t h = 0x1234567ull;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const uint64_t k0 = mine ? R.ptr[p][r] : 0, k1 = mine ? R.ptr[p][r + 1] : 0;
            if (threadIdx.x == 0) s_lo = k0;
            if (r + 1 == R.n || (mine && threadIdx.x == 255)) s_hi = k1;
            __syncthreads();
            const uint64_t lo = s_lo, hi = s_hi;
            __syncthreads();                 // (everybody has read the range before the next part overwrites it)
            uint64_t sum = 0, cnt = 0;
            for (uint64_t t0 = lo; t0 < hi; t0 += ECNE_ABS_TILE) {
                const uint64_t t1 = t0 + ECNE_ABS_TILE < hi ? t0 + ECNE_ABS_TILE : hi;
                for (uint64_t k = t0 + threadIdx.x; k < t1; k += 256) {
                    const ulonglong2* c = reinterpret_cast<const ulonglong2*>(R.coef[p] + 4 * k);
                    const ulonglong2 a = c[0], b = c[1];
                    // explicit zeros do not count (:232); a non-zero coefficient hashing to 0 would be dropped as well -- still a
                    // function of the multiset, which is all a fingerprint has to be
                    s_h[k - t0] = (a.x | a.y | b.x | b.y) == 0 ? 0ull : abs_mix(abs_mix(abs_mix(abs_mix(a.x) ^ a.y) ^ b.x) ^ b.y);
                }
                __syncthreads();
                const uint64_t a0 = k0 > t0 ? k0 : t0, a1 = k1 < t1 ? k1 : t1;
                for (uint64_t k = a0; k < a1; ++k) { const uint64_t x = s_h[k - t0]; sum += x; cnt += x != 0; }
                __syncthreads();
            }
            h = abs_mix(h ^ (sum + 0x9e3779b97f4a7c15ull * (cnt + 1) + (uint64_t)p));
        }
        if (mine) out[r] = h;
    }
}

// v_t = f_t * r^t, block-local exclusive scan (1024 elements per block), block totals to tops[]
__global__ __launch_bounds__(256) void k_abs_weighted_scan(const uint64_t* __restrict__ f, uint64_t n, uint64_t* __restrict__ P,
                                                           uint64_t* __restrict__ tops) {
    __shared__ uint64_t s_w[4];
    const uint64_t base = (uint64_t)blockIdx.x * 1024 + (uint64_t)threadIdx.x * 4;
    uint64_t v[4], run = 0;
    const uint64_t w0 = abs_pow(base);
    uint64_t w = w0;
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[i] = base + i < n ? f[base + i] * w : 0; w *= ECNE_ABS_R; }
    uint64_t mine = v[0] + v[1] + v[2] + v[3];
    // wave scan of the per-thread sums, then across the 4 waves
    uint64_t incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint64_t lo = (uint64_t)__shfl_up((unsigned)(incl & 0xffffffffu), d, 64), hi = (uint64_t)__shfl_up((unsigned)(incl >> 32), d, 64);
        if ((threadIdx.x & 63) >= d) incl += (hi << 32) | lo;
    }
    if ((threadIdx.x & 63) == 63) s_w[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint64_t wave_off = 0;
    for (unsigned k = 0; k < (threadIdx.x >> 6); ++k) wave_off += s_w[k];
    run = wave_off + incl - mine;
#pragma unroll
    for (int i = 0; i < 4; ++i) { if (base + i < n) P[base + i] = run; run += v[i]; }
    if (threadIdx.x == 255) tops[blockIdx.x] = run;
}
// exclusive scan of the block totals (one workgroup; a few thousand entries at most per pass)
__global__ __launch_bounds__(256) void k_abs_scan_tops(uint64_t* tops, uint64_t nb) {
    __shared__ uint64_t s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (uint64_t b0 = 0; b0 < nb; b0 += 256) {
        const uint64_t i = b0 + threadIdx.x;
        const uint64_t x = i < nb ? tops[i] : 0;
        uint64_t incl = x;
        __shared__ uint64_t s_w[4];
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint64_t lo = (uint64_t)__shfl_up((unsigned)(incl & 0xffffffffu), d, 64), hi = (uint64_t)__shfl_up((unsigned)(incl >> 32), d, 64);
            if ((threadIdx.x & 63) >= d) incl += (hi << 32) | lo;
        }
        if ((threadIdx.x & 63) == 63) s_w[threadIdx.x >> 6] = incl;
        __syncthreads();
        uint64_t off = s_carry;
        for (unsigned k = 0; k < (threadIdx.x >> 6); ++k) off += s_w[k];
        if (i < nb) tops[i] = off + incl - x;
        __syncthreads();
        if (threadIdx.x == 255) s_carry = off + incl;
        __syncthreads();
    }
}
// window [i, i + m) is a candidate iff (P[i + m] - P[i]) == T * r^i, with the global prefix = block prefix + tops
// (f, P hold n + 1 entries, f[n] = 0, so that P[n] is the sum of everything)
__global__ __launch_bounds__(256) void k_abs_candidates(const uint64_t* __restrict__ P, const uint64_t* __restrict__ tops, uint64_t n,
                                                        uint64_t m, uint64_t nS, uint64_t T,
                                                        uint32_t* __restrict__ cand, unsigned long long* __restrict__ ncand, uint64_t cap) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i + nS <= n; i += (uint64_t)gridDim.x * 256) {
        const uint64_t a = P[i] + tops[i >> 10];
        const uint64_t e = i + m;
        const uint64_t b = P[e] + tops[e >> 10];
        if (b - a == T * abs_pow(i)) {
            const unsigned long long pos = atomicAdd(ncand, 1ull);
            if (pos < cap) cand[pos] = (uint32_t)i;
        }
    }
}

}  // namespace ecne
