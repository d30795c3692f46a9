// wg_tools.hip.hpp — workgroup scans, the P3 group-key / group-table helpers and the odd-permutation sum, workgroup-uniform error poll.
// Part of the gfx950 device code of libecne_hip (see kernels.hip.hpp for the overview).
#pragma once
#include "rules_wave.hip.hpp"

namespace ecne {

// ---------------------------------------------------------------------------------- workgroup tools
// exclusive prefix sum of one value per thread over the workgroup; returns the
// thread's offset, *total receives the sum. lds: ECNE_NWAVES + 1 words.
__device__ uint32_t wg_exclusive_scan(uint32_t x, uint32_t* lds, uint32_t* total) {
    const int lane = lane_id(), w = wave_id();
    uint32_t incl = x;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t t = __shfl_up(incl, d, 64);
        if (lane >= d) incl += t;
    }
    __syncthreads();
    if (lane == 63) lds[w] = incl;
    __syncthreads();
    if (w == 0) {
        uint32_t v = lane < ECNE_NWAVES ? lds[lane] : 0, inc = v;
#pragma unroll
        for (int d = 1; d < ECNE_NWAVES; d <<= 1) {
            uint32_t t = __shfl_up(inc, d, 64);
            if (lane >= d) inc += t;
        }
        if (lane < ECNE_NWAVES) lds[lane] = inc - v;
        if (lane == ECNE_NWAVES - 1) lds[ECNE_NWAVES] = inc;
    }
    __syncthreads();
    uint32_t off = lds[w] + incl - x;
    *total = lds[ECNE_NWAVES];
    __syncthreads();
    return off;
}

// Ordered multi-source REQUEUE: events[0..n) are variables in the reference's order; equivalent to
// calling requeue() for each in turn. Driven by one wavefront (the caller passes the queue cursor).
__device__ void requeue_events(const Job& J, QState& q, const uint32_t* events, uint32_t n) {
    for (uint32_t e = 0; e < n; ++e) requeue(J, q, events[e]);
}

// 128-bit commutative hash of a set of variable ids (P3 group key: the sorted unknown tuple, :1386-1387)
__device__ __forceinline__ uint64_t mixA(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}
__device__ __forceinline__ uint64_t mixB(uint64_t x) {
    x += 0x9e3779b97f4a7c15ULL; x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ULL; x = (x ^ (x >> 27)) * 0x94d049bb133111ebULL;
    return x ^ (x >> 31);
}

// P3 eligibility of one row (one lane per row): no non-unique variable in A or B; k = number of
// non-unique variables of C; h/h2 = commutative hash of that set (:1360-1386).
__device__ __noinline__ void p3_eval(const Job& J, uint32_t row, uint32_t& k, uint64_t& h, uint64_t& h2) {
    k = 0; h = 0; h2 = 0;
    // entries four per part at a time: ids first, then flag bytes (the constant wire pads short parts)
    const uint32_t a0 = J.rpA[row], a1 = J.rpA[row + 1], b0 = J.rpB[row], b1 = J.rpB[row + 1];
    const uint32_t c0 = J.rpC[row], c1 = J.rpC[row + 1];
    uint32_t n = a1 - a0;
    n = b1 - b0 > n ? b1 - b0 : n;
    n = c1 - c0 > n ? c1 - c0 : n;
    for (uint32_t off = 0; off < n; off += 4) {
        uint32_t v[12];
        uint8_t fl[12];
#pragma unroll
        for (uint32_t i = 0; i < 4; ++i) {
            v[i] = a0 + off + i < a1 ? J.colA[a0 + off + i] : 1u;
            v[4 + i] = b0 + off + i < b1 ? J.colB[b0 + off + i] : 1u;
            v[8 + i] = c0 + off + i < c1 ? J.colC[c0 + off + i] : 1u;
        }
#pragma unroll
        for (uint32_t i = 0; i < 12; ++i) fl[i] = J.flags[v[i]];
        // (the padding counts as unique whatever the constant wire's state is: a caller's known_variables need not hold it)
#pragma unroll
        for (uint32_t i = 0; i < 4; ++i) {
            if (a0 + off + i >= a1) fl[i] = 3;
            if (b0 + off + i >= b1) fl[4 + i] = 3;
            if (c0 + off + i >= c1) fl[8 + i] = 3;
        }
        uint32_t ab = 1;
#pragma unroll
        for (uint32_t i = 0; i < 8; ++i) ab &= fl[i];
        if (!(ab & 1)) { k = 0xFFFFFFFFu; return; }
#pragma unroll
        for (uint32_t i = 8; i < 12; ++i)
            if (!(fl[i] & 1)) { ++k; h += mixA(v[i]); h2 += mixB(v[i]); }
    }
    h = mixA(h + k);   // never 0-sensitive: empty sets are not inserted
}
// Four consecutive rows at once (one lane): the row pointers of all four, then their entries, then the flag bytes -- three dependent
// trips for four rows instead of three per row (a sweep's lane walks its rows one after the other otherwise: P3 was 0.35 ms per
// pass on a 24 000-row single-workgroup job, 25 us per pass on the 170 workgroups of ecdsa_like(26)). want: bit r = evaluate row
// r4 + r. Rows with more than four entries in a part go through p3_eval().
__device__ __noinline__ void p3_eval4(const Job& J, uint32_t r4, uint32_t want, uint32_t* k, uint64_t* h, uint64_t* h2) {
    uint32_t p0[4][3], n_[4][3];
    uint32_t wide = 0;
#pragma unroll
    for (uint32_t r = 0; r < 4; ++r) {
        const bool on = (want >> r) & 1u;
        const uint32_t row = on ? r4 + r : r4;
        const uint32_t a0 = J.rpA[row], a1 = J.rpA[row + 1], b0 = J.rpB[row], b1 = J.rpB[row + 1], c0 = J.rpC[row], c1 = J.rpC[row + 1];
        p0[r][0] = a0; p0[r][1] = b0; p0[r][2] = c0;
        n_[r][0] = on ? a1 - a0 : 0u; n_[r][1] = on ? b1 - b0 : 0u; n_[r][2] = on ? c1 - c0 : 0u;
        if (n_[r][0] > 4u || n_[r][1] > 4u || n_[r][2] > 4u) wide |= 1u << r;
    }
    uint32_t v[4][12];
#pragma unroll
    for (uint32_t r = 0; r < 4; ++r) {
        const bool on = ((want & ~wide) >> r) & 1u;
#pragma unroll
        for (uint32_t i = 0; i < 4; ++i) {
            v[r][i] = (on && i < n_[r][0]) ? J.colA[p0[r][0] + i] : 1u;
            v[r][4 + i] = (on && i < n_[r][1]) ? J.colB[p0[r][1] + i] : 1u;
            v[r][8 + i] = (on && i < n_[r][2]) ? J.colC[p0[r][2] + i] : 1u;
        }
    }
    uint8_t fl[4][12];
#pragma unroll
    for (uint32_t r = 0; r < 4; ++r)
#pragma unroll
        for (uint32_t i = 0; i < 12; ++i) { const uint8_t f = J.flags[v[r][i]]; fl[r][i] = (i & 3u) < n_[r][i >> 2] ? f : (uint8_t)3; }
#pragma unroll
    for (uint32_t r = 0; r < 4; ++r) {
        k[r] = 0; h[r] = 0; h2[r] = 0;
        if (!((want >> r) & 1u)) continue;
        if ((wide >> r) & 1u) { p3_eval(J, r4 + r, k[r], h[r], h2[r]); continue; }
        uint32_t ab = 1;
#pragma unroll
        for (uint32_t i = 0; i < 8; ++i) ab &= fl[r][i];
        if (!(ab & 1)) { k[r] = 0xFFFFFFFFu; continue; }
#pragma unroll
        for (uint32_t i = 8; i < 12; ++i)
            if (!(fl[r][i] & 1)) { ++k[r]; h[r] += mixA(v[r][i]); h2[r] += mixB(v[r][i]); }
        h[r] = mixA(h[r] + k[r]);
    }
}
// Open-addressing table keyed by the 64-bit half of the group hash; the other half is recorded with
// a second CAS by every visitor, so two different keys that agree on 64 bits are DETECTED (the solve
// stops with ECNE_ECAPACITY) instead of being merged. No lane ever spins on another lane.
__device__ __forceinline__ uint32_t ht_slot(const Job& J, uint64_t h, uint64_t h2, bool insert, bool* created = nullptr) {
    const unsigned long long key = (unsigned long long)(h | 1ull), key2 = (unsigned long long)(h2 | 1ull);
    uint32_t s = (uint32_t)((h >> 1) & J.htmask);
    for (uint32_t probe = 0; probe <= J.htmask; ++probe) {
        unsigned long long cur = __hip_atomic_load((unsigned long long*)&J.ht_key[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur == 0ull) {
            if (!insert) return 0xFFFFFFFFu;
            cur = atomicCAS((unsigned long long*)&J.ht_key[s], 0ull, key);
            if (cur == 0ull) { cur = key; if (created) *created = true; }
        }
        if (cur == key) {
            unsigned long long o2 = atomicCAS((unsigned long long*)&J.ht_key2[s], 0ull, key2);
            if (o2 != 0ull && o2 != key2) { raise(J, K_ECAPACITY); return 0xFFFFFFFFu; }
            return s;
        }
        s = (s + 1) & J.htmask;
    }
    raise(J, K_ECAPACITY);
    return 0xFFFFFFFFu;
}

// coefficient of variable v in row's C part (0 if absent)
__device__ fp::u256 c_coef(const Job& J, uint32_t row, uint32_t v) {
    for (uint32_t e = J.rpC[row]; e < J.rpC[row + 1]; ++e)
        if (J.colC[e] == v) return ld256(J.coefC + 4ull * e);
    return fp::make(0);
}

// slow_det (:1389-1400): sum over ODD permutations only (Combinatorics.parity is 0 for even and
// 1 for odd permutations and is used as a factor). Wave-parallel over permutation indices.
// rows[0..k) in arrival order, vars[0..k) ascending. Returns non-zero?
__device__ __noinline__ bool p3_odd_perm_sum_nonzero(const Job& J, const uint32_t* rows, const uint32_t* vars, uint32_t k) {
    const int lane = lane_id();
    uint64_t nperm = 1;
    for (uint32_t i = 2; i <= k; ++i) nperm *= i;
    fp::u256 acc = fp::make(0);
    for (uint64_t pi = lane; pi < nperm; pi += 64) {
        // decode permutation number pi (factoradic) into perm[], count inversions
        uint32_t perm[10], avail[10];
        for (uint32_t i = 0; i < k; ++i) avail[i] = i;
        uint64_t rem = pi;
        uint64_t f = nperm;
        uint32_t inv = 0;
        for (uint32_t i = 0; i < k; ++i) {
            f /= (k - i);
            uint32_t d = (uint32_t)(rem / f);
            rem -= (uint64_t)d * f;
            perm[i] = avail[d];
            inv += d;
            for (uint32_t j = d; j + 1 < k - i; ++j) avail[j] = avail[j + 1];
        }
        if (inv & 1) {
            fp::u256 term = fp::make(1);
            for (uint32_t j = 0; j < k; ++j) term = fp::mul(term, c_coef(J, rows[j], vars[perm[j]]));
            acc = fp::add(acc, term);
        }
    }
    // wave reduction (field addition)
    for (int d = 32; d >= 1; d >>= 1) {
        fp::u256 o = shfl256(acc, (lane + d) & 63);
        if (lane < d) acc = fp::add(acc, o);
    }
    acc = shfl256(acc, 0);
    return !fp::is_zero(acc);
}

// ---------------------------------------------------------------------------------------- k_solve
// workgroup-uniform view of the device error word (every thread takes the same branch)
__device__ __forceinline__ int wg_error(const Job& J, int* s_err) {
    __syncthreads();
    if (threadIdx.x == 0) *s_err = atomicAdd(&J.ctr->error, 0);
    __syncthreads();
    return *s_err;
}

// ---- wavefront-level helpers
__device__ __forceinline__ void lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ uint32_t wave_min(uint32_t x) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { const uint32_t y = __shfl_xor(x, d, 64); x = y < x ? y : x; }
    return x;
}

}  // namespace ecne
