// frontend.hip.hpp — gfx950 kernels of the device front-end (ecne_frontend.hip): .r1cs parse, abstraction's exact
// verification + compaction, and the flat-array layout of a constraint system, all on the GPU.
//
// 1. PARSE (readR1CS, /root/reference/src/ParseR1CS.jl:50-124). The constraint section is a chain of length-prefixed
//    parts (u32 n, then n x (u32 wire, 32-byte coefficient) = 1 + 9 n words): where part k starts is only known once parts
//    0..k-1 have been walked. The walk is made parallel by walking from EVERY word ("candidate") at once:
//      k_fe_exit1   a workgroup stages a 1 KB chunk in LDS; lane j walks from candidate j until it leaves the chunk and
//                   records (exit word, parts walked) -- ~24 dependent LDS reads per lane
//      k_fe_exit2   the same for 256 KB tiles by dynamic programming over the chunks of a tile, last chunk first: one
//                   table lookup per candidate and step (exit2[j] = exit2[exit1[j]] while that stays inside the tile)
//      k_fe_chain   ONE lane follows exit2 from the first constraint: a dependent load per tile (142 MB: 555 loads)
//      k_fe_chunk_entries / k_fe_part_offsets   one lane per tile follows exit1 to the chunks the chain enters, then one lane
//                   per chunk walks its parts and writes the offset of every part, checking the reader's bounds
//    Then the rows are built in the reference's DICTIONARY order: k_fe_terms (term counts) -> scans -> k_fe_fill_small
//    (one lane per part of up to 10 terms: a 16-slot Julia hash table per lane in LDS) / k_fe_fill_big (one wavefront per
//    longer part: lane 0 replays the insertions -- growth steps included -- then all lanes emit the slots), coefficients
//    reduced mod p on the way ("last value wins at the first occurrence's position", :104-111).
// 2. ABSTRACTION (:237-395): k_abs_sig / k_abs_match / k_abs_exact verify a candidate window by building the variable
//    bijection from commutative signature hashes and then PROVING it entry by entry (an isomorphism of the two windows is
//    exactly what the reference's sorted-signature comparison :334-351 decides); k_fe_compact_* copy the surviving rows.
// 3. LAYOUT: nonzeroKeys order per row part (a Julia Set filled in dictionary order, :26-34) with the same per-lane /
//    per-wavefront hash tables, the structural row descriptors (which rule a row can ever feed and with which variables),
//    value slots, the P4 / P5 / long-row lists, variable_to_indices (:628-633) by a radix sort of (variable, row) pairs,
//    the row records and inline fan-out lists of the chain executor.
// No MFMA anywhere: byte / index work bound by HBM and by dependent-load latency.
#pragma once
#include "dev_common.hip.hpp"
#include "frontend.hpp"
#include "jlslot.hpp"

namespace ecne {
namespace fe {

#define FE_CH 256u                      // words per level-1 chunk (1 KB)
#define FE_TILE_CH 256u                 // chunks per tile (256 KB)
#define FE_TILE_W (FE_CH * FE_TILE_CH)
#define FE_NONE 0xFFFFFFFFFFFFFFFFull
#define FE_LANE_MAX 10u                 // parts of up to this many terms: one lane each (a 16-slot table never grows with <= 10 keys)
#define FE_MID_MAX 170u                 // up to this many: one wavefront, tables of <= 1024 slots in LDS (256 by count; a probe sequence of 16 -- which
                                        // Julia's hash of consecutive ids does produce -- quadruples the table early); beyond, or when
                                        // even that overflows: tables in HBM scratch
#define FE_MID_CAP 1024u
#define FE_MID_WAVES 2u                 // wavefronts per workgroup in the LDS tier (2 x 4 arrays x 1024 slots x 4 B = 32 KB)
#define FE_LARGE_MAX 2730u              // up to this many: one wavefront per workgroup, tables of <= 4096 slots in 64 KB of (dynamic) LDS
#define FE_LARGE_CAP 4096u
// tiers of the hash-order kernels: 1 = LDS 1024 slots x 2 wavefronts, 2 = LDS 4096 slots x 1 wavefront, 0 = HBM scratch x 4 wavefronts
// table storage of the hash-order kernels: LDS-address-space pointers for the LDS tiers (ds_read / ds_write in jlslot's walk; through
// a generic pointer an LDS access costs what an L2 hit costs -- lane 0 replays up to 2 730 insertions, one dependent probe after the other)
typedef __attribute__((address_space(3))) uint32_t* FeLdsPtr;
template <bool LDS> struct FeTabPtr { typedef uint32_t* type; };
template <> struct FeTabPtr<true> { typedef FeLdsPtr type; };
template <int TIER> struct FeTier {
    static constexpr uint32_t cap = TIER == 1 ? FE_MID_CAP : TIER == 2 ? FE_LARGE_CAP : 0u;
    static constexpr uint32_t waves = TIER == 1 ? FE_MID_WAVES : TIER == 2 ? 1u : 4u;
    // (round 6) per wavefront: the four table arrays, then the staged insertions (key, hash, payload) and the list a growth re-inserts from
    // (key, payload + 1, hash) -- fe_wave_table; `stage` entries each (the most keys a part of the tier holds, rounded up)
    static constexpr uint32_t stage = TIER == 1 ? 192u : TIER == 2 ? 2752u : 0u;
    static constexpr uint32_t words = 4u * cap + 6u * stage;
    static constexpr uint32_t lds_bytes = waves * words * 4u;
};
static_assert(FeTier<1>::stage >= FE_MID_MAX + 1 && FeTier<2>::stage >= FE_LARGE_MAX + 1, "staging holds every key of a part of the tier");

__device__ __forceinline__ uint64_t ld64_agent(const uint64_t* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint32_t ld32_agent(const uint32_t* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ====================================================================================== scan (exclusive, u32)
// 1024 elements per workgroup; in may alias out
__global__ __launch_bounds__(256) void k_scan_blocks(const uint32_t* in, uint32_t n, uint32_t* out, uint32_t* tops) {
    __shared__ uint32_t s_w[4];
    const uint32_t base = blockIdx.x * 1024u + threadIdx.x * 4u;
    uint32_t v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = base + i < n ? in[base + i] : 0u;
    const uint32_t mine = v[0] + v[1] + v[2] + v[3];
    uint32_t incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t t = __shfl_up(incl, d, 64); if ((threadIdx.x & 63) >= (unsigned)d) incl += t; }
    if ((threadIdx.x & 63) == 63) s_w[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t off = 0;
    for (unsigned k = 0; k < (threadIdx.x >> 6); ++k) off += s_w[k];
    uint32_t run = off + incl - mine;
#pragma unroll
    for (int i = 0; i < 4; ++i) { if (base + i < n) out[base + i] = run; run += v[i]; }
    if (threadIdx.x == 255) tops[blockIdx.x] = run;
}
__global__ __launch_bounds__(256) void k_scan_tops(uint32_t* tops, uint32_t nb, uint32_t* total) {
    __shared__ uint32_t s_carry, s_w[4];
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (uint32_t b0 = 0; b0 < nb; b0 += 256) {
        const uint32_t i = b0 + threadIdx.x;
        const uint32_t x = i < nb ? tops[i] : 0u;
        uint32_t incl = x;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t t = __shfl_up(incl, d, 64); if ((threadIdx.x & 63) >= (unsigned)d) incl += t; }
        if ((threadIdx.x & 63) == 63) s_w[threadIdx.x >> 6] = incl;
        __syncthreads();
        uint32_t off = s_carry;
        for (unsigned k = 0; k < (threadIdx.x >> 6); ++k) off += s_w[k];
        if (i < nb) tops[i] = off + incl - x;
        __syncthreads();
        if (threadIdx.x == 255) s_carry = off + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0 && total) *total = s_carry;
}
__global__ __launch_bounds__(256) void k_scan_add(uint32_t* out, uint32_t n, const uint32_t* tops) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) out[i] += tops[i >> 10];
}

// ====================================================================================== 1. parse: where the parts start
// packed (parts walked << 32 | exit word)
__global__ __launch_bounds__(256) void k_fe_exit1(const uint32_t* __restrict__ W, uint32_t NW, uint64_t* __restrict__ E1) {
    __shared__ uint32_t s[FE_CH];
    const uint32_t nchunks = (NW + FE_CH - 1) / FE_CH;
    for (uint32_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
        const uint32_t base = chunk * FE_CH, j = base + threadIdx.x;
        s[threadIdx.x] = j < NW ? W[j] : 0u;
        __syncthreads();
        uint32_t cur = threadIdx.x, cnt = 0;
        uint64_t nx;
        for (;;) {
            nx = (uint64_t)cur + 1ull + 9ull * (uint64_t)s[cur];
            ++cnt;
            if (nx >= FE_CH) break;
            cur = (uint32_t)nx;
        }
        uint64_t e = (uint64_t)base + nx;
        if (e > NW) e = NW;
        if (j < NW) E1[j] = ((uint64_t)cnt << 32) | e;
        __syncthreads();
    }
}
// one workgroup per tile, chunks last to first; E2 entries of later chunks were written by this workgroup (read past the L1)
__global__ __launch_bounds__(256) void k_fe_exit2(const uint64_t* __restrict__ E1, uint64_t* E2, uint32_t NW) {
    const uint32_t ntiles = (NW + FE_TILE_W - 1) / FE_TILE_W;
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint64_t tile_base = (uint64_t)tile * FE_TILE_W, tile_end = tile_base + FE_TILE_W;
        uint32_t nch = FE_TILE_CH;
        if (tile_end > NW) nch = (uint32_t)((NW - tile_base + FE_CH - 1) / FE_CH);
        for (int c = (int)nch - 1; c >= 0; --c) {
            const uint64_t j = tile_base + (uint64_t)c * FE_CH + threadIdx.x;
            if (j < NW) {
                uint64_t e = E1[j];
                const uint32_t x = (uint32_t)e;
                if (x < tile_end && x < NW) {
                    const uint64_t f = ld64_agent(&E2[x]);
                    e = ((e >> 32) + (f >> 32)) << 32 | (f & 0xFFFFFFFFull);
                }
                E2[j] = e;
            }
            __syncthreads();
        }
    }
}
// tile_entry[t] = (part index << 32 | word) where the chain enters tile t, FE_NONE for tiles it jumps over
__global__ void k_fe_chain(const uint64_t* __restrict__ E2, uint32_t NW, uint32_t total, uint64_t* __restrict__ tile_entry,
                           uint64_t* __restrict__ final_state) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    uint32_t j = 0, idx = 0;
    while (j < NW && idx < total) {
        tile_entry[j / FE_TILE_W] = ((uint64_t)idx << 32) | j;
        const uint64_t e = E2[j];
        j = (uint32_t)e;
        idx += (uint32_t)(e >> 32);
    }
    final_state[0] = j;
    final_state[1] = idx;
}
__global__ __launch_bounds__(256) void k_fe_chunk_entries(const uint64_t* __restrict__ E1, const uint64_t* __restrict__ tile_entry,
                                                          uint64_t* __restrict__ chunk_entry, uint32_t NW, uint32_t total) {
    const uint32_t ntiles = (NW + FE_TILE_W - 1) / FE_TILE_W;
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= ntiles) return;
    const uint64_t e0 = tile_entry[t];
    if (e0 == FE_NONE) return;
    uint32_t j = (uint32_t)e0, idx = (uint32_t)(e0 >> 32);
    const uint64_t tile_end = ((uint64_t)t + 1) * FE_TILE_W;
    while (j < tile_end && j < NW && idx < total) {
        chunk_entry[j / FE_CH] = ((uint64_t)idx << 32) | j;
        const uint64_t e = E1[j];
        j = (uint32_t)e;
        idx += (uint32_t)(e >> 32);
    }
}
// poff[k] = word of part k's header; err = lowest part index that violates the reader's bounds (ParseR1CS.jl reads past the
// end: here K_EFORMAT), all ones = none
__global__ __launch_bounds__(256) void k_fe_part_offsets(const uint32_t* __restrict__ W, const uint64_t* __restrict__ chunk_entry,
                                                         uint32_t* __restrict__ poff, uint32_t NW, uint64_t len_bytes, uint32_t total,
                                                         uint32_t* err) {
    const uint32_t nchunks = (NW + FE_CH - 1) / FE_CH;
    const uint32_t c = blockIdx.x * 256u + threadIdx.x;
    if (c >= nchunks) return;
    const uint64_t e0 = chunk_entry[c];
    if (e0 == FE_NONE) return;
    uint64_t j = (uint32_t)e0;
    uint32_t idx = (uint32_t)(e0 >> 32);
    const uint64_t cend = ((uint64_t)c + 1) * FE_CH;
    while (j < cend && idx < total) {
        if (j >= NW || 4ull * j + 4ull > len_bytes) { atomicMin(err, idx); return; }
        const uint64_t n = W[j];
        if (4ull * j + 4ull + 36ull * n > len_bytes) { atomicMin(err, idx); return; }
        poff[idx] = (uint32_t)j;
        j += 1ull + 9ull * n;
        ++idx;
    }
}

// ====================================================================================== 1. parse: the rows in dictionary order
struct FeMeta {      // device-side results of the parse / layout passes (zeroed before use)
    uint32_t n_mid, n_large, n_huge, maxn, dup, unsupported, maxvar, err_idx, maxlenC, dsu_err;
    unsigned long long nnz[3];
};
struct FeRowsOut { uint64_t* ptr[3]; uint32_t* var[3]; uint64_t* coef[3]; };

__device__ __forceinline__ fp::u256 fe_ld_coef(const uint32_t* __restrict__ W, uint64_t w) {   // 8 words from W[w] on
    fp::u256 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v.w[i] = (uint64_t)W[w + 2 * i] | ((uint64_t)W[w + 2 * i + 1] << 32);
    return fp::reduce(v);
}

// ---- the Julia hash table of a long part, built by the whole wavefront (round 6). The slot a key ends up in depends on the insertions before it
// (linear probing, growth re-inserts in slot order), so the insertions stay one after the other on lane 0 -- but everything around them does not:
// the keys and their hashes are staged in LDS by all lanes (lane 0's loop was one dependent global load and one 64-bit hash chain per key), a
// growth zeroes the new table, collects the old slots in order and hashes their keys with all lanes, and lane 0 only walks the probes.
// 1 025 keys: 0.95 -> 0.58 ms (k_fe_fill_big<2>) / 0.83 -> 0.60 ms (k_lay_order_big<2>) for the 208 long parts of ecdsa_like(26), one wavefront each;
// the table build is 95 % of that (cycle counters: staging 20 k, table 1.25 M, emission 34 k cycles): two dependent LDS round trips per insertion.
// t: the table (LDS pointers; the same value in every lane on entry and on return). st_key / st_hash / st_pay: n staged insertions in order
// (payload = what tab_upsert stores + 1 later). gl_*: scratch for a growth (>= the keys the table holds). Returns nonzero when the table would
// outgrow t.cap (the caller hands the part to the next tier).
typedef __attribute__((address_space(3))) uint64_t* FeLdsPtr64;
// (the table itself: ONE array of 64-bit slots per buffer, payload + 1 in the high word, key in the low one -- jlslot::Tab64T -- so that a probe
//  is one ds_read_b64 and not two dependent reads; the two buffers are the 4 * cap words the separate arrays took)
template <int ADD>
__device__ __forceinline__ uint32_t fe_wave_table(jlslot::Tab64T<FeLdsPtr64>& t, uint32_t n, FeLdsPtr st_key, FeLdsPtr st_hash, FeLdsPtr st_pay,
                                                  FeLdsPtr gl_key, FeLdsPtr gl_pay, FeLdsPtr gl_hash) {
    const int lane = lane_id();
    n = (uint32_t)__builtin_amdgcn_readfirstlane((int)n);      // (the same in every lane: say so, the walks below are meant for the scalar unit)
    for (uint32_t i = lane; i < 16; i += 64) t.cur[i] = 0ull;
    t.sz = 16; t.n = 0; t.maxprobe = 0;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    uint32_t k = 0;
    while (k < n) {
        // insertions from k on, until the table has to grow: EVERY lane walks, on the same values (jlslot.hpp, UNI: the walk runs on the scalar unit)
        uint32_t rc = 0, want = 0;
        {
            uint64_t w64 = 0;
            while (k < n) {
                const uint32_t ky = (uint32_t)__builtin_amdgcn_readfirstlane((int)st_key[k]), py = (uint32_t)__builtin_amdgcn_readfirstlane((int)st_pay[k]),
                               hh = (uint32_t)__builtin_amdgcn_readfirstlane((int)st_hash[k]);
                rc = (uint32_t)jlslot::tab_try_upsert64<true>(t, ky, py, hh, &w64);
                if (rc != 1) ++k;
                if (rc) break;
            }
            want = (uint32_t)(w64 > 0x80000000ull ? 0x80000000ull : w64);
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        if (!rc) break;
        // ---- growth (tab_grow, by the wavefront): new size, zero the other buffer, the old slots in ascending order with their hashes
        uint32_t nsz = 16;
        while (nsz < want) nsz <<= 1;
        if (nsz > t.cap || want > t.cap) return 1u;
        for (uint32_t i = lane; i < nsz; i += 64) t.nxt[i] = 0ull;
        uint32_t cnt = 0;
        for (uint32_t s0 = 0; s0 < t.sz; s0 += 64) {
            const uint32_t sl = s0 + lane;
            const uint64_t sv = sl < t.sz ? t.cur[sl] : 0ull;
            const uint32_t py = (uint32_t)(sv >> 32);
            const uint64_t m = __ballot(py != 0);
            if (py) {
                const uint32_t o = cnt + (uint32_t)__popcll(m & lanes_below());
                const uint32_t ky = (uint32_t)sv;
                gl_key[o] = ky; gl_pay[o] = py; gl_hash[o] = (uint32_t)jlslot::hash64((uint64_t)ky + (uint64_t)ADD);
            }
            cnt += (uint32_t)__popcll(m);
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        uint32_t mp = 0;
        {
            const uint32_t mask = nsz - 1;
            for (uint32_t i = 0; i < cnt; ++i) {      // (every lane, uniformly)
                const uint32_t home = (uint32_t)__builtin_amdgcn_readfirstlane((int)gl_hash[i]) & mask;
                const uint32_t idx = jlslot::tab_free_slot64<true>(t.nxt, mask, home);
                const uint32_t probe = (idx - home) & mask;
                if (probe > mp) mp = probe;
                t.nxt[idx] = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)gl_pay[i]) << 32) | (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)gl_key[i]);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        const FeLdsPtr64 a = t.cur; t.cur = t.nxt; t.nxt = a;
        t.sz = nsz;
        t.maxprobe = mp;
    }
    return 0u;
}

// cnt[p * (nC + 1) + r] = entries part p of row r will hold at most (an empty part becomes {1 => 0}, :113-115)
__global__ __launch_bounds__(256) void k_fe_terms(const uint32_t* __restrict__ W, const uint32_t* __restrict__ poff, uint32_t total, uint32_t nC,
                                                  uint32_t* __restrict__ cnt, uint32_t* __restrict__ midlist, uint32_t* __restrict__ largelist,
                                                  uint32_t* __restrict__ hugelist, FeMeta* M) {
    __shared__ uint32_t s_max;
    if (threadIdx.x == 0) s_max = 0;
    __syncthreads();
    uint32_t mx = 0;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        const uint32_t n = W[poff[i]];
        const uint32_t r = i / 3u, p = i - 3u * r;
        cnt[(size_t)p * (nC + 1) + r] = n ? n : 1u;
        if (n > FE_LARGE_MAX) hugelist[atomicAdd(&M->n_huge, 1u)] = i;
        else if (n > FE_MID_MAX) largelist[atomicAdd(&M->n_large, 1u)] = i;
        else if (n > FE_LANE_MAX) midlist[atomicAdd(&M->n_mid, 1u)] = i;
        mx = n > mx ? n : mx;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { const uint32_t y = __shfl_xor(mx, d, 64); mx = y > mx ? y : mx; }
    if ((threadIdx.x & 63) == 0 && mx) atomicMax(&s_max, mx);
    __syncthreads();
    if (threadIdx.x == 0 && s_max) atomicMax(&M->maxn, s_max);      // (one device atomic per workgroup: thousands on one word cost milliseconds)
}

__global__ __launch_bounds__(256) void k_fe_fill_small(const uint32_t* __restrict__ W, const uint32_t* __restrict__ poff, uint32_t total, uint32_t nC,
                                                       const uint32_t* __restrict__ pos, FeRowsOut O, uint32_t* __restrict__ len, FeMeta* M) {
    __shared__ uint32_t s_key[16 * 256], s_pay[16 * 256];
    __shared__ uint32_t s_nz[3], s_maxvar, s_dup;
    if (threadIdx.x < 3) s_nz[threadIdx.x] = 0;
    if (threadIdx.x == 3) { s_maxvar = 0; s_dup = 0; }
    __syncthreads();
    uint32_t nz3[3] = {0, 0, 0}, mvall = 0;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        const uint64_t j = poff[i];
        const uint32_t n = W[j];
        const uint32_t r = i / 3u, p = i - 3u * r;
        if (n > FE_LANE_MAX) continue;
        const uint32_t at = pos[(size_t)p * (nC + 1) + r];
        uint32_t m = 0, nz = 0, mv = 0;
        if (n == 0) {
            O.var[p][at] = 1u;
            st256(O.coef[p] + 4ull * at, fp::make(0));
            m = 1; mv = 1;
        } else if (n == 1) {
            const fp::u256 c = fe_ld_coef(W, j + 2);
            const uint32_t v = W[j + 1] + 1u;
            O.var[p][at] = v;
            st256(O.coef[p] + 4ull * at, c);
            m = 1; mv = v; nz = !fp::is_zero(c);
        } else {
            jlslot::TabT<FeLdsPtr> t;
            t.key = (FeLdsPtr)(s_key + threadIdx.x); t.pay = (FeLdsPtr)(s_pay + threadIdx.x); t.key2 = nullptr; t.pay2 = nullptr;
            t.cap = 16; t.stride = 256;
            jlslot::tab_init(t);
            for (uint32_t k = 0; k < n; ++k) (void)jlslot::tab_upsert<1>(t, W[j + 1 + 9ull * k], k);
            for (uint32_t s = 0; s < 16; ++s) {
                const uint32_t py = t.pay[s * 256];
                if (!py) continue;
                const uint32_t k = py - 1u, v = t.key[s * 256] + 1u;
                const fp::u256 c = fe_ld_coef(W, j + 2 + 9ull * k);
                O.var[p][at + m] = v;
                st256(O.coef[p] + 4ull * (at + m), c);
                ++m;
                mv = v > mv ? v : mv;
                nz += !fp::is_zero(c);
            }
            if (m < n) s_dup = 1;
        }
        len[(size_t)p * (nC + 1) + r] = m;
        if (p == 0) nz3[0] += nz; else if (p == 1) nz3[1] += nz; else nz3[2] += nz;
        mvall = mv > mvall ? mv : mvall;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        nz3[0] += __shfl_xor(nz3[0], d, 64); nz3[1] += __shfl_xor(nz3[1], d, 64); nz3[2] += __shfl_xor(nz3[2], d, 64);
        const uint32_t y = __shfl_xor(mvall, d, 64); mvall = y > mvall ? y : mvall;
    }
    if ((threadIdx.x & 63) == 0) {
        for (int p = 0; p < 3; ++p) if (nz3[p]) atomicAdd(&s_nz[p], nz3[p]);
        atomicMax(&s_maxvar, mvall);
    }
    __syncthreads();
    if (threadIdx.x < 3 && s_nz[threadIdx.x]) atomicAdd(&M->nnz[threadIdx.x], (unsigned long long)s_nz[threadIdx.x]);
    if (threadIdx.x == 3) { if (s_maxvar) atomicMax(&M->maxvar, s_maxvar); if (s_dup) atomicOr(&M->dup, 1u); }
}

// One wavefront per listed part; the tables live in (dynamic) LDS (TIER 1, 2) or in the wavefront's HBM scratch (TIER 0, gcap
// slots per buffer). Lane 0 replays the dictionary insertions; all lanes emit the slots. A part whose table outgrows its tier
// goes to the next one's list (overflow_list / overflow_count), from the HBM tier to the host path.
template <int TIER>
__global__ __launch_bounds__(64 * FeTier<TIER>::waves) void k_fe_fill_big(const uint32_t* __restrict__ W, const uint32_t* __restrict__ poff, const uint32_t* __restrict__ list,
                                                     uint32_t nlist, uint32_t nC, const uint32_t* __restrict__ pos, FeRowsOut O,
                                                     uint32_t* __restrict__ len, FeMeta* M, uint32_t* gscratch, uint32_t gcap, uint32_t* __restrict__ overflow_list,
                                                     uint32_t* overflow_count) {
    constexpr bool LDS_TABLES = TIER != 0;
    constexpr uint32_t WAVES = FeTier<TIER>::waves;
    uint32_t* const s_tab = reinterpret_cast<uint32_t*>(ecne_dyn_lds);
    const int lane = lane_id();
    const uint32_t wave = threadIdx.x >> 6, gw = blockIdx.x * WAVES + wave, nw = gridDim.x * WAVES;
    uint32_t* base = LDS_TABLES ? s_tab + (size_t)wave * FeTier<TIER>::words : gscratch + (size_t)gw * 4 * gcap;
    const uint32_t cap = LDS_TABLES ? FeTier<TIER>::cap : gcap;
    for (uint32_t b = gw; b < nlist; b += nw) {
        const uint32_t i = list[b];
        const uint64_t j = poff[i];
        const uint32_t n = W[j];
        const uint32_t r = i / 3u, p = i - 3u * r;
        const uint32_t at = pos[(size_t)p * (nC + 1) + r];
        uint32_t sz = 0, flipped = 0, bad = 0;
        if constexpr (LDS_TABLES) {
            // (round 6) keys and hashes staged by all lanes, the table built by fe_wave_table
            const FeLdsPtr tb = (FeLdsPtr)base;
            const FeLdsPtr st = tb + 4 * FeTier<TIER>::cap;
            constexpr uint32_t SG = FeTier<TIER>::stage;
            for (uint32_t k = lane; k < n; k += 64) {
                const uint32_t ky = W[j + 1 + 9ull * k];
                st[k] = ky; st[SG + k] = (uint32_t)jlslot::hash64((uint64_t)ky + 1ull); st[2 * SG + k] = k;
            }
            jlslot::Tab64T<FeLdsPtr64> t;
            t.cur = (FeLdsPtr64)tb; t.nxt = (FeLdsPtr64)tb + cap; t.cap = cap;
            bad = n > SG ? 1u : fe_wave_table<1>(t, n, st, st + SG, st + 2 * SG, st + 3 * SG, st + 4 * SG, st + 5 * SG);
            sz = t.sz;
            flipped = t.cur != (FeLdsPtr64)tb;
        } else {
        if (lane == 0) {
            typedef typename FeTabPtr<LDS_TABLES>::type TP;
            const TP tb = (TP)base;
            jlslot::TabT<TP> t;
            t.key = tb; t.pay = tb + cap; t.key2 = tb + 2 * cap; t.pay2 = tb + 3 * cap;
            t.cap = cap; t.stride = 1;
            jlslot::tab_init(t);
            for (uint32_t k = 0; k < n && !bad; ++k) bad = jlslot::tab_upsert<1>(t, W[j + 1 + 9ull * k], k) != 0;
            sz = t.sz;
            flipped = t.key != tb;
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        sz = __shfl(sz, 0, 64); flipped = __shfl(flipped, 0, 64); bad = __shfl(bad, 0, 64);
        }
        if (bad) {      // the table outgrew this tier: hand the part to the next one, or give the file back to the host path
            if (lane == 0) { if (LDS_TABLES) overflow_list[atomicAdd(overflow_count, 1u)] = i; else atomicOr(&M->unsupported, 1u); }
            continue;
        }
        // LDS tiers: the occupied slots (64-bit: key in the low word, payload + 1 in the high one) are first collected in slot order -- LDS work
        // only -- so that the emission below runs over full wavefronts of entries (a 1 025-entry part sits in 4 096 slots: 64 trips of
        // dependent global loads with a quarter of the lanes became 17 with all of them). HBM tier: slot by slot from the separate arrays.
        const uint32_t* key = flipped ? base + 2 * cap : base;
        const uint32_t* pay = flipped ? base + 3 * cap : base + cap;
        uint32_t n_em = sz;
        const uint32_t* const ckey = base + 4 * FeTier<TIER>::cap + 3 * FeTier<TIER>::stage;      // (the growth's list arrays are free now)
        const uint32_t* const cpay = ckey + FeTier<TIER>::stage;
        if constexpr (LDS_TABLES) {
            const FeLdsPtr64 slots = (FeLdsPtr64)(flipped ? base + 2 * cap : base);
            const FeLdsPtr lk = (FeLdsPtr)base + 4 * FeTier<TIER>::cap + 3 * FeTier<TIER>::stage, lp = lk + FeTier<TIER>::stage;
            uint32_t cnt = 0;
            for (uint32_t s0 = 0; s0 < sz; s0 += 64) {
                const uint32_t s = s0 + lane;
                const uint64_t sv = s < sz ? slots[s] : 0ull;
                const uint64_t mk = __ballot((uint32_t)(sv >> 32) != 0);
                if ((uint32_t)(sv >> 32)) { const uint32_t o = cnt + (uint32_t)__popcll(mk & lanes_below()); lk[o] = (uint32_t)sv; lp[o] = (uint32_t)(sv >> 32); }
                cnt += (uint32_t)__popcll(mk);
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            n_em = cnt;
        }
        uint32_t m = 0, nz = 0, mv = 0;
        for (uint32_t s0 = 0; s0 < n_em; s0 += 64) {
            const uint32_t s = s0 + lane;
            const uint32_t py = s < n_em ? (LDS_TABLES ? cpay[s] : ld32_agent(&pay[s])) : 0u;
            const uint64_t mask = __ballot(py != 0);
            if (py) {
                const uint32_t k = py - 1u, v = (LDS_TABLES ? ckey[s] : ld32_agent(&key[s])) + 1u;
                const uint32_t o = at + m + (uint32_t)__popcll(mask & lanes_below());
                const fp::u256 c = fe_ld_coef(W, j + 2 + 9ull * k);
                O.var[p][o] = v;
                st256(O.coef[p] + 4ull * o, c);
                mv = v > mv ? v : mv;
                nz += !fp::is_zero(c);
            }
            m += (uint32_t)__popcll(mask);
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            nz += __shfl_xor(nz, d, 64);
            const uint32_t y = __shfl_xor(mv, d, 64); mv = y > mv ? y : mv;
        }
        if (lane == 0) {
            len[(size_t)p * (nC + 1) + r] = m;
            if (nz) atomicAdd(&M->nnz[p], (unsigned long long)nz);
            atomicMax(&M->maxvar, mv);
            if (m < n) atomicOr(&M->dup, 1u);
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    }
}
// ptr[p][r] = pos (u32 -> u64), r = 0..nC (pos holds nC + 1 entries per part, the last one = the total)
__global__ __launch_bounds__(256) void k_fe_ptr(const uint32_t* __restrict__ pos, uint32_t nC, FeRowsOut O) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i > nC) return;
#pragma unroll
    for (int p = 0; p < 3; ++p) O.ptr[p][i] = pos[(size_t)p * (nC + 1) + i];
}
// parts that repeated a wire id came out shorter than their term count: close the gaps (old position pos, new position npos)
__global__ __launch_bounds__(256) void k_fe_close_gaps(uint32_t nC, const uint32_t* __restrict__ pos, const uint32_t* __restrict__ npos,
                                                       const uint32_t* __restrict__ len, FeRowsOut from, FeRowsOut to) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= 3u * nC) return;
    const uint32_t r = i / 3u, p = i - 3u * r;
    const size_t q = (size_t)p * (nC + 1) + r;
    const uint32_t a = pos[q], b = npos[q], m = len[q];
    for (uint32_t k = 0; k < m; ++k) {
        to.var[p][b + k] = from.var[p][a + k];
        st256(to.coef[p] + 4ull * (b + k), ld256(from.coef[p] + 4ull * (a + k)));
    }
}

// ====================================================================================== 2. abstraction: exact verification
struct AbsPattern {      // the trusted function, prepared on the host (ecne_frontend.hip: build_pattern)
    uint32_t nS, nvS, nEnt, nclass, capP, nio;
    const uint32_t* ent_cnt0;     // per non-zero entry: 3 * row + part
    const uint32_t* ent_var;      // pattern variable index 0..nvS-1
    const uint64_t* ent_coef;     // 4 limbs
    const uint32_t* part_nz;      // [3 * nS] non-zero entries per part
    const uint64_t *tab_h1, *tab_h2;   // signature-hash table (open addressing, capP slots)
    const uint32_t* tab_class;    // class id + 1, 0 = empty slot
    const uint32_t* class_start;  // [nclass + 1]
    const uint32_t* class_members;// pattern variable indices grouped by class
    const uint32_t* io_idx;       // pattern variable index of every mapped input / output (0xFFFFFFFF: absent)
};
struct AbsWindows {      // one batch of candidate windows
    uint32_t nwin, capW;
    const uint32_t* start;        // first row of each window
    unsigned long long* wkey;     // [nwin][capW] variable + 1, 0 = empty
    unsigned long long *wh1, *wh2;
    uint32_t* ccount;             // [nwin][nclass]
    uint32_t* phi;                // [nwin][nvS] window variable of every pattern variable
    uint32_t* nvars;              // [nwin] distinct variables inserted
    uint32_t* nmatched;           // [nwin]
    uint32_t* status;             // [nwin] bit0 = definitely no match, bit1 = the entry-by-entry proof failed
    uint32_t* io_out;             // [nwin][nio]
};
struct AbsRowsDev { const uint64_t* ptr[3]; const uint32_t* var[3]; const uint64_t* coef[3]; };

// lane per (window row, part): every non-zero entry adds its (counter, coefficient) hash to its variable's slot
__device__ __forceinline__ void abs_sig_add(AbsWindows& Wn, const AbsPattern& P, uint32_t w, unsigned long long key, uint64_t h1, uint64_t h2) {
    unsigned long long* wkey = Wn.wkey + (size_t)w * Wn.capW;
    const uint32_t mask = Wn.capW - 1;
    uint32_t s = (uint32_t)(sig_mix(key) & mask);
    for (uint32_t probe = 0;; ++probe) {
        if (probe > mask) { atomicOr(&Wn.status[w], 1u); return; }
        unsigned long long cur = __hip_atomic_load(&wkey[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur == 0ull) {      // (distinct variables are counted by k_abs_match, one atomic per wavefront: a counter bumped here by every
            cur = atomicCAS(&wkey[s], 0ull, key);      //  insertion serialises 16 000 atomics per window on one word)
            if (cur == 0ull) cur = key;
        }
        if (cur == key) {
            atomicAdd(&Wn.wh1[(size_t)w * Wn.capW + s], (unsigned long long)h1);
            atomicAdd(&Wn.wh2[(size_t)w * Wn.capW + s], (unsigned long long)h2);
            return;
        }
        s = (s + 1) & mask;
    }
}
__device__ __forceinline__ uint64_t wave_sum64(uint64_t x) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const uint32_t lo = __shfl_xor((uint32_t)x, d, 64), hi = __shfl_xor((uint32_t)(x >> 32), d, 64);
        x += ((uint64_t)hi << 32) | lo;
    }
    return x;
}
// lane per (window row, part): every non-zero entry adds its (counter, coefficient) hash to its variable's slot. The constant
// wire sits in most rows of a circuit: its appearances are summed per wavefront first (thousands of atomics on one slot serialise).
__global__ __launch_bounds__(256) void k_abs_sig(AbsRowsDev R, AbsPattern P, AbsWindows Wn) {
    const uint32_t w = blockIdx.y;
    const uint32_t q = blockIdx.x * 256u + threadIdx.x;
    uint64_t one1 = 0, one2 = 0;
    uint32_t one_n = 0;
    // (round 6) a long part -- a window of secp256k1 holds 87-term decompositions -- is walked by the whole wavefront (the sums are commutative),
    // not by its one lane while the other 63 wait
    const bool on = q < 3u * P.nS;
    const uint32_t jrow = on ? q / 3u : 0u, p = on ? q - 3u * jrow : 0u;
    const uint64_t row = (uint64_t)Wn.start[w] + jrow;
    const uint64_t k0 = on ? R.ptr[p][row] : 0ull, k1 = on ? R.ptr[p][row + 1] : 0ull;
    const bool longp = k1 - k0 > 32;
    uint32_t nz = 0;
    auto entry = [&](uint32_t pp, uint64_t k, uint64_t qq) {
        const uint64_t* c = R.coef[pp] + 4 * k;
        if ((c[0] | c[1] | c[2] | c[3]) == 0) return 0u;
        uint64_t h1, h2;
        sig_hash(qq + 1ull, c, h1, h2);
        const uint32_t v = R.var[pp][k];
        if (v == 1u) { one1 += h1; one2 += h2; ++one_n; }
        else abs_sig_add(Wn, P, w, (unsigned long long)v + 1ull, h1, h2);
        return 1u;
    };
    if (on && !longp) for (uint64_t k = k0; k < k1; ++k) nz += entry(p, k, (uint64_t)q);
    for (uint64_t lm = __ballot(longp); lm; lm &= lm - 1) {
        const int src = __ffsll((long long)lm) - 1;
        const uint32_t pp = (uint32_t)__shfl((int)p, src, 64), qq = (uint32_t)__shfl((int)q, src, 64);
        const uint64_t a0 = ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(k0 >> 32), src, 64) << 32) | (uint32_t)__shfl((int)(uint32_t)k0, src, 64);
        const uint64_t a1 = ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(k1 >> 32), src, 64) << 32) | (uint32_t)__shfl((int)(uint32_t)k1, src, 64);
        uint32_t z = 0;
        for (uint64_t k = a0 + (uint32_t)lane_id(); k < a1; k += 64) z += entry(pp, k, (uint64_t)qq);
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) z += __shfl_xor(z, d, 64);
        if (lane_id() == src) nz = z;
    }
    if (on && nz != P.part_nz[q]) atomicOr(&Wn.status[w], 1u);
    const uint64_t any = __ballot(one_n != 0);
    if (any) {
        one1 = wave_sum64(one1); one2 = wave_sum64(one2);
        if (lane_id() == 0) abs_sig_add(Wn, P, w, 2ull, one1, one2);
    }
}
// lane per window-table slot: look the variable's signature hash up among the pattern's, take the next free member of its class
__global__ __launch_bounds__(256) void k_abs_match(AbsPattern P, AbsWindows Wn) {
    const uint32_t w = blockIdx.y;
    const uint32_t s = blockIdx.x * 256u + threadIdx.x;
    const unsigned long long key = s < Wn.capW ? Wn.wkey[(size_t)w * Wn.capW + s] : 0ull;
    bool matched = false;
    if (key) {
        const uint64_t h1 = Wn.wh1[(size_t)w * Wn.capW + s], h2 = Wn.wh2[(size_t)w * Wn.capW + s];
        const uint32_t pm = P.capP - 1;
        uint32_t t = (uint32_t)(sig_mix(h1 ^ (h2 * 0x9e3779b97f4a7c15ULL)) & pm), cls = 0;
        for (uint32_t probe = 0; probe <= pm; ++probe) {
            const uint32_t c = P.tab_class[t];
            if (!c) break;
            if (P.tab_h1[t] == h1 && P.tab_h2[t] == h2) { cls = c; break; }
            t = (t + 1) & pm;
        }
        if (!cls) atomicOr(&Wn.status[w], 1u);
        else {
            const uint32_t c0 = P.class_start[cls - 1], c1 = P.class_start[cls];
            const uint32_t k = atomicAdd(&Wn.ccount[(size_t)w * P.nclass + (cls - 1)], 1u);
            if (k >= c1 - c0) atomicOr(&Wn.status[w], 1u);
            else { Wn.phi[(size_t)w * P.nvS + P.class_members[c0 + k]] = (uint32_t)(key - 1ull); matched = true; }
        }
    }
    const uint32_t nk = (uint32_t)__popcll(__ballot(key != 0ull)), nm = (uint32_t)__popcll(__ballot(matched));
    if (lane_id() == 0) {
        if (nk) atomicAdd(&Wn.nvars[w], nk);
        if (nm) atomicAdd(&Wn.nmatched[w], nm);
    }
}
// lane per pattern entry: the window's part must hold (phi(variable), same coefficient). Together with equal non-zero counts
// per part (k_abs_sig) and phi being a bijection (k_abs_match) this is an isomorphism of the two windows.
__global__ __launch_bounds__(256) void k_abs_exact(AbsRowsDev R, AbsPattern P, AbsWindows Wn) {
    const uint32_t w = blockIdx.y;
    const uint32_t e = blockIdx.x * 256u + threadIdx.x;
    if (e >= P.nEnt) return;
    const uint32_t st = Wn.status[w];
    if ((st & 1u) || Wn.nmatched[w] != P.nvS || Wn.nvars[w] != P.nvS) return;      // decided already
    const uint32_t q = P.ent_cnt0[e], jrow = q / 3u, p = q - 3u * jrow;
    const uint64_t row = (uint64_t)Wn.start[w] + jrow;
    const uint32_t v = Wn.phi[(size_t)w * P.nvS + P.ent_var[e]];
    const uint64_t* pc = P.ent_coef + 4ull * e;
    bool found = false;
    for (uint64_t k = R.ptr[p][row]; k < R.ptr[p][row + 1] && !found; ++k) {
        if (R.var[p][k] != v) continue;
        const uint64_t* c = R.coef[p] + 4 * k;
        found = c[0] == pc[0] && c[1] == pc[1] && c[2] == pc[2] && c[3] == pc[3];
        break;      // (a variable occurs once per part)
    }
    if (!found) atomicOr(&Wn.status[w], 2u);
}
__global__ __launch_bounds__(256) void k_abs_io(AbsPattern P, AbsWindows Wn) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= Wn.nwin * P.nio) return;
    const uint32_t w = i / P.nio, t = i - w * P.nio;
    const uint32_t u = P.io_idx[t];
    Wn.io_out[i] = u == 0xFFFFFFFFu ? 0xFFFFFFFFu : Wn.phi[(size_t)w * P.nvS + u];
}

// ---- compaction of the surviving row ranges [a_i, b_i) (rows outside the replaced windows, :368-388)
struct KeepRanges {
    uint32_t n;
    const uint64_t *a, *b;        // row ranges, ascending
    const uint64_t* row0;         // first new row of every range (n + 1 entries)
    const uint64_t* at[3];        // first new entry of every range per part (n + 1 entries)
    const uint64_t* src0[3];      // ptr[p][a_i]
};
__device__ __forceinline__ uint32_t fe_range_of(const uint64_t* starts, uint32_t n, uint64_t x) {   // last i with starts[i] <= x
    uint32_t lo = 0, hi = n;
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (starts[mid] <= x) lo = mid; else hi = mid; }
    return lo;
}
__global__ __launch_bounds__(256) void k_fe_compact_ptr(AbsRowsDev R, KeepRanges K, uint64_t nrow, FeRowsOut O) {
    const uint64_t r = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (r > nrow) return;
    if (r == nrow) { for (int p = 0; p < 3; ++p) O.ptr[p][r] = K.at[p][K.n]; return; }
    const uint32_t g = fe_range_of(K.row0, K.n, r);
    const uint64_t src = K.a[g] + (r - K.row0[g]);
#pragma unroll
    for (int p = 0; p < 3; ++p) O.ptr[p][r] = R.ptr[p][src] - K.src0[p][g] + K.at[p][g];
}
__global__ __launch_bounds__(256) void k_fe_compact_entries(AbsRowsDev R, KeepRanges K, int p, uint64_t nent, FeRowsOut O) {
    for (uint64_t q = (uint64_t)blockIdx.x * 256u + threadIdx.x; q < nent; q += (uint64_t)gridDim.x * 256u) {
        const uint32_t g = fe_range_of(K.at[p], K.n, q);
        const uint64_t src = K.src0[p][g] + (q - K.at[p][g]);
        O.var[p][q] = R.var[p][src];
        const ulonglong2* c = reinterpret_cast<const ulonglong2*>(R.coef[p] + 4 * src);
        ulonglong2* d = reinterpret_cast<ulonglong2*>(O.coef[p] + 4 * q);
        d[0] = c[0]; d[1] = c[1];
    }
}
__global__ void k_fe_gather_u64(const uint64_t* __restrict__ src, const uint64_t* __restrict__ idx, uint32_t n, uint64_t* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = src[idx[i]];
}

// ====================================================================================== 3. layout
struct PartSum {      // one per (part, row): what the row descriptor needs to know about nonzeroKeys(part) in Set order
    uint32_t first_var, first_non1, last_non1;
    uint32_t bits;    // bits 0-1: non-constant variables (capped at 2); bit 2: the constant wire has a non-zero coefficient;
                      // bit 3: the dictionary holds key 1 (any value)
};
struct LayTemp {      // device scratch of the layout (pointers carved from one allocation)
    uint32_t* nzc;      // [3][nC + 1] non-zero count per part -> exclusive scan (the CSR row pointers)
    PartSum* sum;       // [3][nC]
    uint32_t* midlist; uint32_t* largelist;
    uint32_t *f_p4, *f_cls, *f_big, *f_val, *f_p5;   // per-row flags -> exclusive scans ([nC + 1] each)
    uint8_t* aeq;       // A-map of row i equals A-map of row i + 1, zeros included (:1512)
    uint64_t *pairs, *pairs2;   // (variable << 32 | row) per non-zero entry; sorted
    uint32_t* f_uniq;   // first occurrence of a (variable, row) pair -> scan
    uint32_t* deg;      // [nVall + 3] rows per variable -> scan = fo_ptr
};

// non-zero count per (part, row); largest variable id over ALL dictionary entries; parts too long for a lane
__global__ __launch_bounds__(256) void k_lay_count(AbsRowsDev R, uint32_t nC, uint32_t* __restrict__ nzc, uint32_t* __restrict__ midlist,
                                                   uint32_t* __restrict__ largelist, uint32_t* __restrict__ hugelist, FeMeta* M) {
    __shared__ uint32_t s_nz[3], s_mv, s_mc, s_mn;
    if (threadIdx.x < 3) s_nz[threadIdx.x] = 0;
    if (threadIdx.x == 3) { s_mv = 0; s_mc = 0; s_mn = 0; }
    __syncthreads();
    uint32_t mv = 0, mc = 0, mn = 0, nz3[3] = {0, 0, 0};
    for (uint32_t i0 = blockIdx.x * 256u + (threadIdx.x & ~63u); i0 < 3u * nC; i0 += gridDim.x * 256u) {      // (uniform per wavefront: shuffles inside)
        const uint32_t i = i0 + (threadIdx.x & 63u);
        const bool on = i < 3u * nC;
        const uint32_t r = on ? i / 3u : 0u, p = on ? i - 3u * r : 0u;
        uint32_t nz = 0;
        const uint64_t ka = on ? R.ptr[p][r] : 0ull, kz = on ? R.ptr[p][r + 1] : 0ull;
        const bool longp = kz - ka > 48;      // (round 6) a long part is walked by the whole wavefront, not by its lane: 1 025 entries one after the other were 0.5 ms
        if (!longp)
        for (uint64_t k = ka; k < kz; ++k) {
            const uint32_t v = R.var[p][k];
            mv = v > mv ? v : mv;
            const uint64_t* c = R.coef[p] + 4 * k;
            nz += (c[0] | c[1] | c[2] | c[3]) != 0;
        }
        for (uint64_t lm = __ballot(longp); lm; lm &= lm - 1) {      // (the lanes of a wavefront that are in the loop at all reach this together: i advances by the same stride)
            const int src = __ffsll((long long)lm) - 1;
            const uint32_t pp = (uint32_t)__shfl((int)p, src, 64);
            const uint64_t a0 = ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(ka >> 32), src, 64) << 32) | (uint32_t)__shfl((int)(uint32_t)ka, src, 64);
            const uint64_t a1 = ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(kz >> 32), src, 64) << 32) | (uint32_t)__shfl((int)(uint32_t)kz, src, 64);
            uint32_t z = 0;
            for (uint64_t k = a0 + (threadIdx.x & 63u); k < a1; k += 64) {
                const uint32_t v = R.var[pp][k];
                mv = v > mv ? v : mv;
                const uint64_t* c = R.coef[pp] + 4 * k;
                z += (c[0] | c[1] | c[2] | c[3]) != 0;
            }
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) z += __shfl_xor(z, d, 64);
            if ((int)(threadIdx.x & 63u) == src) nz = z;
        }
        if (!on) continue;
        nzc[(size_t)p * (nC + 1) + r] = nz;
        if (nz > FE_LARGE_MAX) hugelist[atomicAdd(&M->n_huge, 1u)] = i;
        else if (nz > FE_MID_MAX) largelist[atomicAdd(&M->n_large, 1u)] = i;
        else if (nz > FE_LANE_MAX) midlist[atomicAdd(&M->n_mid, 1u)] = i;
        if (p == 0) nz3[0] += nz; else if (p == 1) nz3[1] += nz; else { nz3[2] += nz; mc = nz > mc ? nz : mc; }
        mn = nz > mn ? nz : mn;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        uint32_t y = __shfl_xor(mv, d, 64); mv = y > mv ? y : mv;
        y = __shfl_xor(mc, d, 64); mc = y > mc ? y : mc;
        y = __shfl_xor(mn, d, 64); mn = y > mn ? y : mn;
        nz3[0] += __shfl_xor(nz3[0], d, 64); nz3[1] += __shfl_xor(nz3[1], d, 64); nz3[2] += __shfl_xor(nz3[2], d, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMax(&s_mv, mv); atomicMax(&s_mc, mc); atomicMax(&s_mn, mn);
        for (int p = 0; p < 3; ++p) if (nz3[p]) atomicAdd(&s_nz[p], nz3[p]);
    }
    __syncthreads();
    // one device atomic per workgroup and word (millions of lanes adding to three words cost 10 ms)
    if (threadIdx.x < 3 && s_nz[threadIdx.x]) atomicAdd(&M->nnz[threadIdx.x], (unsigned long long)s_nz[threadIdx.x]);
    if (threadIdx.x == 3) {
        if (s_mv) atomicMax(&M->maxvar, s_mv);
        if (s_mc) atomicMax(&M->maxlenC, s_mc);
        if (s_mn) atomicMax(&M->maxn, s_mn);
    }
}

struct LayCsr { uint32_t* rp[3]; uint32_t* col[3]; uint64_t* coef[3]; };

__device__ __forceinline__ void laysum_add(PartSum& S, uint32_t v, uint32_t m) {
    if (m == 0) S.first_var = v;
    if (v == 1u) S.bits |= 4u;
    else {
        const uint32_t c = S.bits & 3u;
        if (c == 0) S.first_non1 = v;
        S.last_non1 = v;
        if (c < 2) S.bits = (S.bits & ~3u) | (c + 1);
    }
}

// nonzeroKeys(part) in Set order for parts of up to FE_LANE_MAX non-zero entries: one lane per part
__global__ __launch_bounds__(256) void k_lay_order_small(AbsRowsDev R, uint32_t nC, const uint32_t* __restrict__ rp, LayCsr L,
                                                         PartSum* __restrict__ sum, uint8_t* __restrict__ nontrivial) {
    __shared__ uint32_t s_key[16 * 256], s_pay[16 * 256];
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= 3u * nC) return;
    const uint32_t r = i / 3u, p = i - 3u * r;
    const size_t q = (size_t)p * (nC + 1) + r;
    const uint32_t at = rp[q], nz = rp[q + 1] - at;
    if (nz > FE_LANE_MAX) return;
    const uint64_t k0 = R.ptr[p][r], k1 = R.ptr[p][r + 1];
    PartSum S;
    S.first_var = S.first_non1 = S.last_non1 = 0; S.bits = 0;
    jlslot::TabT<FeLdsPtr> t;
    t.key = (FeLdsPtr)(s_key + threadIdx.x); t.pay = (FeLdsPtr)(s_pay + threadIdx.x); t.key2 = nullptr; t.pay2 = nullptr;
    t.cap = 16; t.stride = 256;
    if (nz > 1) jlslot::tab_init(t);
    uint32_t m = 0;
    for (uint64_t k = k0; k < k1; ++k) {
        const uint32_t v = R.var[p][k];
        if (v == 1u) S.bits |= 8u;
        const uint64_t* c = R.coef[p] + 4 * k;
        if ((c[0] | c[1] | c[2] | c[3]) == 0) continue;
        if (nz > 1) (void)jlslot::tab_upsert<0>(t, v, (uint32_t)(k - k0));
        else {
            L.col[p][at] = v;
            st256(L.coef[p] + 4ull * at, ld256(c));
            nontrivial[v] = 1;
            laysum_add(S, v, 0);
        }
    }
    if (nz > 1)
        for (uint32_t s = 0; s < 16; ++s) {
            const uint32_t py = t.pay[s * 256];
            if (!py) continue;
            const uint64_t k = k0 + (py - 1u);
            const uint32_t v = R.var[p][k];
            L.col[p][at + m] = v;
            st256(L.coef[p] + 4ull * (at + m), ld256(R.coef[p] + 4 * k));
            nontrivial[v] = 1;
            laysum_add(S, v, m);
            ++m;
        }
    sum[(size_t)p * nC + r] = S;
}
// ... and for longer parts: one wavefront per part (lane 0 replays the Set insertions, all lanes emit the slots)
template <int TIER>
__global__ __launch_bounds__(64 * FeTier<TIER>::waves) void k_lay_order_big(AbsRowsDev R, uint32_t nC, const uint32_t* __restrict__ rp, LayCsr L, PartSum* __restrict__ sum,
                                                       uint8_t* __restrict__ nontrivial, const uint32_t* __restrict__ list, uint32_t nlist, FeMeta* M,
                                                       uint32_t* gscratch, uint32_t gcap, uint32_t* __restrict__ overflow_list, uint32_t* overflow_count) {
    constexpr bool LDS_TABLES = TIER != 0;
    constexpr uint32_t WAVES = FeTier<TIER>::waves;
    uint32_t* const s_tab = reinterpret_cast<uint32_t*>(ecne_dyn_lds);
    const int lane = lane_id();
    const uint32_t wave = threadIdx.x >> 6, gw = blockIdx.x * WAVES + wave, nw = gridDim.x * WAVES;
    uint32_t* base = LDS_TABLES ? s_tab + (size_t)wave * FeTier<TIER>::words : gscratch + (size_t)gw * 4 * gcap;
    const uint32_t cap = LDS_TABLES ? FeTier<TIER>::cap : gcap;
    for (uint32_t b = gw; b < nlist; b += nw) {
        const uint32_t i = list[b];
        const uint32_t r = i / 3u, p = i - 3u * r;
        const size_t q = (size_t)p * (nC + 1) + r;
        const uint32_t at = rp[q];
        const uint64_t k0 = R.ptr[p][r], k1 = R.ptr[p][r + 1];
        uint32_t sz = 0, flipped = 0, bad = 0, haskey1 = 0;
        typedef typename FeTabPtr<LDS_TABLES>::type TP;
        const TP tb = (TP)base;
        jlslot::TabT<TP> t;
        t.key = tb; t.pay = tb + cap; t.key2 = tb + 2 * cap; t.pay2 = tb + 3 * cap;
        t.cap = cap; t.stride = 1;
        if constexpr (LDS_TABLES) {
            // (round 6) the non-zero entries staged in order (variable, hash, position) by all lanes, the table built by fe_wave_table
            const FeLdsPtr st = (FeLdsPtr)base + 4 * FeTier<TIER>::cap;
            constexpr uint32_t SG = FeTier<TIER>::stage;
            uint32_t cnt = 0;
            for (uint64_t kb = k0; kb < k1; kb += 64) {
                const uint64_t k = kb + lane;
                bool nzq = false;
                uint32_t v = 0;
                if (k < k1) {
                    v = R.var[p][k];
                    const uint64_t* c = R.coef[p] + 4 * k;
                    nzq = (c[0] | c[1] | c[2] | c[3]) != 0;
                }
                if (__ballot(k < k1 && v == 1u)) haskey1 = 1;
                const uint64_t mask = __ballot(nzq);
                const uint32_t o = cnt + (uint32_t)__popcll(mask & lanes_below());
                if (nzq && o < SG) { st[o] = v; st[SG + o] = (uint32_t)jlslot::hash64((uint64_t)v); st[2 * SG + o] = (uint32_t)(k - k0); }
                cnt += (uint32_t)__popcll(mask);
            }
            jlslot::Tab64T<FeLdsPtr64> t6;
            t6.cur = (FeLdsPtr64)base; t6.nxt = (FeLdsPtr64)base + cap; t6.cap = cap;
            bad = cnt > SG ? 1u : fe_wave_table<0>(t6, cnt, st, st + SG, st + 2 * SG, st + 3 * SG, st + 4 * SG, st + 5 * SG);
            sz = t6.sz;
            flipped = t6.cur != (FeLdsPtr64)base;
        } else {
        if (lane == 0) jlslot::tab_init(t);
        // all lanes look at 64 dictionary entries at a time (non-zero? key 1?), lane 0 inserts the non-zero ones in order
        for (uint64_t kb = k0; kb < k1; kb += 64) {
            const uint64_t k = kb + lane;
            bool nzq = false;
            uint32_t v = 0;
            if (k < k1) {
                v = R.var[p][k];
                const uint64_t* c = R.coef[p] + 4 * k;
                nzq = (c[0] | c[1] | c[2] | c[3]) != 0;
            }
            if (__ballot(k < k1 && v == 1u)) haskey1 = 1;
            uint64_t mask = __ballot(nzq);
            while (mask) {
                const int src = __ffsll((long long)mask) - 1;
                mask &= mask - 1;
                const uint32_t vv = __shfl(v, src, 64);
                if (lane == 0 && !bad) bad = jlslot::tab_upsert<0>(t, vv, (uint32_t)(kb + src - k0)) != 0;
            }
        }
        }
        if (!LDS_TABLES && lane == 0) { sz = t.sz; flipped = t.key != tb; }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        sz = __shfl(sz, 0, 64); flipped = __shfl(flipped, 0, 64); bad = __shfl(bad, 0, 64);
        if (bad) {
            if (lane == 0) { if (LDS_TABLES) overflow_list[atomicAdd(overflow_count, 1u)] = i; else atomicOr(&M->unsupported, 1u); }
            continue;
        }
        // (LDS tiers: the occupied 64-bit slots collected in slot order first, the emission over full wavefronts -- as in k_fe_fill_big)
        const uint32_t* pay = flipped ? base + 3 * cap : base + cap;
        uint32_t n_em = sz;
        const uint32_t* const cpay = base + 4 * FeTier<TIER>::cap + 4 * FeTier<TIER>::stage;
        if constexpr (LDS_TABLES) {
            const FeLdsPtr64 slots = (FeLdsPtr64)(flipped ? base + 2 * cap : base);
            const FeLdsPtr lp = (FeLdsPtr)base + 4 * FeTier<TIER>::cap + 4 * FeTier<TIER>::stage;
            uint32_t cnt = 0;
            for (uint32_t s0 = 0; s0 < sz; s0 += 64) {
                const uint32_t s = s0 + lane;
                const uint64_t sv = s < sz ? slots[s] : 0ull;
                const uint64_t mk = __ballot((uint32_t)(sv >> 32) != 0);
                if ((uint32_t)(sv >> 32)) lp[cnt + (uint32_t)__popcll(mk & lanes_below())] = (uint32_t)(sv >> 32);
                cnt += (uint32_t)__popcll(mk);
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            n_em = cnt;
        }
        uint32_t m = 0, first_var = 0, first_non1 = 0, last_non1 = 0, n_non1 = 0, has1 = 0;
        for (uint32_t s0 = 0; s0 < n_em; s0 += 64) {
            const uint32_t s = s0 + lane;
            const uint32_t py = s < n_em ? (LDS_TABLES ? cpay[s] : ld32_agent(&pay[s])) : 0u;
            const uint64_t mask = __ballot(py != 0);
            uint32_t v = 0;
            if (py) {
                const uint64_t k = k0 + (py - 1u);
                v = R.var[p][k];
                const uint32_t o = at + m + (uint32_t)__popcll(mask & lanes_below());
                L.col[p][o] = v;
                st256(L.coef[p] + 4ull * o, ld256(R.coef[p] + 4 * k));
                nontrivial[v] = 1;
            }
            if (mask) {
                if (m == 0) first_var = __shfl(v, __ffsll((long long)mask) - 1, 64);
                const uint64_t non1 = __ballot(py != 0 && v != 1u);
                if (__ballot(py != 0 && v == 1u)) has1 = 1;
                if (non1) {
                    if (n_non1 == 0) first_non1 = __shfl(v, __ffsll((long long)non1) - 1, 64);
                    last_non1 = __shfl(v, 63 - __clzll((long long)non1), 64);
                    n_non1 += (uint32_t)__popcll(non1);
                }
            }
            m += (uint32_t)__popcll(mask);
        }
        if (lane == 0) {
            PartSum S;
            S.first_var = first_var; S.first_non1 = first_non1; S.last_non1 = last_non1;
            S.bits = (n_non1 > 2 ? 2u : n_non1) | (has1 ? 4u : 0u) | (haskey1 ? 8u : 0u);
            sum[(size_t)p * nC + r] = S;
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    }
}

// The structural row descriptor (what build_layout in ecne_engine.hip lays down on the host): one lane per row.
// Flags out: f_p4 / f_cls / f_big / f_val (0 / 1 per row, scanned afterwards), aeq.
__global__ __launch_bounds__(256) void k_lay_rows(AbsRowsDev R, uint32_t nC, const uint32_t* __restrict__ rp, const PartSum* __restrict__ sum,
                                                  RowInfo* __restrict__ rinfo, uint32_t* __restrict__ f_p4, uint32_t* __restrict__ f_cls,
                                                  uint32_t* __restrict__ f_big, uint32_t* __restrict__ f_val, uint8_t* __restrict__ aeq, FeMeta* __restrict__ M) {
    const uint32_t r = blockIdx.x * 256u + threadIdx.x;
    if (r >= nC) return;
    PartSum S[3];
    uint32_t n[3], dl[3];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        S[p] = sum[(size_t)p * nC + r];
        const size_t q = (size_t)p * (nC + 1) + r;
        n[p] = rp[q + 1] - rp[q];
        dl[p] = (uint32_t)(R.ptr[p][r + 1] - R.ptr[p][r]);
    }
    RowInfo ri;
    ri.shape = 0; ri.x = 0; ri.kpos = 0; ri.kneg = 0; ri.k1 = 0; ri.k2 = 0; ri.validx = 0xFFFFFFFFu;
    ri.lenC = n[2];
    uint32_t shape = 0;
    if (S[2].bits & 4u) shape |= SH_C_HAS1;
    if (n[0] + n[1] + n[2] > ECNE_SMALL_ROW) shape |= SH_BIG;
    if (n[0] || n[1]) shape |= SH_HAS_AB;
    bool p4 = false;
    if (n[2] == 0) {
        shape |= SH_C_EMPTY;
        const uint32_t cA = S[0].bits & 3u, cB = S[1].bits & 3u;
        const uint32_t a = cA ? S[0].first_non1 : 0u, b = cB ? S[1].first_non1 : 0u;
        uint32_t distinct;
        if (cA == 2 || cB == 2) distinct = 2;
        else if (cA + cB == 0) distinct = 0;
        else distinct = (cA && cB && a != b) ? 2u : 1u;
        if (distinct == 0) shape |= SH_R2_BOUNDSERR;
        else if (distinct == 1) {
            shape |= SH_R2;
            const uint32_t x = cA ? a : b;
            ri.x = x;
            const bool inA = cA == 1 && a == x, inB = cB == 1 && b == x;
            if (!inA || !inB) shape |= SH_R2_DIV0;
        }
        if (n[1] == 1 && n[0] <= 2) {      // P4 static test (:1427-1466)
            shape |= SH_P4;
            ri.kpos = S[1].first_var;
            ri.kneg = cA ? S[0].last_non1 : 0u;      // the last non-constant variable of A wins (:1462-1465)
            if (!cA) shape |= SH_P4_DIV0;
            p4 = true;
        }
    }
    if (!(shape & SH_HAS_AB) && n[2] > 0) {
        if ((S[2].bits & 3u) == 1) { shape |= SH_R3; ri.x = S[2].last_non1; }      // (:949-960)
        const uint32_t zc = dl[2] - n[2];
        const uint32_t czero_eff = zc + (((shape & SH_R3) && !(S[2].bits & 8u)) ? 1u : 0u);
        if (czero_eff) shape |= SH_CZERO;
        const uint64_t d0 = R.ptr[2][r];
        const fp::u256 ONE = fp::make(1), PM1 = fp::pminus1();
        if (!czero_eff && dl[2] == 2) {      // R5 in dictionary order (:1082-1092)
            const fp::u256 u = ld256(R.coef[2] + 4 * d0), v = ld256(R.coef[2] + 4 * (d0 + 1));
            if ((fp::eq(u, ONE) && fp::eq(v, PM1)) || (fp::eq(u, PM1) && fp::eq(v, ONE))) {
                shape |= SH_R5;
                ri.k1 = R.var[2][d0];
                ri.k2 = R.var[2][d0 + 1];
            }
        }
        if (!czero_eff && dl[2] == 3) {      // R6 (:1154-1173)
            int ones = 0, mones = 0;
            bool one_on_const = true;
            uint32_t k1 = 0, k2 = 0;
            for (uint64_t k = d0; k < d0 + 3; ++k) {
                const fp::u256 c = ld256(R.coef[2] + 4 * k);
                if (fp::eq(c, ONE)) { ones++; if (R.var[2][k] != 1u) one_on_const = false; }
                else if (fp::eq(c, PM1)) { if (mones == 0) k1 = R.var[2][k]; else k2 = R.var[2][k]; mones++; }
            }
            if (ones == 1 && mones == 2 && one_on_const) { shape |= SH_R6; ri.k1 = k1; ri.k2 = k2; }
        }
        if ((shape & (SH_R5 | SH_R6)) && ri.k1 != ri.k2 && jlslot::pair_second_first(ri.k1, ri.k2)) shape |= SH_R56_SWAP;
    }
    // secp_solve's dsu setup (:634-678): a two-entry C without A and B whose non-zero keys are none (`l[1]`, :650) or the constant
    // wire alone (`l[2]`, :652) raises BoundsError
    if (n[0] == 0 && n[1] == 0 && dl[2] == 2 && (n[2] == 0 || (n[2] == 1 && (S[2].bits & 3u) == 0))) M->dsu_err = 1u;
    ri.shape = shape;
    rinfo[r] = ri;
    f_p4[r] = p4 ? 1u : 0u;
    f_cls[r] = n[2] > ECNE_CLS_LANE ? 1u : 0u;
    f_big[r] = (shape & SH_BIG) ? 1u : 0u;
    f_val[r] = ((shape & SH_R2) || (!(shape & SH_HAS_AB) && n[2] > 0)) ? 1u : 0u;
    // A-map equality with the next row, zeros included (:1512): same keys, same values (position by position first)
    uint8_t eq = 0;
    if (r + 1 < nC) {
        const uint64_t x0 = R.ptr[0][r], x1 = R.ptr[0][r + 1], y1 = R.ptr[0][r + 2];
        bool e = (x1 - x0) == (y1 - x1);
        const uint64_t len = x1 - x0;
        for (uint64_t k = 0; e && k < len; ++k) {
            const uint32_t v = R.var[0][x0 + k];
            bool found = false;
            for (uint64_t t = 0; t < len; ++t) {
                const uint64_t m = x1 + ((k + t) % len);
                if (R.var[0][m] == v) { found = fp::eq(ld256(R.coef[0] + 4 * m), ld256(R.coef[0] + 4 * (x0 + k))); break; }
            }
            e = found;
        }
        eq = e ? 1 : 0;
    }
    aeq[r] = eq;
}
// value slots, the P4 / long-row lists (flags scanned: f_x[r] = rows before r that carry the flag)
__global__ __launch_bounds__(256) void k_lay_lists(uint32_t nC, RowInfo* __restrict__ rinfo, const uint32_t* __restrict__ f_p4, const uint32_t* __restrict__ f_cls,
                                                   const uint32_t* __restrict__ f_big, const uint32_t* __restrict__ f_val, LayoutDst D) {
    const uint32_t r = blockIdx.x * 256u + threadIdx.x;
    if (r >= nC) return;
    const RowInfo ri = rinfo[r];
    if (f_val[r + 1] != f_val[r]) rinfo[r].validx = 2u * f_val[r];
    if (f_p4[r + 1] != f_p4[r]) {
        const uint32_t o = f_p4[r];
        D.p4_list[o] = r;
        D.p4_b[o] = ri.kpos;
        D.p4_s[o] = ri.kneg | ((ri.shape & SH_P4_DIV0) ? 0x80000000u : 0u);
    }
    if (f_cls[r + 1] != f_cls[r]) D.cls_list[f_cls[r]] = r;
    uint16_t tb = 0;
    if (f_big[r + 1] != f_big[r]) {
        const uint32_t o = f_big[r];
        D.long_list[o] = r;
        if (o < ECNE_BIGTAB) { D.bigrows[o] = r; tb = (uint16_t)(o + 1); }
    }
    D.tbig[r] = tb;
}
// P5 static candidates (:1492-1536)
__global__ __launch_bounds__(256) void k_lay_p5_flag(uint32_t nC, const uint32_t* __restrict__ rp, LayCsr L, const uint8_t* __restrict__ aeq,
                                                     uint32_t* __restrict__ f_p5) {
    const uint32_t r = blockIdx.x * 256u + threadIdx.x;
    if (r >= nC) return;
    uint32_t f = 0;
    if (r + 1 < nC) {
        const uint32_t* rpB = rp + (size_t)(nC + 1);
        const uint32_t* rpC = rp + 2 * (size_t)(nC + 1);
        const uint32_t nc_next = rpC[r + 2] - rpC[r + 1], nb_next = rpB[r + 2] - rpB[r + 1], nc_this = rpC[r + 1] - rpC[r];
        if (nc_next == 0 && nb_next == 1 && nc_this == 2 && aeq[r]) {
            const uint32_t y = L.col[1][rpB[r + 1]];
            if (y != 1u) {
                bool bad = false;
                for (uint32_t k = rpC[r]; k < rpC[r + 1]; ++k) { const uint32_t v = L.col[2][k]; if (v != 1u && v != y) bad = true; }
                f = bad ? 0u : 1u;
            }
        }
    }
    f_p5[r] = f;
}
__global__ __launch_bounds__(256) void k_lay_p5_write(uint32_t nC, const uint32_t* __restrict__ rp, LayCsr L, const uint32_t* __restrict__ f_p5, LayoutDst D) {
    const uint32_t r = blockIdx.x * 256u + threadIdx.x;
    if (r >= nC || f_p5[r + 1] == f_p5[r]) return;
    const uint32_t* rpB = rp + (size_t)(nC + 1);
    D.p5_rows[f_p5[r]] = r;
    D.p5_y[f_p5[r]] = L.col[1][rpB[r + 1]];
}
// (variable << 32 | row) for every non-zero entry (variable_to_indices :628-633 lists the rows a variable occurs in)
__global__ __launch_bounds__(256) void k_lay_pairs(uint32_t nC, const uint32_t* __restrict__ rp, LayCsr L, uint64_t off1, uint64_t off2, uint64_t* __restrict__ pairs) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= 3u * nC) return;
    const uint32_t r = i / 3u, p = i - 3u * r;
    const size_t q = (size_t)p * (nC + 1) + r;
    const uint64_t off = p == 0 ? 0ull : p == 1 ? off1 : off2;
    for (uint32_t k = rp[q]; k < rp[q + 1]; ++k) pairs[off + k] = ((uint64_t)L.col[p][k] << 32) | r;
}
__global__ __launch_bounds__(256) void k_lay_uniq_flag(const uint64_t* __restrict__ sorted, uint32_t n, uint32_t* __restrict__ f_uniq) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    f_uniq[i] = (i == 0 || sorted[i - 1] != sorted[i]) ? 1u : 0u;
}
// fo_rows = the rows of the distinct pairs in sorted order; fo_ptr[v] = number of distinct pairs of variables below v, written at
// the run boundaries of the sorted list (no per-variable atomics: the constant wire alone has 10^5 rows). nvar1 = entries of fo_ptr.
__global__ __launch_bounds__(256) void k_lay_fo_rows(const uint64_t* __restrict__ sorted, uint32_t n, const uint32_t* __restrict__ f_uniq, uint32_t* __restrict__ fo_rows,
                                                     uint32_t* __restrict__ fo_ptr, uint32_t nvar1) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint32_t q = f_uniq[i];
    const uint64_t x = sorted[i];
    const uint32_t v = (uint32_t)(x >> 32);
    if (i == n - 1) {
        const uint32_t tot = f_uniq[n];
        for (uint32_t u = v + 1u; u < nvar1; ++u) fo_ptr[u] = tot;
    }
    if (f_uniq[i + 1] == q) return;      // a repeat of the pair before it
    fo_rows[q] = (uint32_t)x;
    const uint32_t vprev = i == 0 ? 0xFFFFFFFFu : (uint32_t)(sorted[i - 1] >> 32);
    if (i == 0 || vprev != v)
        for (uint32_t u = vprev + 1u; u <= v && u < nvar1; ++u) fo_ptr[u] = q;      // v's first pair: v and the unused ids below it start here
}
// row records and inline fan-out lists (chain executor / fast rounds, engine_types.hpp)
__global__ __launch_bounds__(256) void k_lay_rec(uint32_t nC, const uint32_t* __restrict__ rp, LayCsr L, uint32_t* __restrict__ rec) {
    const uint32_t r = blockIdx.x * 256u + threadIdx.x;
    if (r >= nC) return;
    uint32_t w[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) w[k] = 0;
    uint32_t a[3], l[3];
#pragma unroll
    for (int p = 0; p < 3; ++p) { const size_t q = (size_t)p * (nC + 1) + r; a[p] = rp[q]; l[p] = rp[q + 1] - a[p]; }
    if (l[0] + l[1] + l[2] > 15) w[1] = 0xFFFFFFFFu;
    else {
        w[0] = l[0] | l[1] << 8 | l[2] << 16 | 1u << 24;
        uint32_t k = 1;
#pragma unroll
        for (int p = 0; p < 3; ++p)
            for (uint32_t e = 0; e < l[p]; ++e) {
                const uint32_t v = L.col[p][a[p] + e];
#pragma unroll
                for (int s = 1; s < 16; ++s) if ((uint32_t)s == k) w[s] = v;
                ++k;
            }
    }
    uint4* d = reinterpret_cast<uint4*>(rec + 16ull * r);
    d[0] = make_uint4(w[0], w[1], w[2], w[3]);
    d[1] = make_uint4(w[4], w[5], w[6], w[7]);
    d[2] = make_uint4(w[8], w[9], w[10], w[11]);
    d[3] = make_uint4(w[12], w[13], w[14], w[15]);
}
__global__ __launch_bounds__(256) void k_lay_foi(uint32_t nvar, const uint32_t* __restrict__ fo_ptr, const uint32_t* __restrict__ fo_rows, uint32_t* __restrict__ foi) {
    const uint32_t v = blockIdx.x * 256u + threadIdx.x;
    if (v > nvar) return;
    uint4 w = make_uint4(0, 0, 0, 0);
    if (v < nvar) {
        const uint32_t f0 = fo_ptr[v], n = fo_ptr[v + 1] - f0;
        w.x = n;
        if (n <= 3) {
            if (n > 0) w.y = fo_rows[f0];
            if (n > 1) w.z = fo_rows[f0 + 1];
            if (n > 2) w.w = fo_rows[f0 + 2];
        } else w.y = f0;
    }
    reinterpret_cast<uint4*>(foi)[v] = w;
}
__global__ void k_mark_bytes(uint8_t* dst, const uint32_t* ids, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[ids[i]] = 1;
}
// "Bad Constraints" (:1609-1618): rows that hold a variable without the unique bit
__global__ __launch_bounds__(256) void k_bad_flag(Job J, uint32_t* __restrict__ f_bad) {
    const uint32_t r = blockIdx.x * 256u + threadIdx.x;
    if (r >= J.nC) return;
    bool bad = false;
    for (uint32_t k = J.rpA[r]; k < J.rpA[r + 1] && !bad; ++k) bad = !(J.flags[J.colA[k]] & 1);
    for (uint32_t k = J.rpB[r]; k < J.rpB[r + 1] && !bad; ++k) bad = !(J.flags[J.colB[k]] & 1);
    for (uint32_t k = J.rpC[r]; k < J.rpC[r + 1] && !bad; ++k) bad = !(J.flags[J.colC[k]] & 1);
    f_bad[r] = bad ? 1u : 0u;
}
__global__ __launch_bounds__(256) void k_bad_write(uint32_t nC, const uint32_t* __restrict__ f_bad, uint32_t* __restrict__ out) {
    const uint32_t r = blockIdx.x * 256u + threadIdx.x;
    if (r >= nC || f_bad[r + 1] == f_bad[r]) return;
    out[f_bad[r]] = r;
}

}  // namespace fe
}  // namespace ecne
