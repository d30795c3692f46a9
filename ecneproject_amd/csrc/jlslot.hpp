// jlslot.hpp — the slot order of a Julia 1.7 hash table keyed by Int64, on caller-provided storage.
//
// Same model as jlorder.hpp (base/dict.jl of Julia 1.7: open addressing, linear probing, 16 slots to start
// with, home slot = hash_64_64(key) & (size - 1), x4 growth (x2 past 64 000 keys) when more than 2/3 full or
// when a probe sequence reaches max(16, size / 64), a growth re-inserts the old slots in ascending slot order,
// iteration = ascending slot order) -- but written so that the device front-end (frontend.hip.hpp: the parse
// kernels reproduce the DefaultDict order of ParseR1CS.jl:94-111, the layout kernels the Set order of
// nonzeroKeys, R1CSConstraintSolver.jl:26-34) can run it per lane / per wavefront on LDS or HBM storage. Compiled for
// the host too: tests/test_jlslot_host.py checks it against jl::SlotTable on random key sequences.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define JLQ __host__ __device__ __forceinline__
#else
#define JLQ inline
#endif

namespace jlslot {

JLQ uint64_t hash64(uint64_t a) {
    a = ~a + (a << 21);
    a ^= a >> 24;
    a += (a << 3) + (a << 8);
    a ^= a >> 14;
    a += (a << 2) + (a << 4);
    a ^= a >> 28;
    a += a << 31;
    return a;
}

// key[] holds the stored 32-bit key, pay[] = payload + 1 (0 = empty slot). Two buffers of `cap` slots each: a growth
// re-inserts from one into the other. STRIDE: distance between consecutive slots in the arrays (per-lane tables are
// interleaved in LDS so that the lanes of a wavefront hit different banks).
// P: the pointer type of the storage -- plain `uint32_t*` on the host and for tables in device memory; the device front-end passes
// LDS-address-space pointers for its LDS tables, so that the walk below is ds_read / ds_write and not flat accesses (a flat access
// to LDS costs what an L2 hit costs, and lane 0 replays up to 1 025 insertions one dependent probe after the other).
template <class P = uint32_t*>
struct TabT {
    P key, pay, key2, pay2;
    uint32_t cap;        // slots per buffer
    uint32_t stride;
    uint32_t sz, n, maxprobe;
};
using Tab = TabT<>;

template <class P>
JLQ void tab_init(TabT<P>& t) {
    t.sz = 16; t.n = 0; t.maxprobe = 0;
    for (uint32_t i = 0; i < 16; ++i) t.pay[i * t.stride] = 0;
}

// ADD: the hashed Int64 is (stored key + ADD) -- the parser hashes wire id + 1 (which may be 2^32), everybody else the id itself
template <int ADD, class P>
JLQ int tab_grow(TabT<P>& t, uint64_t want) {
    uint64_t nsz = 16;
    while (nsz < want) nsz <<= 1;
    if (nsz > t.cap) return -1;
    const uint32_t st = t.stride, mask = (uint32_t)nsz - 1;
    for (uint32_t i = 0; i < (uint32_t)nsz; ++i) t.pay2[i * st] = 0;
    uint32_t mp = 0;
    for (uint32_t i = 0; i < t.sz; ++i) {
        const uint32_t p = t.pay[i * st];
        if (!p) continue;
        const uint32_t k = t.key[i * st];
        const uint32_t home = (uint32_t)(hash64((uint64_t)k + (uint64_t)ADD) & mask);
        uint32_t idx = home;
        while (t.pay2[idx * st]) idx = (idx + 1) & mask;
        const uint32_t probe = (idx - home) & mask;
        if (probe > mp) mp = probe;
        t.pay2[idx * st] = p;
        t.key2[idx * st] = k;
    }
    P a = t.key; t.key = t.key2; t.key2 = a;
    a = t.pay; t.pay = t.pay2; t.pay2 = a;
    t.sz = (uint32_t)nsz;
    t.maxprobe = mp;
    return 0;
}

// insert `skey` with `payload`, or replace the payload of the slot that holds it already ("last wins" at the first
// occurrence's position, ParseR1CS.jl:104-108). Returns 0, or -1 when the table would have to grow beyond `cap`.
template <int ADD, class P>
JLQ int tab_upsert(TabT<P>& t, uint32_t skey, uint32_t payload) {
    const uint64_t hk = (uint64_t)skey + (uint64_t)ADD;
    const uint32_t st = t.stride;
    for (;;) {
        const uint32_t sz = t.sz, mask = sz - 1;
        uint32_t idx = (uint32_t)(hash64(hk) & mask), it = 0;
        bool found_empty = false;
        for (;;) {
            if (!t.pay[idx * st]) { found_empty = true; break; }
            if (t.key[idx * st] == skey) { t.pay[idx * st] = payload + 1; return 0; }
            idx = (idx + 1) & mask;
            if (++it > t.maxprobe) break;
        }
        if (!found_empty) {
            const uint32_t lim = (sz >> 6) > 16 ? (sz >> 6) : 16;
            while (it < lim) {
                if (!t.pay[idx * st]) { found_empty = true; t.maxprobe = it; break; }
                idx = (idx + 1) & mask;
                ++it;
            }
        }
        if (!found_empty) {
            if (tab_grow<ADD>(t, t.n > 64000 ? (uint64_t)sz * 2 : (uint64_t)sz * 4)) return -1;
            continue;
        }
        t.key[idx * st] = skey;
        t.pay[idx * st] = payload + 1;
        ++t.n;
        if ((uint64_t)t.n * 3 > (uint64_t)sz * 2)
            if (tab_grow<ADD>(t, t.n > 64000 ? (uint64_t)t.n * 2 : (uint64_t)t.n * 4)) return -1;
        return 0;
    }
}

// tab_upsert with the key's hash at hand (its low 32 bits: a table has at most 2^31 slots) and WITHOUT the growth: 0 = done, 1 = nothing
// stored, grow to *want slots and call again with the same key, 2 = stored, then grow to *want. The device front-end's long parts grow their
// tables with the whole wavefront (frontend.hip.hpp, fe_wave_table): same table as tab_upsert's, slot for slot.
template <class P>
JLQ int tab_try_upsert(TabT<P>& t, uint32_t skey, uint32_t payload, uint32_t h32, uint64_t* want) {
    const uint32_t st = t.stride;
    const uint32_t sz = t.sz, mask = sz - 1;
    uint32_t idx = h32 & mask, it = 0;
    bool found_empty = false;
    for (;;) {
        if (!t.pay[idx * st]) { found_empty = true; break; }
        if (t.key[idx * st] == skey) { t.pay[idx * st] = payload + 1; return 0; }
        idx = (idx + 1) & mask;
        if (++it > t.maxprobe) break;
    }
    if (!found_empty) {
        const uint32_t lim = (sz >> 6) > 16 ? (sz >> 6) : 16;
        while (it < lim) {
            if (!t.pay[idx * st]) { found_empty = true; t.maxprobe = it; break; }
            idx = (idx + 1) & mask;
            ++it;
        }
    }
    if (!found_empty) { *want = t.n > 64000 ? (uint64_t)sz * 2 : (uint64_t)sz * 4; return 1; }
    t.key[idx * st] = skey;
    t.pay[idx * st] = payload + 1;
    ++t.n;
    if ((uint64_t)t.n * 3 > (uint64_t)sz * 2) { *want = t.n > 64000 ? (uint64_t)t.n * 2 : (uint64_t)t.n * 4; return 2; }
    return 0;
}

// The same once more on ONE array of 64-bit slots, payload + 1 in the high word (0 = empty), key in the low word: a probe is one load instead of
// two dependent ones (frontend.hip.hpp, fe_wave_table: lane 0's probe walk is what is left of a long part's table).
template <class P64>
struct Tab64T {
    P64 cur, nxt;        // the table and the buffer a growth re-inserts into
    uint32_t cap;        // slots per buffer
    uint32_t sz, n, maxprobe;
};
// UNI = true (device, frontend.hip.hpp): the walk is executed by EVERY lane of the wavefront on the same values -- each LDS read is made
// wave-uniform (readfirstlane) so that the compiler keeps the walk's arithmetic and branches on the scalar unit. One lane walking by itself
// is slower still (650 cycles per insertion against 600 this way: what is left is two dependent LDS round trips per insertion -- the staged key,
// then its slot); the stores are the same value to the same address from all lanes.
template <bool UNI>
JLQ uint64_t jl_uni64(uint64_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (UNI) return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(x >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)x);
#endif
    return x;
}
// (Tried, round 6: reading FOUR slots per probe step -- independent loads, one latency -- 0.60 -> 0.83 ms for the 208 long parts of
//  ecdsa_like(26), lane-0 and wave-uniform versions alike: most insertions find their slot at the first probe, the extra reads only cost.)
template <bool UNI, class P64>
JLQ int tab_try_upsert64(Tab64T<P64>& t, uint32_t skey, uint32_t payload, uint32_t h32, uint64_t* want) {
    const uint32_t sz = t.sz, mask = sz - 1;
    uint32_t idx = h32 & mask, it = 0;
    bool found_empty = false;
    const uint64_t mine = ((uint64_t)(payload + 1u) << 32) | (uint64_t)skey;
    for (;;) {
        const uint64_t s = jl_uni64<UNI>(t.cur[idx]);
        if (!(uint32_t)(s >> 32)) { found_empty = true; break; }
        if ((uint32_t)s == skey) { t.cur[idx] = mine; return 0; }
        idx = (idx + 1) & mask;
        if (++it > t.maxprobe) break;
    }
    if (!found_empty) {
        const uint32_t lim = (sz >> 6) > 16 ? (sz >> 6) : 16;
        while (it < lim) {
            if (!(uint32_t)(jl_uni64<UNI>(t.cur[idx]) >> 32)) { found_empty = true; t.maxprobe = it; break; }
            idx = (idx + 1) & mask;
            ++it;
        }
    }
    if (!found_empty) { *want = t.n > 64000 ? (uint64_t)sz * 2 : (uint64_t)sz * 4; return 1; }
    t.cur[idx] = mine;
    ++t.n;
    if ((uint64_t)t.n * 3 > (uint64_t)sz * 2) { *want = t.n > 64000 ? (uint64_t)t.n * 2 : (uint64_t)t.n * 4; return 2; }
    return 0;
}
// the first free slot at or behind `home` in the buffer a growth fills (tab_grow's walk)
template <bool UNI, class P64>
JLQ uint32_t tab_free_slot64(P64 tab, uint32_t mask, uint32_t home) {
    uint32_t idx = home;
    while ((uint32_t)(jl_uni64<UNI>(tab[idx]) >> 32)) idx = (idx + 1) & mask;
    return idx;
}

// Which of two DIFFERENT keys comes first when a fresh table that holds just the two is iterated (`for j in Set([k1, k2])`,
// R1CSConstraintSolver.jl:1130, :1216): 16 slots, k1 inserted first.
JLQ bool pair_second_first(uint32_t k1, uint32_t k2) {
    const uint32_t h1 = (uint32_t)(hash64((uint64_t)k1) & 15u);
    uint32_t h2 = (uint32_t)(hash64((uint64_t)k2) & 15u);
    if (h2 == h1) h2 = (h2 + 1) & 15u;     // k2 probes on; slot 15 wraps to 0
    return h2 < h1;
}

}  // namespace jlslot
