// CPU check of csrc/jlslot.hpp (the slot-order model the device front-end runs) against csrc/jlorder.hpp's jl::SlotTable
// (the host model, itself pinned to the reference's dumps by tests/test_julia_order.py). Built and run by tests/test_jlslot_host.py.
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "../../ecneproject_amd/csrc/jlorder.hpp"
#include "../../ecneproject_amd/csrc/jlslot.hpp"

template <int ADD>
static int one(std::mt19937_64& rng, size_t n, uint64_t keyspace, uint32_t stride) {
    std::vector<uint32_t> keys(n);
    for (auto& k : keys) k = (uint32_t)(rng() % keyspace) + (keyspace > 1000000 ? 0xFFFFFF00u : 0u) * (rng() % 7 == 0);
    uint32_t cap = 16;
    while (cap < 8 * n + 64) cap <<= 1;
    std::vector<uint32_t> a((size_t)cap * stride), b((size_t)cap * stride), c((size_t)cap * stride), d((size_t)cap * stride);
    jlslot::Tab t{a.data(), b.data(), c.data(), d.data(), cap, stride, 0, 0, 0};
    jlslot::tab_init(t);
    jl::SlotTable ref;
    for (size_t i = 0; i < n; ++i) {
        bool ins;
        int64_t& slot = ref.upsert((int64_t)keys[i] + ADD, (int64_t)i, ins);
        if (!ins) slot = (int64_t)i;
        if (jlslot::tab_upsert<ADD>(t, keys[i], (uint32_t)i)) { std::printf("capacity\n"); return 1; }
    }
    std::vector<std::pair<int64_t, int64_t>> want, got;
    ref.for_each([&](int64_t k, int64_t p) { want.push_back({k, p}); });
    for (uint32_t s = 0; s < t.sz; ++s)
        if (t.pay[s * stride]) got.push_back({(int64_t)t.key[s * stride] + ADD, (int64_t)t.pay[s * stride] - 1});
    if (want != got) { std::printf("MISMATCH n=%zu keyspace=%llu add=%d\n", n, (unsigned long long)keyspace, ADD); return 1; }
    return 0;
}

// tab_try_upsert + a growth that re-inserts from a LIST of the old slots (in ascending slot order, hashes precomputed) -- the way the device
// front-end's wavefront builder does it (frontend.hip.hpp, fe_wave_table) -- against tab_upsert: the same table, slot for slot, buffers and all.
template <int ADD>
static int two(std::mt19937_64& rng, size_t n, uint64_t keyspace) {
    std::vector<uint32_t> keys(n);
    for (auto& k : keys) k = (uint32_t)(rng() % keyspace);
    uint32_t cap = 16;
    while (cap < 8 * n + 64) cap <<= 1;
    std::vector<uint32_t> a(cap), b(cap), c(cap), d(cap), a2(cap), b2(cap), c2(cap), d2(cap);
    jlslot::Tab t{a.data(), b.data(), c.data(), d.data(), cap, 1, 0, 0, 0};
    jlslot::tab_init(t);
    for (size_t i = 0; i < n; ++i) if (jlslot::tab_upsert<ADD>(t, keys[i], (uint32_t)i)) return 1;
    jlslot::Tab u{a2.data(), b2.data(), c2.data(), d2.data(), cap, 1, 0, 0, 0};
    jlslot::tab_init(u);
    std::vector<uint32_t> lk, lp, lh;
    size_t k = 0;
    while (k < n) {
        uint64_t want = 0;
        const int rc = jlslot::tab_try_upsert(u, keys[k], (uint32_t)k, (uint32_t)jlslot::hash64((uint64_t)keys[k] + ADD), &want);
        if (rc != 1) ++k;
        if (!rc) continue;
        uint32_t nsz = 16;
        while (nsz < want) nsz <<= 1;
        if (nsz > cap) return 1;
        for (uint32_t i = 0; i < nsz; ++i) u.pay2[i] = 0;
        lk.clear(); lp.clear(); lh.clear();
        for (uint32_t sl = 0; sl < u.sz; ++sl) if (u.pay[sl]) { lk.push_back(u.key[sl]); lp.push_back(u.pay[sl]); lh.push_back((uint32_t)jlslot::hash64((uint64_t)u.key[sl] + ADD)); }
        uint32_t mp = 0;
        const uint32_t mask = nsz - 1;
        for (size_t i = 0; i < lk.size(); ++i) {
            const uint32_t home = lh[i] & mask;
            uint32_t idx = home;
            while (u.pay2[idx]) idx = (idx + 1) & mask;
            const uint32_t probe = (idx - home) & mask;
            if (probe > mp) mp = probe;
            u.pay2[idx] = lp[i]; u.key2[idx] = lk[i];
        }
        std::swap(u.key, u.key2); std::swap(u.pay, u.pay2);
        u.sz = nsz; u.maxprobe = mp;
    }
    if (t.sz != u.sz || t.n != u.n || t.maxprobe != u.maxprobe) { std::printf("TRY MISMATCH shape n=%zu\n", n); return 1; }
    for (uint32_t sl = 0; sl < t.sz; ++sl)
        if (t.pay[sl] != u.pay[sl] || (t.pay[sl] && t.key[sl] != u.key[sl])) { std::printf("TRY MISMATCH slot %u n=%zu\n", sl, n); return 1; }
    return 0;
}

// ... and the 64-bit-slot variant (tab_try_upsert64) the same way
template <int ADD>
static int three(std::mt19937_64& rng, size_t n, uint64_t keyspace) {
    std::vector<uint32_t> keys(n);
    for (auto& k : keys) k = (uint32_t)(rng() % keyspace);
    uint32_t cap = 16;
    while (cap < 8 * n + 64) cap <<= 1;
    std::vector<uint32_t> a(cap), b(cap), c(cap), d(cap);
    jlslot::Tab t{a.data(), b.data(), c.data(), d.data(), cap, 1, 0, 0, 0};
    jlslot::tab_init(t);
    for (size_t i = 0; i < n; ++i) if (jlslot::tab_upsert<ADD>(t, keys[i], (uint32_t)i)) return 1;
    std::vector<uint64_t> x(cap, 0), y(cap, 0);
    jlslot::Tab64T<uint64_t*> u{x.data(), y.data(), cap, 16, 0, 0};
    std::vector<uint32_t> lk, lp, lh;
    size_t k = 0;
    while (k < n) {
        uint64_t want = 0;
        const int rc = jlslot::tab_try_upsert64<false>(u, keys[k], (uint32_t)k, (uint32_t)jlslot::hash64((uint64_t)keys[k] + ADD), &want);
        if (rc != 1) ++k;
        if (!rc) continue;
        uint32_t nsz = 16;
        while (nsz < want) nsz <<= 1;
        if (nsz > cap) return 1;
        for (uint32_t i = 0; i < nsz; ++i) u.nxt[i] = 0;
        lk.clear(); lp.clear(); lh.clear();
        for (uint32_t sl = 0; sl < u.sz; ++sl) if (u.cur[sl] >> 32) { lk.push_back((uint32_t)u.cur[sl]); lp.push_back((uint32_t)(u.cur[sl] >> 32)); lh.push_back((uint32_t)jlslot::hash64((uint64_t)(uint32_t)u.cur[sl] + ADD)); }
        uint32_t mp = 0;
        const uint32_t mask = nsz - 1;
        for (size_t i = 0; i < lk.size(); ++i) {
            const uint32_t home = lh[i] & mask;
            const uint32_t idx = jlslot::tab_free_slot64<false>(u.nxt, mask, home);
            const uint32_t probe = (idx - home) & mask;
            if (probe > mp) mp = probe;
            u.nxt[idx] = ((uint64_t)lp[i] << 32) | lk[i];
        }
        std::swap(u.cur, u.nxt);
        u.sz = nsz; u.maxprobe = mp;
    }
    if (t.sz != u.sz || t.n != u.n || t.maxprobe != u.maxprobe) { std::printf("TRY64 MISMATCH shape n=%zu\n", n); return 1; }
    for (uint32_t sl = 0; sl < t.sz; ++sl) {
        const uint32_t py = (uint32_t)(u.cur[sl] >> 32), ky = (uint32_t)u.cur[sl];
        if (t.pay[sl] != py || (py && t.key[sl] != ky)) { std::printf("TRY64 MISMATCH slot %u n=%zu\n", sl, n); return 1; }
    }
    return 0;
}

int main() {
    std::mt19937_64 rng(12345);
    int bad = 0;
    size_t cases = 0;
    for (int rep = 0; rep < 4000; ++rep) {
        const size_t n = 1 + rng() % (rep % 50 == 0 ? 3000 : rep % 7 == 0 ? 200 : 24);
        const uint64_t ks = rep % 3 == 0 ? 16 + rng() % 64 : rep % 3 == 1 ? 1 + rng() % 100000 : 4000000000ull;
        bad += one<0>(rng, n, ks, 1 + rep % 3);
        bad += one<1>(rng, n, ks, 1 + rep % 3);
        cases += 2;
    }
    for (int rep = 0; rep < 3000; ++rep) {
        const size_t n = 1 + rng() % (rep % 20 == 0 ? 3000 : rep % 5 == 0 ? 200 : 40);
        const uint64_t ks = rep % 3 == 0 ? 16 + rng() % 64 : rep % 3 == 1 ? 1 + rng() % 100000 : 4000000000ull;
        bad += two<0>(rng, n, ks);
        bad += two<1>(rng, n, ks);
        bad += three<0>(rng, n, ks);
        bad += three<1>(rng, n, ks);
        cases += 4;
    }
    for (size_t n : {64001u, 70000u, 130000u}) { bad += one<0>(rng, n, 1u << 30, 1); bad += one<1>(rng, n, 50000, 1); cases += 2; }
    // the two-key shortcut
    for (int rep = 0; rep < 200000; ++rep) {
        const uint32_t k1 = (uint32_t)(rng() % (rep % 2 ? 64 : 2000000)) + 1, k2 = (uint32_t)(rng() % (rep % 2 ? 64 : 2000000)) + 1;
        if (k1 == k2) continue;
        jl::SlotTable s;
        bool ins;
        s.upsert(k1, 0, ins);
        s.upsert(k2, 1, ins);
        int64_t first = -1;
        s.for_each([&](int64_t k, int64_t) { if (first < 0) first = k; });
        if (((uint32_t)first == k2) != jlslot::pair_second_first(k1, k2)) { std::printf("PAIR MISMATCH %u %u\n", k1, k2); ++bad; }
    }
    std::printf("%zu sequences, %d mismatches\n", cases, bad);
    return bad ? 1 : 0;
}
