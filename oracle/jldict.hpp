// oracle/jldict.hpp — TEST INFRASTRUCTURE ONLY (parity oracle). Not part of the shipped product.
//
// Emulation of Julia 1.7.2 Base.Dict / Set{Any} / DataStructures.DefaultDict *iteration order*
// for Int64 keys. The reference iterates such tables at
//   src/R1CSConstraintSolver.jl:26-56 (nonzeroKeys / getVariables), :1005, :1020, :1050, :1087,
//   :1130, :1158, :1165, :1216, :1237, :1285, :1307, :1328, :334-335 (abstraction tie-breaks),
// and the order is observable through the FIFO queue.
// Julia Base is not under /root/reference (Manifest: julia_version = "1.7.2"); the table
// algorithm below is restated from Julia 1.7 base/dict.jl (hash_64_64, hashindex,
// ht_keyindex2!, _setindex!, rehash!) and is pinned by the known-answer test
// tests/test_julia_order.py against the reference's own equation dumps
// Circom_Functions/benchmarks/*.txt and README.md:103-105.
#pragma once
#include <cstdint>
#include <vector>
#include <algorithm>
#include <random>

namespace orc {

inline uint64_t jl_hash_64_64(uint64_t n) {
    uint64_t a = n;
    a = ~a + (a << 21);
    a = a ^ (a >> 24);
    a = a + (a << 3) + (a << 8);
    a = a ^ (a >> 14);
    a = a + (a << 2) + (a << 4);
    a = a ^ (a >> 28);
    a = a + (a << 31);
    return a;
}

// Iteration-order policy: JULIA is *the reference result*;
// ASCENDING and RANDOM exist only for the schedule-invariance property test.
enum OrderPolicy { ORDER_JULIA = 0, ORDER_ASCENDING = 1, ORDER_RANDOM = 2 };
struct OrderCtx {
    int policy = ORDER_JULIA;
    std::mt19937_64 rng{12345};
};

template <class V>
class JlDict {
  public:
    std::vector<uint8_t> slots;
    std::vector<int64_t> keys;
    std::vector<V> vals;
    int64_t count = 0;
    int64_t maxprobe = 0;

    JlDict() : slots(16, 0), keys(16, 0), vals(16) {}

    int64_t size() const { return count; }

    static int64_t tablesz(int64_t x) {
        if (x < 16) return 16;
        int64_t s = 1;
        while (s < x) s <<= 1;
        return s;
    }
    int64_t hashindex(int64_t key, int64_t sz) const {
        return (int64_t)(jl_hash_64_64((uint64_t)key) & (uint64_t)(sz - 1));  // 0-based
    }

    void rehash(int64_t newsz_req) {
        int64_t sz = (int64_t)slots.size();
        int64_t newsz = tablesz(newsz_req);
        if (count == 0) {
            slots.assign(newsz, 0);
            keys.assign(newsz, 0);
            vals.assign(newsz, V());
            return;
        }
        std::vector<uint8_t> ns(newsz, 0);
        std::vector<int64_t> nk(newsz, 0);
        std::vector<V> nv(newsz);
        int64_t mp = 0;
        for (int64_t i = 0; i < sz; ++i) {
            if (slots[i]) {
                int64_t index0 = hashindex(keys[i], newsz), index = index0;
                while (ns[index]) index = (index + 1) & (newsz - 1);
                int64_t probe = (index - index0) & (newsz - 1);
                if (probe > mp) mp = probe;
                ns[index] = 1;
                nk[index] = keys[i];
                nv[index] = std::move(vals[i]);
            }
        }
        slots.swap(ns);
        keys.swap(nk);
        vals.swap(nv);
        maxprobe = mp;
    }

    // >=0: slot of an existing key; <0: -(slot+1) where the key would be inserted
    int64_t keyindex2(int64_t key) {
        for (;;) {
            int64_t sz = (int64_t)keys.size();
            int64_t iter = 0;
            int64_t index = hashindex(key, sz);
            bool broke = false;
            for (;;) {
                if (!slots[index]) return -(index + 1);
                if (keys[index] == key) return index;
                index = (index + 1) & (sz - 1);
                iter += 1;
                if (iter > maxprobe) { broke = true; break; }
            }
            (void)broke;
            int64_t maxallowed = std::max<int64_t>(16, sz >> 6);
            while (iter < maxallowed) {
                if (!slots[index]) {
                    maxprobe = iter;
                    return -(index + 1);
                }
                index = (index + 1) & (sz - 1);
                iter += 1;
            }
            rehash(count > 64000 ? sz * 2 : sz * 4);
        }
    }

    int64_t find(int64_t key) const {
        int64_t sz = (int64_t)keys.size();
        int64_t iter = 0;
        int64_t index = hashindex(key, sz);
        for (;;) {
            if (!slots[index]) return -1;
            if (keys[index] == key) return index;
            index = (index + 1) & (sz - 1);
            iter += 1;
            if (iter > maxprobe) return -1;
        }
    }
    bool contains(int64_t key) const { return find(key) >= 0; }

    // setindex!(h, v, key)
    void set(int64_t key, const V& v) {
        int64_t idx = keyindex2(key);
        if (idx >= 0) {
            keys[idx] = key;
            vals[idx] = v;
        } else {
            insert_at(-idx - 1, key, v);
        }
    }
    // get!(h, key, default)
    V& get_or_insert(int64_t key, const V& dflt) {
        int64_t idx = keyindex2(key);
        if (idx >= 0) return vals[idx];
        insert_at(-idx - 1, key, dflt);
        return vals[find(key)];
    }
    const V* get(int64_t key) const {
        int64_t i = find(key);
        return i < 0 ? nullptr : &vals[i];
    }

    // slot indices of the filled slots in Julia iteration order (ascending slot), or in the
    // order an invariance-test policy prescribes
    std::vector<int64_t> order(OrderCtx* ctx = nullptr) const {
        std::vector<int64_t> o;
        o.reserve((size_t)count);
        for (int64_t i = 0; i < (int64_t)slots.size(); ++i)
            if (slots[i]) o.push_back(i);
        if (ctx && ctx->policy == ORDER_ASCENDING) {
            std::sort(o.begin(), o.end(), [&](int64_t x, int64_t y) { return keys[x] < keys[y]; });
        } else if (ctx && ctx->policy == ORDER_RANDOM) {
            std::sort(o.begin(), o.end(), [&](int64_t x, int64_t y) { return keys[x] < keys[y]; });
            std::shuffle(o.begin(), o.end(), ctx->rng);
        }
        return o;
    }
    std::vector<int64_t> ordered_keys(OrderCtx* ctx = nullptr) const {
        std::vector<int64_t> o = order(ctx);
        for (auto& s : o) s = keys[s];
        return o;
    }

  private:
    void insert_at(int64_t index, int64_t key, const V& v) {
        slots[index] = 1;
        keys[index] = key;
        vals[index] = v;
        count += 1;
        int64_t sz = (int64_t)keys.size();
        if (count * 3 > sz * 2) rehash(count > 64000 ? count * 2 : count * 4);
    }
};

struct Nothing {};
typedef JlDict<Nothing> JlSet;

}  // namespace orc
