// jlorder.hpp — host-side model of the slot order of Julia 1.7 hash tables keyed by Int64.
//
// Why the product needs this at all: the reference walks `Set{Any}` / `Dict` objects in slot
// order when it re-enqueues the rows of several variables (R1CSConstraintSolver.jl:1050, :1130,
// :1216, :1285, :1328) and when abstraction() breaks ties (:334-335), and that order is visible
// in the FIFO queue and therefore, in principle, in the fixed point.
// The engine never hashes on the device: at load time the host computes, once per row, the
// order in which the reference would visit that row's variables and lays the CSR entries out in
// exactly that order.  This file is that load-time computation.  It is written independently of
// the test oracle (oracle/jldict.hpp); both are checked against the same known-answer vectors.
//
// Model (Julia 1.7 base/dict.jl): open addressing, linear probing, table size a power of two
// starting at 16, home slot = hash_64_64(key) & (size-1), grow x4 (x2 past 64 000 keys) when
// more than 2/3 full or when a probe sequence would exceed max(16, size/64); a grow re-inserts
// the old slots in ascending slot order; iteration = ascending slot order; no deletions occur.
#pragma once
#include <cstdint>
#include <vector>

namespace jl {

inline uint64_t hash64(uint64_t a) {
    a = ~a + (a << 21);
    a ^= a >> 24;
    a += (a << 3) + (a << 8);
    a ^= a >> 14;
    a += (a << 2) + (a << 4);
    a ^= a >> 28;
    a += a << 31;
    return a;
}

// A table that only remembers which key sits in which slot plus a caller-defined payload index.
class SlotTable {
  public:
    SlotTable() { reset(); }
    void reset() {
        if (key_.size() != 16) {
            key_.assign(16, 0);
            pay_.assign(16, 0);
            used_.assign(16, 0);
        } else {
            for (int i = 0; i < 16; ++i) used_[i] = 0;
        }
        n_ = 0;
        maxprobe_ = 0;
    }
    int64_t count() const { return n_; }
    // returns the payload slot of `key`, inserting it with payload `pay` when absent;
    // `inserted` tells which happened
    int64_t& upsert(int64_t key, int64_t pay, bool& inserted) {
        for (;;) {
            const int64_t sz = (int64_t)key_.size(), mask = sz - 1;
            int64_t idx = (int64_t)(hash64((uint64_t)key) & (uint64_t)mask), it = 0;
            bool found_empty = false;
            for (;;) {
                if (!used_[idx]) { found_empty = true; break; }
                if (key_[idx] == key) { inserted = false; return pay_[idx]; }
                idx = (idx + 1) & mask;
                if (++it > maxprobe_) break;
            }
            if (!found_empty) {
                const int64_t lim = sz >> 6 > 16 ? sz >> 6 : 16;
                while (it < lim) {
                    if (!used_[idx]) { found_empty = true; maxprobe_ = it; break; }
                    idx = (idx + 1) & mask;
                    ++it;
                }
            }
            if (!found_empty) {
                grow(n_ > 64000 ? sz * 2 : sz * 4);
                continue;
            }
            used_[idx] = 1;
            key_[idx] = key;
            pay_[idx] = pay;
            ++n_;
            inserted = true;
            if (n_ * 3 > sz * 2) {
                grow(n_ > 64000 ? n_ * 2 : n_ * 4);
                return pay_[locate(key)];
            }
            return pay_[idx];
        }
    }
    // sizehint!(d, n) (base/dict.jl): room for n keys up front -- what union!(Set(), itr) does before it pushes,
    // and it changes the slot order compared with growing step by step
    void reserve(int64_t n) {
        if (n < n_) n = n_;
        const int64_t want = (3 * n + 1) / 2;
        int64_t nsz = 16;
        while (nsz < want) nsz <<= 1;
        if (nsz > (int64_t)key_.size()) grow(nsz);      // (sizehint! never shrinks)
    }
    bool contains(int64_t key) const { return locate(key) >= 0; }
    int64_t payload_of(int64_t key) const {
        int64_t i = locate(key);
        return i < 0 ? -1 : pay_[i];
    }
    // visit (key, payload) in iteration order
    template <class F>
    void for_each(F f) const {
        for (size_t i = 0; i < key_.size(); ++i)
            if (used_[i]) f(key_[i], pay_[i]);
    }

  private:
    int64_t locate(int64_t key) const {
        const int64_t sz = (int64_t)key_.size(), mask = sz - 1;
        int64_t idx = (int64_t)(hash64((uint64_t)key) & (uint64_t)mask), it = 0;
        for (;;) {
            if (!used_[idx]) return -1;
            if (key_[idx] == key) return idx;
            idx = (idx + 1) & mask;
            if (++it > maxprobe_) return -1;
        }
    }
    void grow(int64_t want) {
        int64_t nsz = 16;
        while (nsz < want) nsz <<= 1;
        std::vector<int64_t> k(nsz, 0), p(nsz, 0);
        std::vector<uint8_t> u(nsz, 0);
        int64_t mp = 0;
        const int64_t mask = nsz - 1;
        for (size_t i = 0; i < key_.size(); ++i) {
            if (!used_[i]) continue;
            int64_t home = (int64_t)(hash64((uint64_t)key_[i]) & (uint64_t)mask), idx = home;
            while (u[idx]) idx = (idx + 1) & mask;
            int64_t probe = (idx - home) & mask;
            if (probe > mp) mp = probe;
            u[idx] = 1;
            k[idx] = key_[i];
            p[idx] = pay_[i];
        }
        key_.swap(k);
        pay_.swap(p);
        used_.swap(u);
        maxprobe_ = mp;
    }
    std::vector<int64_t> key_, pay_;
    std::vector<uint8_t> used_;
    int64_t n_ = 0, maxprobe_ = 0;
};

}  // namespace jl
