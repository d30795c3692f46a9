// oracle/u256.hpp — TEST INFRASTRUCTURE ONLY (parity oracle). Not part of the shipped product.
//
// 256-bit unsigned integers and BN254-scalar-field arithmetic for the CPU oracle.
// Restates the arithmetic the reference obtains from AbstractAlgebra.GF(bjj_p) on BigInt
// (reference: src/R1CSConstraintSolver.jl:21-24; AbstractAlgebra 0.23.0 is an un-vendored
// dependency, Manifest.toml:3-7 — field arithmetic on canonical residues is mathematically
// determined, so it is restated from number theory, and pinned by tests/test_field.py against
// Python big integers).
//
// Deliberately a different formulation from the product's device code (csrc/fp256.hpp uses
// CIOS Montgomery on 64-bit mul-hi/lo): here products are formed with unsigned __int128 and
// reduced with a separate REDC pass; inversion is the binary extended Euclid.
#pragma once
#include <cstdint>
#include <cstring>
#include <string>
#include <algorithm>

namespace orc {

typedef unsigned __int128 u128;

struct U256 {
    uint64_t w[4];
    U256() { w[0] = w[1] = w[2] = w[3] = 0; }
    explicit U256(uint64_t x) { w[0] = x; w[1] = w[2] = w[3] = 0; }
    U256(uint64_t a, uint64_t b, uint64_t c, uint64_t d) { w[0] = a; w[1] = b; w[2] = c; w[3] = d; }
    bool is_zero() const { return (w[0] | w[1] | w[2] | w[3]) == 0; }
    bool is_one() const { return w[0] == 1 && (w[1] | w[2] | w[3]) == 0; }
    bool bit(int i) const { return (w[i >> 6] >> (i & 63)) & 1; }
    int bitlen() const {
        for (int i = 3; i >= 0; --i)
            if (w[i]) return 64 * i + (64 - __builtin_clzll(w[i]));
        return 0;
    }
};

inline int cmp(const U256& a, const U256& b) {
    for (int i = 3; i >= 0; --i) {
        if (a.w[i] < b.w[i]) return -1;
        if (a.w[i] > b.w[i]) return 1;
    }
    return 0;
}
inline bool operator==(const U256& a, const U256& b) { return cmp(a, b) == 0; }
inline bool operator!=(const U256& a, const U256& b) { return cmp(a, b) != 0; }
inline bool operator<(const U256& a, const U256& b) { return cmp(a, b) < 0; }
inline bool operator>(const U256& a, const U256& b) { return cmp(a, b) > 0; }
inline bool operator<=(const U256& a, const U256& b) { return cmp(a, b) <= 0; }
inline bool operator>=(const U256& a, const U256& b) { return cmp(a, b) >= 0; }

// r = a + b, returns carry
inline uint64_t add_c(U256& r, const U256& a, const U256& b) {
    u128 c = 0;
    for (int i = 0; i < 4; ++i) {
        c += (u128)a.w[i] + b.w[i];
        r.w[i] = (uint64_t)c;
        c >>= 64;
    }
    return (uint64_t)c;
}
// r = a - b, returns borrow
inline uint64_t sub_b(U256& r, const U256& a, const U256& b) {
    uint64_t br = 0;
    for (int i = 0; i < 4; ++i) {
        u128 d = (u128)a.w[i] - b.w[i] - br;
        r.w[i] = (uint64_t)d;
        br = (uint64_t)(d >> 64) & 1;
    }
    return br;
}
inline U256 shr1(const U256& a) {
    U256 r;
    for (int i = 0; i < 4; ++i) r.w[i] = (a.w[i] >> 1) | (i < 3 ? a.w[i + 1] << 63 : 0);
    return r;
}
inline U256 shl1(const U256& a, uint64_t* out = nullptr) {
    U256 r;
    for (int i = 3; i >= 0; --i) r.w[i] = (a.w[i] << 1) | (i > 0 ? a.w[i - 1] >> 63 : 0);
    if (out) *out = a.w[3] >> 63;
    return r;
}

struct U512 {
    uint64_t w[8];
};
inline U512 mul_wide(const U256& a, const U256& b) {
    U512 r;
    std::memset(r.w, 0, sizeof r.w);
    for (int i = 0; i < 4; ++i) {
        u128 carry = 0;
        for (int j = 0; j < 4; ++j) {
            u128 t = (u128)a.w[i] * b.w[j] + r.w[i + j] + carry;
            r.w[i + j] = (uint64_t)t;
            carry = t >> 64;
        }
        r.w[i + 4] = (uint64_t)carry;
    }
    return r;
}
// compare a 512-bit value with a 256-bit value
inline int cmp512_256(const U512& a, const U256& b) {
    for (int i = 7; i >= 4; --i)
        if (a.w[i]) return 1;
    for (int i = 3; i >= 0; --i) {
        if (a.w[i] < b.w[i]) return -1;
        if (a.w[i] > b.w[i]) return 1;
    }
    return 0;
}

// integer division a = q*b + r (b != 0); schoolbook shift-subtract (only used by the rare
// mixed-radix rule R7, reference :1267-1268)
inline void divmod(const U256& a, const U256& b, U256& q, U256& r) {
    q = U256();
    r = U256();
    int n = a.bitlen();
    for (int i = n - 1; i >= 0; --i) {
        uint64_t out;
        r = shl1(r, &out);
        if (a.bit(i)) r.w[0] |= 1;
        if (out || cmp(r, b) >= 0) {
            U256 t;
            sub_b(t, r, b);
            r = t;
            q.w[i >> 6] |= (uint64_t)1 << (i & 63);
        }
    }
}

// ---- the field: BN254 scalar prime (reference :21-22) ----
static const U256 P(0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL,
                    0x30644e72e131a029ULL);
static const U256 R2MODP(0x1bb8e645ae216da7ULL, 0x53fe3ab1e35c59e3ULL, 0x8c49833d53bb8085ULL,
                         0x0216d0b17f4e44a5ULL);
static const uint64_t N0INV = 0xc2e1f593efffffffULL;  // -p^{-1} mod 2^64

inline U256 fp_add(const U256& a, const U256& b) {
    U256 r;
    uint64_t c = add_c(r, a, b);
    if (c || cmp(r, P) >= 0) {
        U256 t;
        sub_b(t, r, P);
        return t;
    }
    return r;
}
inline U256 fp_sub(const U256& a, const U256& b) {
    U256 r;
    if (sub_b(r, a, b)) {
        U256 t;
        add_c(t, r, P);
        return t;
    }
    return r;
}
inline U256 fp_neg(const U256& a) {
    if (a.is_zero()) return a;
    U256 r;
    sub_b(r, P, a);
    return r;
}
// REDC of a 512-bit T < p*2^256 : T * 2^-256 mod p
inline U256 redc(const U512& T) {
    uint64_t t[9];
    for (int i = 0; i < 8; ++i) t[i] = T.w[i];
    t[8] = 0;
    for (int i = 0; i < 4; ++i) {
        uint64_t m = t[i] * N0INV;
        u128 carry = 0;
        for (int j = 0; j < 4; ++j) {
            u128 s = (u128)m * P.w[j] + t[i + j] + carry;
            t[i + j] = (uint64_t)s;
            carry = s >> 64;
        }
        for (int k = i + 4; k < 9 && carry; ++k) {
            u128 s = (u128)t[k] + carry;
            t[k] = (uint64_t)s;
            carry = s >> 64;
        }
    }
    U256 r(t[4], t[5], t[6], t[7]);
    if (t[8] || cmp(r, P) >= 0) {
        U256 u;
        sub_b(u, r, P);
        return u;
    }
    return r;
}
// canonical * canonical -> canonical
inline U256 fp_mul(const U256& a, const U256& b) {
    U256 x = redc(mul_wide(a, b));       // a*b/R
    return redc(mul_wide(x, R2MODP));    // a*b
}
// reduce an arbitrary 256-bit integer mod p (F(coeff) in ParseR1CS.jl:111)
inline U256 fp_reduce(const U256& a) {
    U256 r = a;
    while (cmp(r, P) >= 0) {
        U256 t;
        sub_b(t, r, P);
        r = t;
    }
    return r;
}
struct DivideError {};
// binary extended Euclid; a in [1,p)
inline U256 fp_inv(const U256& a) {
    if (a.is_zero()) throw DivideError();
    U256 u = a, v = P, x1(1), x2(0);
    while (!u.is_one() && !v.is_one()) {
        while (!(u.w[0] & 1)) {
            u = shr1(u);
            if (x1.w[0] & 1) {
                U256 t;
                uint64_t c = add_c(t, x1, P);
                x1 = shr1(t);
                if (c) x1.w[3] |= (uint64_t)1 << 63;
            } else
                x1 = shr1(x1);
        }
        while (!(v.w[0] & 1)) {
            v = shr1(v);
            if (x2.w[0] & 1) {
                U256 t;
                uint64_t c = add_c(t, x2, P);
                x2 = shr1(t);
                if (c) x2.w[3] |= (uint64_t)1 << 63;
            } else
                x2 = shr1(x2);
        }
        if (cmp(u, v) >= 0) {
            U256 t;
            sub_b(t, u, v);
            u = t;
            x1 = fp_sub(x1, x2);
        } else {
            U256 t;
            sub_b(t, v, u);
            v = t;
            x2 = fp_sub(x2, x1);
        }
    }
    return u.is_one() ? x1 : x2;
}
// AbstractAlgebra.divexact(a, b) on GF(p): a * b^-1, DivideError when b == 0
inline U256 fp_div(const U256& a, const U256& b) { return fp_mul(a, fp_inv(b)); }

}  // namespace orc
