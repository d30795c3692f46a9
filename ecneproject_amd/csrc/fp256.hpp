// fp256.hpp — BN254 scalar field on 4x64-bit limbs, usable from host and gfx950 device code.
//
// Replaces what the reference gets from AbstractAlgebra.GF(bjj_p) over BigInt
// (/root/reference/src/R1CSConstraintSolver.jl:21-24; live call sites: divexact :919-920,
// :961-964, :1467; unary minus :919-920, :962, :1006; F(2)^i :999-1000, :1033; * and + :1395-1397;
// canonical-integer compares through `.d` :1035, :1113-1116, :1194-1196, :1257-1274).
//
// Storage format everywhere (CSR coefficients, lb/ub/values state) is the CANONICAL residue in
// [0,p) as 4 little-endian u64 limbs, because the reference's bound logic compares canonical
// integers. Montgomery form exists only in registers inside mul/inv.  No MFMA: this is integer
// modular arithmetic (64x64->128 via mul_lo/mul_hi), not a dense contraction.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define FPQ __host__ __device__ __forceinline__
#else
#define FPQ inline
#endif

namespace fp {

struct u256 {
    uint64_t w[4];
};

FPQ u256 make(uint64_t a, uint64_t b = 0, uint64_t c = 0, uint64_t d = 0) {
    u256 r;
    r.w[0] = a; r.w[1] = b; r.w[2] = c; r.w[3] = d;
    return r;
}
// p, R^2 mod p (R = 2^256) and -p^-1 mod 2^64
#define FP_P0 0x43e1f593f0000001ULL
#define FP_P1 0x2833e84879b97091ULL
#define FP_P2 0xb85045b68181585dULL
#define FP_P3 0x30644e72e131a029ULL
#define FP_N0INV 0xc2e1f593efffffffULL
FPQ u256 modulus() { return make(FP_P0, FP_P1, FP_P2, FP_P3); }
FPQ u256 r2modp() {
    return make(0x1bb8e645ae216da7ULL, 0x53fe3ab1e35c59e3ULL, 0x8c49833d53bb8085ULL, 0x0216d0b17f4e44a5ULL);
}
FPQ u256 pminus1() { return make(FP_P0 - 1, FP_P1, FP_P2, FP_P3); }

FPQ bool is_zero(const u256& a) { return (a.w[0] | a.w[1] | a.w[2] | a.w[3]) == 0; }
FPQ bool is_one(const u256& a) { return a.w[0] == 1 && (a.w[1] | a.w[2] | a.w[3]) == 0; }
FPQ bool eq(const u256& a, const u256& b) {
    return ((a.w[0] ^ b.w[0]) | (a.w[1] ^ b.w[1]) | (a.w[2] ^ b.w[2]) | (a.w[3] ^ b.w[3])) == 0;
}
// unsigned 256-bit compare: -1, 0, 1
FPQ int cmp(const u256& a, const u256& b) {
#pragma unroll
    for (int i = 3; i >= 0; --i) {
        if (a.w[i] < b.w[i]) return -1;
        if (a.w[i] > b.w[i]) return 1;
    }
    return 0;
}
FPQ bool lt(const u256& a, const u256& b) { return cmp(a, b) < 0; }
FPQ bool gt(const u256& a, const u256& b) { return cmp(a, b) > 0; }

FPQ uint64_t addc(uint64_t a, uint64_t b, uint64_t& carry) {
    uint64_t s = a + b;
    uint64_t c1 = s < a;
    uint64_t t = s + carry;
    uint64_t c2 = t < s;
    carry = c1 | c2;
    return t;
}
FPQ uint64_t subb(uint64_t a, uint64_t b, uint64_t& borrow) {
    uint64_t d = a - b;
    uint64_t b1 = a < b;
    uint64_t t = d - borrow;
    uint64_t b2 = d < borrow;
    borrow = b1 | b2;
    return t;
}
FPQ uint64_t add_raw(u256& r, const u256& a, const u256& b) {
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) r.w[i] = addc(a.w[i], b.w[i], c);
    return c;
}
FPQ uint64_t sub_raw(u256& r, const u256& a, const u256& b) {
    uint64_t br = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) r.w[i] = subb(a.w[i], b.w[i], br);
    return br;
}
FPQ uint64_t mulhi64(uint64_t a, uint64_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul64hi(a, b);
#else
    return (uint64_t)(((unsigned __int128)a * b) >> 64);
#endif
}

// ---- field ops on canonical residues ----
FPQ u256 add(const u256& a, const u256& b) {
    u256 r, t;
    uint64_t c = add_raw(r, a, b);
    uint64_t br = sub_raw(t, r, modulus());
    return (c || !br) ? t : r;
}
FPQ u256 sub(const u256& a, const u256& b) {
    u256 r, t;
    uint64_t br = sub_raw(r, a, b);
    add_raw(t, r, modulus());
    return br ? t : r;
}
FPQ u256 neg(const u256& a) {
    if (is_zero(a)) return a;
    u256 r;
    sub_raw(r, modulus(), a);
    return r;
}
// F(coeff) for an arbitrary 256-bit integer (ParseR1CS.jl:111): p > 2^253, so at most 5 subtractions
FPQ u256 reduce(u256 a) {
    const u256 p = modulus();
    for (int i = 0; i < 6; ++i) {
        u256 t;
        if (sub_raw(t, a, p)) break;
        a = t;
    }
    return a;
}

// CIOS Montgomery product: a*b*R^-1 mod p for a,b < p
FPQ u256 mont_mul(const u256& a, const u256& b) {
    const uint64_t p[4] = {FP_P0, FP_P1, FP_P2, FP_P3};
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        uint64_t carry = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            uint64_t lo = a.w[j] * b.w[i];
            uint64_t hi = mulhi64(a.w[j], b.w[i]);
            uint64_t c = 0;
            uint64_t s = addc(t[j], lo, c);
            hi += c;
            c = 0;
            s = addc(s, carry, c);
            hi += c;
            t[j] = s;
            carry = hi;
        }
        uint64_t c = 0;
        t[4] = addc(t[4], carry, c);
        t[5] = c;
        uint64_t m = t[0] * FP_N0INV;
        // t = (t + m*p) / 2^64
        {
            uint64_t lo = m * p[0];
            uint64_t hi = mulhi64(m, p[0]);
            uint64_t cc = 0;
            (void)addc(t[0], lo, cc);
            carry = hi + cc;
        }
#pragma unroll
        for (int j = 1; j < 4; ++j) {
            uint64_t lo = m * p[j];
            uint64_t hi = mulhi64(m, p[j]);
            uint64_t cc = 0;
            uint64_t s = addc(t[j], lo, cc);
            hi += cc;
            cc = 0;
            s = addc(s, carry, cc);
            hi += cc;
            t[j - 1] = s;
            carry = hi;
        }
        c = 0;
        t[3] = addc(t[4], carry, c);
        t[4] = t[5] + c;
    }
    u256 r = make(t[0], t[1], t[2], t[3]), u;
    uint64_t br = sub_raw(u, r, modulus());
    return (t[4] || !br) ? u : r;
}
FPQ u256 to_mont(const u256& a) { return mont_mul(a, r2modp()); }
FPQ u256 from_mont(const u256& a) { return mont_mul(a, make(1)); }
// canonical * canonical -> canonical
FPQ u256 mul(const u256& a, const u256& b) { return mont_mul(mont_mul(a, b), r2modp()); }

FPQ u256 shr1(const u256& a) {
    u256 r;
    r.w[0] = (a.w[0] >> 1) | (a.w[1] << 63);
    r.w[1] = (a.w[1] >> 1) | (a.w[2] << 63);
    r.w[2] = (a.w[2] >> 1) | (a.w[3] << 63);
    r.w[3] = a.w[3] >> 1;
    return r;
}
FPQ u256 halve_mod(const u256& x) {  // x/2 mod p for x < p
    if (x.w[0] & 1) {
        u256 t;
        add_raw(t, x, modulus());  // < 2^255, no carry
        return shr1(t);
    }
    return shr1(x);
}
// inverse of a canonical non-zero residue (binary extended Euclid); caller checks a != 0
FPQ u256 inv(const u256& a) {
    u256 u = a, v = modulus(), x1 = make(1), x2 = make(0);
    while (!is_one(u) && !is_one(v)) {
        while (!(u.w[0] & 1)) { u = shr1(u); x1 = halve_mod(x1); }
        while (!(v.w[0] & 1)) { v = shr1(v); x2 = halve_mod(x2); }
        if (cmp(u, v) >= 0) {
            u256 t;
            sub_raw(t, u, v);
            u = t;
            x1 = sub(x1, x2);
        } else {
            u256 t;
            sub_raw(t, v, u);
            v = t;
            x2 = sub(x2, x1);
        }
    }
    return is_one(u) ? x1 : x2;
}
// a^e by square-and-multiply in Montgomery form (utility; KAT cross-check for inv via Fermat)
FPQ u256 pow(const u256& a, const u256& e) {
    u256 base = to_mont(a), acc = to_mont(make(1));
    for (int i = 255; i >= 0; --i) {
        acc = mont_mul(acc, acc);
        if ((e.w[i >> 6] >> (i & 63)) & 1) acc = mont_mul(acc, base);
    }
    return from_mont(acc);
}

FPQ int bitlen(const u256& a) {
#pragma unroll
    for (int i = 3; i >= 0; --i)
        if (a.w[i]) {
#if defined(__HIP_DEVICE_COMPILE__)
            return 64 * i + (64 - __clzll((long long)a.w[i]));
#else
            return 64 * i + (64 - __builtin_clzll(a.w[i]));
#endif
        }
    return 0;
}
FPQ u256 shl(const u256& a, int s) {  // 0 <= s < 256
    u256 r = make(0);
    int ws = s >> 6, bs = s & 63;
#pragma unroll
    for (int i = 3; i >= 0; --i) {
        int src = i - ws;
        uint64_t v = 0;
        if (src >= 0) {
            v = a.w[src] << bs;
            if (bs && src >= 1) v |= a.w[src - 1] >> (64 - bs);
        }
        r.w[i] = v;
    }
    return r;
}
FPQ u256 shr(const u256& a, int s) {  // 0 <= s < 256
    u256 r = make(0);
    int ws = s >> 6, bs = s & 63;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int src = i + ws;
        uint64_t v = 0;
        if (src <= 3) {
            v = a.w[src] >> bs;
            if (bs && src <= 2) v |= a.w[src + 1] << (64 - bs);
        }
        r.w[i] = v;
    }
    return r;
}
// integer division a = q*b + r, b != 0 (mixed-radix rule, reference :1267-1268)
FPQ void divmod(const u256& a, const u256& b, u256& q, u256& r) {
    q = make(0);
    r = a;
    int la = bitlen(a), lb = bitlen(b);
    if (la < lb) return;
    // the divisors circuits produce are almost always 1, a power of two or a small integer
    if (la <= 64) { q.w[0] = a.w[0] / b.w[0]; r = make(a.w[0] % b.w[0]); return; }
    {
        const u256 low = shl(make(1), lb - 1);
        if (eq(b, low)) {   // b = 2^(lb-1)
            q = lb == 1 ? a : shr(a, lb - 1);
            u256 m;
            sub_raw(m, b, make(1));
            for (int i = 0; i < 4; ++i) r.w[i] = a.w[i] & m.w[i];
            return;
        }
    }
    for (int s = la - lb; s >= 0; --s) {
        u256 bs = shl(b, s);
        // shl may drop bits only if lb + s > 256, which cannot happen since lb + s <= la <= 256
        u256 t;
        if (!sub_raw(t, r, bs)) {
            r = t;
            q.w[s >> 6] |= (uint64_t)1 << (s & 63);
        }
    }
}
// is a * b > p as integers?  (a, b < 2^256; reference :1274)
FPQ bool mul_gt_p(const u256& a, const u256& b) {
    uint64_t t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        uint64_t carry = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            uint64_t lo = a.w[i] * b.w[j];
            uint64_t hi = mulhi64(a.w[i], b.w[j]);
            uint64_t c = 0;
            uint64_t s = addc(t[i + j], lo, c);
            hi += c;
            c = 0;
            s = addc(s, carry, c);
            hi += c;
            t[i + j] = s;
            carry = hi;
        }
        t[i + 4] = carry;
    }
    if (t[4] | t[5] | t[6] | t[7]) return true;
    return cmp(make(t[0], t[1], t[2], t[3]), modulus()) > 0;
}

// ---- dead-code utilities named by the north star (reference src/Math.jl:14-90, never called by
// the solver). Self-consistency KATs only: sqrt uses a fixed non-residue
// instead of the reference's rand(), so no bit-parity claim is possible or made.
FPQ bool sqrt(const u256& a, u256& out) {
    if (is_zero(a)) { out = a; return true; }
    // p - 1 = 2^28 * q
    const u256 q = make(0x9b9709143e1f593fULL, 0x181585d2833e8487ULL, 0x131a029b85045b68ULL, 0x000000030644e72eULL);
    const u256 qm1h = make(0xcdcb848a1f0fac9fULL, 0x0c0ac2e9419f4243ULL, 0x098d014dc2822db4ULL, 0x0000000183227397ULL);
    u256 z = pow(make(5), q);          // 5 is a quadratic non-residue mod p
    u256 x = pow(a, qm1h);             // a^((q-1)/2)
    u256 v = mul(a, x), w = mul(v, x);
    u256 y = z;
    int r = 28;
    while (!is_one(w)) {
        int k = 0;
        u256 tw = w;
        do { tw = mul(tw, tw); ++k; } while (!is_one(tw) && k < r);
        if (k >= r) return false;      // a is a non-residue
        u256 d = y;
        for (int i = 0; i < r - k - 1; ++i) d = mul(d, d);
        y = mul(d, d);
        r = k;
        v = mul(d, v);
        w = mul(w, y);
    }
    out = v;
    return true;
}

}  // namespace fp
