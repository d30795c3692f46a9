// classify.hip.hpp — k_classify_rows: static shape of every row, rule constants, R7 order.
// Part of the gfx950 device code of libecne_hip (see kernels.hip.hpp for the overview).
#pragma once
#include "dev_common.hip.hpp"

namespace ecne {

// ====================================================================================== classify
// popcount / ctz of a 256-bit value
__device__ __forceinline__ int popc256(const fp::u256& a) {
    return __popcll(a.w[0]) + __popcll(a.w[1]) + __popcll(a.w[2]) + __popcll(a.w[3]);
}
__device__ __forceinline__ int ctz256(const fp::u256& a) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (a.w[i]) return 64 * i + (__ffsll((long long)a.w[i]) - 1);
    return 256;
}

// -num / den in the field. Almost every divisor a circuit presents is +-1 (bit checks b * (b - 1) = 0, constants
// x = v): those need no inversion, and a lane that takes the binary-EGCD path (thousands of instructions) holds its
// whole wavefront up.
__device__ __forceinline__ fp::u256 neg_div(const fp::u256& num, const fp::u256& den) {
    if (fp::is_one(den)) return fp::neg(num);
    if (fp::is_one(fp::neg(den))) return num;
    return fp::mul(fp::neg(num), fp::inv(den));
}

// One wavefront classifies one row. wave_scratch: 8 u32 of LDS per wave (256-bit exponent bitmap).
__device__ void classify_row(const Job& J, uint32_t row, uint32_t* wave_scratch) {
    const int lane = lane_id();
    RowInfo ri = J.rinfo[row];   // structural bits and keys were laid down by the host
    const uint32_t c0 = J.rpC[row], c1 = J.rpC[row + 1];
    const uint32_t l = c1 - c0;
    uint32_t shape = ri.shape;
    // ---- R2 constants: values = [-a1/ax, -b1/bx]  (:916-927)
    if ((shape & SH_R2) && !(shape & SH_R2_DIV0)) {
        // lanes 0 and 1 each handle one part
        fp::u256 val = fp::make(0);
        if (lane < 2) {
            const uint32_t* rp = lane == 0 ? J.rpA : J.rpB;
            const uint32_t* col = lane == 0 ? J.colA : J.colB;
            const uint64_t* cf = lane == 0 ? J.coefA : J.coefB;
            fp::u256 slope = fp::make(0), icpt = fp::make(0);
            for (uint32_t k = rp[row]; k < rp[row + 1]; ++k) {
                uint32_t v = col[k];
                fp::u256 c = ld256(cf + 4ull * k);
                if (v == ri.x) slope = c;
                else if (v == 1) icpt = c;
            }
            val = neg_div(icpt, slope);
            st256(J.vals + 4ull * (ri.validx + lane), val);
        }
        fp::u256 v0 = shfl256(val, 0), v1 = shfl256(val, 1);
        if ((fp::is_zero(v0) && fp::is_one(v1)) || (fp::is_one(v0) && fp::is_zero(v1))) shape |= SH_R2_IS01;
    }
    if (!(shape & SH_HAS_AB) && l > 0) {
        // ---- R3 constant: -c[1]/c[x]  (:961-964)
        if ((shape & SH_R3) && lane == 0) {
            fp::u256 c1v = fp::make(0), cx = fp::make(0);
            for (uint32_t k = c0; k < c1; ++k) {
                uint32_t v = J.colC[k];
                if (v == 1) c1v = ld256(J.coefC + 4ull * k);
                else if (v == ri.x) cx = ld256(J.coefC + 4ull * k);
            }
            st256(J.vals + 4ull * ri.validx, neg_div(c1v, cx));
        }
        // ---- R4 pattern: multiset {1, -2^0..-2^(l-2)} (T) or its negation (T2)  (:999-1013)
        if (!(shape & SH_CZERO)) {
            bool isT = false, isT2 = false;
            uint32_t kpos = 0, kneg = 0;
            if (l <= 255) {
                if (lane < 8) wave_scratch[lane] = 0;       // T bitmap
                if (lane < 8) wave_scratch[8 + lane] = 0;   // T2 bitmap
                wg_fence();
                int n_one = 0, n_mone = 0, okT = 1, okT2 = 1;
                for (uint32_t base = c0; base < c1; base += 64) {
                    uint32_t k = base + lane;
                    bool act = k < c1;
                    fp::u256 c = act ? ld256(J.coefC + 4ull * k) : fp::make(2);
                    uint32_t v = act ? J.colC[k] : 0;
                    bool one = act && fp::is_one(c);
                    fp::u256 nc = fp::neg(c);
                    bool mone = act && fp::is_one(nc);
                    uint64_t m1 = __ballot(one), m2 = __ballot(mone);
                    n_one += __popcll(m1);
                    n_mone += __popcll(m2);
                    if (m1) kpos = __shfl(v, __ffsll((long long)m1) - 1, 64);
                    if (m2) kneg = __shfl(v, __ffsll((long long)m2) - 1, 64);
                    // exponent of -c (T) / of c (T2); the "1" / "-1" entries are the pivots
                    bool badT = false, badT2 = false;
                    if (act && !one) {           // T: every non-1 entry must be -2^k, k <= l-2, distinct
                        int e = (popc256(nc) == 1) ? ctz256(nc) : 999;
                        if (e > (int)l - 2) badT = true;
                        else if (atomicOr(&wave_scratch[e >> 5], 1u << (e & 31)) & (1u << (e & 31))) badT = true;
                    }
                    if (act && !mone) {          // T2: every non-(-1) entry must be 2^k
                        int e = (popc256(c) == 1) ? ctz256(c) : 999;
                        if (e > (int)l - 2) badT2 = true;
                        else if (atomicOr(&wave_scratch[8 + (e >> 5)], 1u << (e & 31)) & (1u << (e & 31))) badT2 = true;
                    }
                    if (__ballot(badT)) okT = 0;
                    if (__ballot(badT2)) okT2 = 0;
                }
                // l == 1: T = [1], T2 = [p-1]
                isT = okT && n_one == 1;
                isT2 = okT2 && n_mone == 1;
            } else {
                // l > 255: powers 2^k wrap modulo p for k >= 254. Quick reject (exactly one 1 / one -1),
                // then the literal multiset comparison, lanes striding over targets.
                int n_one = 0, n_mone = 0;
                // (four entries per lane and step, all loads first: a 1 025-term row is 5 steps of loads in flight instead of 17
                //  dependent round trips)
                for (uint32_t base = c0; base < c1; base += 256) {
                    fp::u256 c4[4];
                    uint32_t v4[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const uint32_t k = base + 64u * (uint32_t)j + (uint32_t)lane;
                        const bool act = k < c1;
                        c4[j] = act ? ld256(J.coefC + 4ull * k) : fp::make(2);
                        v4[j] = act ? J.colC[k] : 0;
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const uint64_t m1 = __ballot(fp::is_one(c4[j])), m2 = __ballot(fp::is_one(fp::neg(c4[j])));
                        n_one += __popcll(m1);
                        n_mone += __popcll(m2);
                        if (m1) kpos = __shfl(v4[j], __ffsll((long long)m1) - 1, 64);
                        if (m2) kneg = __shfl(v4[j], __ffsll((long long)m2) - 1, 64);
                    }
                }
                if (n_one == 1 && n_mone == 1) {
                    // T  <=> one "1"  and every 2^k mod p (k = 0..l-2) occurs exactly once among the -c
                    // T2 <=> one "-1" and every 2^k mod p occurs exactly once among the c
                    int okT = 1, okT2 = 1;
                    for (uint32_t tb = 0; tb < l - 1; tb += 64) {
                        uint32_t t = tb + lane;
                        bool act = t < l - 1;
                        fp::u256 pw = fp::make(1);
                        for (uint32_t s = 0; act && s < t; ++s) pw = fp::add(pw, pw);
                        int cntT = 0, cntT2 = 0;
                        if (act)
                            for (uint32_t k = c0; k < c1; ++k) {
                                fp::u256 c = ld256(J.coefC + 4ull * k);
                                if (fp::eq(c, pw)) cntT2++;
                                if (fp::eq(fp::neg(c), pw)) cntT++;
                            }
                        if (__ballot(act && cntT != 1)) okT = 0;
                        if (__ballot(act && cntT2 != 1)) okT2 = 0;
                    }
                    isT = okT;
                    isT2 = okT2;
                }
            }
            if (isT) shape |= SH_R4_T;
            if (isT2) shape |= SH_R4_T2;
            if (isT || isT2) {
                ri.kpos = kpos;
                ri.kneg = kneg;
                if (lane == 0) {   // F(2)^(l-1) - F(1), field arithmetic (:1033)
                    fp::u256 pw = fp::make(1);
                    for (uint32_t s = 0; s + 1 < l; ++s) pw = fp::add(pw, pw);
                    st256(J.vals + 4ull * (ri.validx + 1), fp::sub(pw, fp::make(1)));
                }
            }
        }
        // ---- R7 order: stable rank of |coefficient| in the orientation R7 will see (:1256-1265).
        // A T2-only row has been negated by R4 before R7 first looks at it.
        {
            const bool negated = (shape & SH_R4_T2) && !(shape & SH_R4_T);
            // A binary decomposition (the row matched {1, -2^0 .. -2^(l-2)} or its negation) needs no sorting: in the orientation
            // R7 sees, the |coefficients| are 1 (the pivot), 1 (the 2^0 term), 2, 4, ..., each once -- the two 1s keep their stored
            // order, the term with exponent e > 0 has rank e + 1. (Up to l = 250: beyond, -2^e for e >= 249 lies below R7's
            // "negative" threshold and counts as the large positive number it is, :1245-1259.) The rank sort below is O(l^2)
            // 256-bit compares on one wavefront -- 60 us for an 87-term Num2Bits row.
            const bool pattern = (shape & (SH_R4_T | SH_R4_T2)) && l >= 2 && l <= 250;
            // long sum rows usually carry one |coefficient| (all +-1): the stable order is then the stored one
            bool all_same = !pattern;
            if (pattern) {
                // positions of the two entries with |coefficient| 1 (the +1 and the -1 of the row)
                uint32_t p_one = 0xFFFFFFFFu, p_mone = 0xFFFFFFFFu;
                for (uint32_t base = c0; base < c1; base += 64) {
                    const uint32_t k = base + lane;
                    const bool act = k < c1;
                    const fp::u256 c = act ? ld256(J.coefC + 4ull * k) : fp::make(2);
                    const uint64_t m1 = __ballot(act && fp::is_one(c)), m2 = __ballot(act && fp::is_one(fp::neg(c)));
                    if (m1 && p_one == 0xFFFFFFFFu) p_one = base + (uint32_t)(__ffsll((long long)m1) - 1);
                    if (m2 && p_mone == 0xFFFFFFFFu) p_mone = base + (uint32_t)(__ffsll((long long)m2) - 1);
                }
                for (uint32_t base = c0; base < c1; base += 64) {
                    const uint32_t k = base + lane;
                    if (k >= c1) continue;
                    fp::u256 c = ld256(J.coefC + 4ull * k);
                    if (negated) c = fp::neg(c);
                    uint32_t rank;
                    if (k == p_one || k == p_mone) rank = (k == (p_one < p_mone ? p_one : p_mone)) ? 0u : 1u;
                    else { const fp::u256 a = r7_abs(c); rank = (uint32_t)ctz256(a) + 1u; }
                    J.csort[c0 + rank] = k - c0;
                }
            } else
            {
                fp::u256 first = ld256(J.coefC + 4ull * c0);
                if (negated) first = fp::neg(first);
                first = r7_abs(first);
                for (uint32_t base = c0; base < c1 && all_same; base += 256) {
                    fp::u256 c4[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const uint32_t k = base + 64u * (uint32_t)j + (uint32_t)lane;
                        c4[j] = k < c1 ? ld256(J.coefC + 4ull * k) : ld256(J.coefC + 4ull * c0);
                    }
                    bool diff = false;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        fp::u256 c = c4[j];
                        if (negated) c = fp::neg(c);
                        diff |= !fp::eq(r7_abs(c), first);
                    }
                    if (__ballot(diff)) all_same = false;
                }
            }
            if (pattern) { }
            else if (all_same)
                for (uint32_t k = c0 + lane; k < c1; k += 64) J.csort[k] = k - c0;
            else
            for (uint32_t base = c0; base < c1; base += 64) {
                uint32_t k = base + lane;
                bool act = k < c1;
                fp::u256 mine = fp::make(0);
                if (act) {
                    mine = ld256(J.coefC + 4ull * k);
                    if (negated) mine = fp::neg(mine);
                    mine = r7_abs(mine);
                }
                uint32_t rank = 0;
                for (uint32_t ob = c0; ob < c1; ob += 64) {
                    uint32_t ok_ = ob + lane;
                    fp::u256 oth = fp::make(0);
                    bool oact = ok_ < c1;
                    if (oact) {
                        oth = ld256(J.coefC + 4ull * ok_);
                        if (negated) oth = fp::neg(oth);
                        oth = r7_abs(oth);
                    }
                    uint32_t cnt = (c1 - ob) < 64 ? (c1 - ob) : 64;
                    for (uint32_t s = 0; s < cnt; ++s) {
                        fp::u256 o = shfl256(oth, (int)s);
                        uint32_t oidx = ob + s;
                        int cm = fp::cmp(o, mine);
                        if (act && (cm < 0 || (cm == 0 && oidx < k))) rank++;
                    }
                }
                if (act) J.csort[c0 + rank] = k - c0;
            }
            shape |= SH_R7_SORTED;
        }
    }
    if ((shape & SH_C_HAS1) && (shape & (SH_R4_T | SH_R4_T2 | SH_R5))) shape |= SH_TOUCH1;
    if (lane == 0) {
        ri.shape = shape;
        J.rinfo[row] = ri;
    }
}

// Rows with at most ECNE_CLS_LANE entries in C (99 % of an --O0 circuit: products, bit checks, x == y, constants, 1 = x + y) are
// classified by ONE lane each: 64 rows per wavefront, coalesced row descriptors, the C entries of the workgroup's rows staged in LDS.
// Round 5: the lane path is the STREAMING path and nothing else -- three entries at most, every array a fixed set of registers (the
// eight-entry version kept its sort keys in scratch memory, 272 B per lane, and the field inversion's registers capped the kernel at three
// wavefronts per SIMD), divisors +-1 only (bit checks b * (b - 1), constants x = v: every divisor a circuit presents in practice).
// Rows with more entries are on the layout's list (cls_list) and get a wavefront each (k_classify_wave, classify_row: the general path);
// a short row whose divisor needs the binary EGCD is DEFERRED to the same path: its id goes to a list, k_classify_wave runs once more
// behind the streaming pass (same results either way, tests/test_gpu_classify.py).
#ifndef ECNE_CLS_STAGE
#define ECNE_CLS_STAGE 768       // C entries of 256 consecutive rows staged in LDS per workgroup (27 KB: five workgroups per CU); at most 3 x 256 can be staged rows' entries
#endif
// -num / den for den = +-1; false: the caller defers the row
__device__ __forceinline__ bool neg_div_unit(const fp::u256& num, const fp::u256& den, fp::u256& out) {
    if (fp::is_one(den)) { out = fp::neg(num); return true; }
    if (fp::is_one(fp::neg(den))) { out = num; return true; }
    return false;
}
// cfC / clC: where the row's C entries are read from -- the CSR arrays themselves (cbase = 0), or the workgroup's LDS copy of the
// entries [cbase, ...) of its 256 rows. Returns false when the row has to be deferred (nothing it wrote matters: the general path writes
// everything again).
// UNIT = true: the streaming pass (divisors +-1 only, anything else defers the row). UNIT = false: the same lane code with the full
// field inversion, for the deferred rows (k_classify_wave's second part) -- a circuit with arbitrary linear coefficients (circom --O1 / --O2
// output, hand-written R1CS) defers a large share of its rows, and one WAVEFRONT per such row was a cliff (round 5's advisor finding).
// -num / den with 1 / den at hand (UNIT = false: the deferred rows; the inverses come from ONE inversion per wavefront, wave_batch_inv)
__device__ __forceinline__ fp::u256 neg_div_with(const fp::u256& num, const fp::u256& den, const fp::u256& den_inv) {
    if (fp::is_one(den)) return fp::neg(num);
    if (fp::is_one(fp::neg(den))) return num;
    return fp::mul(fp::neg(num), den_inv);
}
template <bool UNIT>
__device__ __forceinline__ bool classify_row_lane(const Job& J, uint32_t row, RowInfo ri, const uint64_t* __restrict__ cfC, const uint32_t* __restrict__ clC, uint32_t cbase,
                                                  const fp::u256* inv2 = nullptr) {
    const uint32_t shape_in = ri.shape, kpos_in = ri.kpos, kneg_in = ri.kneg;
    // a product a * b = c (or any row with C empty that is no bit check) needs nothing from this pass
    if ((shape_in & SH_HAS_AB) && !(shape_in & SH_R2) && !((shape_in & SH_C_HAS1) && (shape_in & SH_R5))) return true;
    const uint32_t c0 = J.rpC[row], c1 = J.rpC[row + 1];
    const uint32_t l = c1 - c0;
    uint32_t shape = ri.shape;
    if ((shape & SH_R2) && !(shape & SH_R2_DIV0)) {
        fp::u256 val[2];
#pragma unroll
        for (int part = 0; part < 2; ++part) {
            const uint32_t* rp = part == 0 ? J.rpA : J.rpB;
            const uint32_t* col = part == 0 ? J.colA : J.colB;
            const uint64_t* cf = part == 0 ? J.coefA : J.coefB;
            fp::u256 slope = fp::make(0), icpt = fp::make(0);
            for (uint32_t k = rp[row]; k < rp[row + 1]; ++k) {
                uint32_t v = col[k];
                fp::u256 c = ld256(cf + 4ull * k);
                if (v == ri.x) slope = c;
                else if (v == 1) icpt = c;
            }
            if (UNIT) { if (!neg_div_unit(icpt, slope, val[part])) return false; }
            else val[part] = neg_div_with(icpt, slope, inv2[part]);
        }
        st256(J.vals + 4ull * ri.validx, val[0]);
        st256(J.vals + 4ull * (ri.validx + 1), val[1]);
        if ((fp::is_zero(val[0]) && fp::is_one(val[1])) || (fp::is_one(val[0]) && fp::is_zero(val[1]))) shape |= SH_R2_IS01;
    }
    if (!(shape & SH_HAS_AB) && l > 0) {
        // the (at most three) entries, in registers
        fp::u256 c[ECNE_CLS_LANE];
        uint32_t v[ECNE_CLS_LANE];
#pragma unroll
        for (uint32_t e = 0; e < ECNE_CLS_LANE; ++e) {
            const bool on = e < l;
            c[e] = on ? ld256(cfC + 4ull * (c0 + e - cbase)) : fp::make(0);
            v[e] = on ? clC[c0 + e - cbase] : 0u;
        }
        if (shape & SH_R3) {
            fp::u256 c1v = fp::make(0), cx = fp::make(0), tv;
#pragma unroll
            for (uint32_t e = 0; e < ECNE_CLS_LANE; ++e) {
                if (e >= l) continue;
                if (v[e] == 1) c1v = c[e];
                else if (v[e] == ri.x) cx = c[e];
            }
            if (UNIT) { if (!neg_div_unit(c1v, cx, tv)) return false; }
            else tv = neg_div_with(c1v, cx, inv2[0]);
            st256(J.vals + 4ull * ri.validx, tv);
        }
        if (!(shape & SH_CZERO)) {
            uint32_t n_one = 0, n_mone = 0, kpos = 0, kneg = 0;
            uint32_t maskT = 0, maskT2 = 0;     // exponents seen (l <= 3: exponents 0..1)
            bool okT = true, okT2 = true;
#pragma unroll
            for (uint32_t e = 0; e < ECNE_CLS_LANE; ++e) {
                if (e >= l) continue;
                const fp::u256 nc = fp::neg(c[e]);
                const bool one = fp::is_one(c[e]), mone = fp::is_one(nc);
                if (one) { ++n_one; if (n_one == 1) kpos = v[e]; }
                if (mone) { ++n_mone; if (n_mone == 1) kneg = v[e]; }
                if (!one) {
                    int ex = (popc256(nc) == 1) ? ctz256(nc) : 999;
                    if (ex > (int)l - 2 || (maskT >> ex & 1)) okT = false; else maskT |= 1u << ex;
                }
                if (!mone) {
                    int ex = (popc256(c[e]) == 1) ? ctz256(c[e]) : 999;
                    if (ex > (int)l - 2 || (maskT2 >> ex & 1)) okT2 = false; else maskT2 |= 1u << ex;
                }
            }
            const bool isT = okT && n_one == 1, isT2 = okT2 && n_mone == 1;
            if (isT) shape |= SH_R4_T;
            if (isT2) shape |= SH_R4_T2;
            if (isT || isT2) {
                ri.kpos = kpos;
                ri.kneg = kneg;
                st256(J.vals + 4ull * (ri.validx + 1), fp::make(l == 1 ? 0ull : l == 2 ? 1ull : 3ull));      // 2^(l-1) - 1
            }
        }
        // R7 order: stable by |signed coefficient| -- the rank of each entry from three comparisons
        {
            const bool negated = (shape & SH_R4_T2) && !(shape & SH_R4_T);
            fp::u256 key[ECNE_CLS_LANE];
#pragma unroll
            for (uint32_t e = 0; e < ECNE_CLS_LANE; ++e) key[e] = r7_abs(negated ? fp::neg(c[e]) : c[e]);
            // before(i, j), i < j: entry i sorts in front of entry j (ties keep the stored order)
            const bool b01 = fp::cmp(key[0], key[1]) <= 0, b02 = fp::cmp(key[0], key[2]) <= 0, b12 = fp::cmp(key[1], key[2]) <= 0;
            uint32_t rank0 = 0, rank1 = 0, rank2 = 0;
            if (l >= 2) { if (b01) rank1++; else rank0++; }
            if (l >= 3) { if (b02) rank2++; else rank0++; if (b12) rank2++; else rank1++; }
            J.csort[c0 + rank0] = 0;
            if (l >= 2) J.csort[c0 + rank1] = 1;
            if (l >= 3) J.csort[c0 + rank2] = 2;
            shape |= SH_R7_SORTED;
        }
    }
    if ((shape & SH_C_HAS1) && (shape & (SH_R4_T | SH_R4_T2 | SH_R5))) shape |= SH_TOUCH1;
    if (shape != shape_in || ri.kpos != kpos_in || ri.kneg != kneg_in) {     // (most descriptors come out as the host laid them down)
        ri.shape = shape;
        J.rinfo[row] = ri;
    }
    return true;
}

// ---- the deferred short rows: a divisor that is neither 1 nor -1. The binary EGCD is data-dependent loops inside loops -- 64 lanes each
// running their own is the worst case of every trip count in lockstep, several hundred microseconds per wavefront -- so the wavefront
// inverts ONCE: Montgomery's trick over the lanes (prefix and suffix products by shuffles, one EGCD of the total that every lane runs
// on the same value, i.e. without divergence), each lane's two divisors folded into one factor first.
__device__ __forceinline__ fp::u256 shfl256_from(const fp::u256& v, int src) { return shfl256(v, src < 0 ? 0 : src > 63 ? 63 : src); }
// 1 / e for every lane's e (e != 0; a lane that needs nothing passes 1). Montgomery form inside.
__device__ __forceinline__ fp::u256 wave_batch_inv(const fp::u256& e) {
    const int lane = lane_id();
    const fp::u256 m = fp::to_mont(e);
    fp::u256 pre = m, suf = m;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const fp::u256 a = shfl256_from(pre, lane - d), b = shfl256_from(suf, lane + d);
        if (lane >= d) pre = fp::mont_mul(pre, a);
        if (lane + d < 64) suf = fp::mont_mul(suf, b);
    }
    const fp::u256 total = fp::from_mont(shfl256(pre, 63));
    const fp::u256 tinv = fp::to_mont(fp::inv(total));            // (uniform across the wavefront)
    const fp::u256 one = fp::to_mont(fp::make(1));
    const fp::u256 left = shfl256_from(pre, lane - 1), right = shfl256_from(suf, lane + 1);
    fp::u256 r = fp::mont_mul(lane > 0 ? left : one, lane < 63 ? right : one);
    r = fp::mont_mul(r, tinv);
    return fp::from_mont(r);
}
// the divisors classify_row_lane will divide by for this row (1 where it divides by nothing): R2's two slopes (:916-927) or R3's c[x] (:961-964)
__device__ __forceinline__ void deferred_divisors(const Job& J, uint32_t row, const RowInfo& ri, fp::u256& d0, fp::u256& d1) {
    d0 = fp::make(1); d1 = fp::make(1);
    const uint32_t shape = ri.shape;
    if ((shape & SH_HAS_AB) && !(shape & SH_R2) && !((shape & SH_C_HAS1) && (shape & SH_R5))) return;
    if ((shape & SH_R2) && !(shape & SH_R2_DIV0)) {
        for (uint32_t k = J.rpA[row]; k < J.rpA[row + 1]; ++k) if (J.colA[k] == ri.x) d0 = ld256(J.coefA + 4ull * k);
        for (uint32_t k = J.rpB[row]; k < J.rpB[row + 1]; ++k) if (J.colB[k] == ri.x) d1 = ld256(J.coefB + 4ull * k);
    } else if (!(shape & SH_HAS_AB) && (shape & SH_R3)) {
        const uint32_t c0 = J.rpC[row], c1 = J.rpC[row + 1];
        for (uint32_t k = c0; k < c1 && k < c0 + ECNE_CLS_LANE; ++k) if (J.colC[k] != 1u && J.colC[k] == ri.x) d0 = ld256(J.coefC + 4ull * k);
    }
}
// 64 deferred rows, one per lane (all lanes of the wavefront call this; `on` = this lane has a row)
__device__ __forceinline__ void classify_deferred64(const Job& J, uint32_t row, bool on) {
    RowInfo ri;
    fp::u256 d[2];
    d[0] = fp::make(1); d[1] = fp::make(1);
    if (on) { ri = J.rinfo[row]; deferred_divisors(J, row, ri, d[0], d[1]); }
    fp::u256 e[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) e[i] = (fp::is_zero(d[i]) || fp::is_one(d[i]) || fp::is_one(fp::neg(d[i]))) ? fp::make(1) : d[i];
    const fp::u256 pinv = wave_batch_inv(fp::mul(e[0], e[1]));
    fp::u256 inv2[2];
    inv2[0] = fp::mul(pinv, e[1]);
    inv2[1] = fp::mul(pinv, e[0]);
    if (on) (void)classify_row_lane<false>(J, row, ri, J.coefC, J.colC, 0u, inv2);
}

// The streaming pass: the short rows, one lane per row (round 5: nothing else in this kernel -- with the long rows' wavefront path in
// the same kernel, its field inversion and rank sort set the register count for the streaming lanes too, 148 VGPRs = three wavefronts
// per SIMD; the long rows and the deferred ones get k_classify_wave, launched behind it).
__global__ __launch_bounds__(256) void k_classify_rows(const Job* jobs, uint32_t job_index) {
    __shared__ Job sJ;
    if (threadIdx.x < sizeof(Job) / 4) ((uint32_t*)&sJ)[threadIdx.x] = ((const uint32_t*)&jobs[job_index])[threadIdx.x];
    __syncthreads();
    {
        // the C entries of the workgroup's 256 rows are one contiguous stretch of the CSR arrays: staged in LDS with coalesced 16-byte
        // loads (36 B per entry), then every lane classifies its row from there. A stretch that does not fit (a long row among the 256:
        // its own entries alone are more than the buffer) is read from device memory as before.
        __shared__ uint4 s_coef[2 * ECNE_CLS_STAGE];
        __shared__ uint32_t s_col[ECNE_CLS_STAGE];
        const uint32_t nb0 = gridDim.x;
        for (uint32_t rb = blockIdx.x * 256; rb < sJ.nC; rb += nb0 * 256) {      // (uniform trip count: barriers inside)
            const uint32_t rend = rb + 256 < sJ.nC ? rb + 256 : sJ.nC;
            const uint32_t clo = sJ.rpC[rb], chi = sJ.rpC[rend];
            const uint32_t n_ent = chi - clo;
            const bool staged = n_ent <= ECNE_CLS_STAGE;
            if (staged) {
                const uint4* const src = reinterpret_cast<const uint4*>(sJ.coefC) + 2ull * clo;
                // (plain strided loops: holding a thread's loads in registers first -- twelve in flight -- cost occupancy and was 51 us instead of 38)
                for (uint32_t i = threadIdx.x; i < 2 * n_ent; i += 256) s_coef[i] = src[i];
                for (uint32_t i = threadIdx.x; i < n_ent; i += 256) s_col[i] = sJ.colC[clo + i];
            }
            __syncthreads();
            const uint32_t row = rb + threadIdx.x;
            bool defer = false;
            if (row < sJ.nC) {
                const RowInfo ri = sJ.rinfo[row];
                if (ri.lenC <= ECNE_CLS_LANE) {
                    if (staged) defer = !classify_row_lane<true>(sJ, row, ri, reinterpret_cast<const uint64_t*>(s_coef), s_col, clo);
                    else defer = !classify_row_lane<true>(sJ, row, ri, sJ.coefC, sJ.colC, 0u);
                }      // (longer rows are on the layout's list, cls_list: k_classify_wave)
            }
            {   // the deferred rows of a wavefront in one atomic (cls_defer[0] = count, zeroed by the host before the launch)
                const uint64_t m = __ballot(defer);
                if (m) {
                    uint32_t base = 0;
                    if ((threadIdx.x & 63) == (uint32_t)(__ffsll((long long)m) - 1)) base = atomicAdd(&sJ.cls_defer[0], (uint32_t)__popcll(m));
                    base = (uint32_t)__shfl((int)base, __ffsll((long long)m) - 1, 64);
                    if (defer) sJ.cls_defer[1 + base + (uint32_t)__popcll(m & ((1ull << (threadIdx.x & 63)) - 1ull))] = row;
                }
            }
            __syncthreads();
        }
    }
}

// Behind the streaming pass: one wavefront per row, the general path (classify_row), for the rows the layout listed (cls_list: more than
// ECNE_CLS_LANE entries in C -- latency-bound, each a chain of loads); one LANE per row, with the field inversion, for the short rows
// k_classify_rows deferred (round 6; until then a wavefront each).
__global__ __launch_bounds__(256) void k_classify_wave(const Job* jobs, uint32_t job_index) {
    __shared__ uint32_t scratch[4][16];
    __shared__ Job sJ;
    if (threadIdx.x < sizeof(Job) / 4) ((uint32_t*)&sJ)[threadIdx.x] = ((const uint32_t*)&jobs[job_index])[threadIdx.x];
    __syncthreads();
    const uint32_t n0 = sJ.nBigCls, nd = sJ.cls_defer[0];
    const uint32_t wave = threadIdx.x >> 6;
    for (uint32_t i = blockIdx.x * 4 + wave; i < n0; i += gridDim.x * 4) classify_row(sJ, sJ.cls_list[i], scratch[wave]);
    // (the workgroups are dealt the deferred rows from the far end of the grid: the first ones hold the long rows; uniform trip count per
    //  wavefront: classify_deferred64 shuffles across its lanes)
    for (uint32_t i0 = ((gridDim.x - 1u - blockIdx.x) * 4u + wave) * 64u; i0 < nd; i0 += gridDim.x * 256u) {
        const uint32_t i = i0 + (uint32_t)lane_id();
        const bool on = i < nd;
        classify_deferred64(sJ, on ? sJ.cls_defer[1 + i] : 0u, on);
    }
}

}  // namespace ecne
