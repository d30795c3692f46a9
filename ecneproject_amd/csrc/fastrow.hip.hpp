// fastrow.hip.hpp — what one pop does, decided in registers: the decision (fast_decide) for the common row shapes from the row
// record and the flag bytes, its commit, and the wavefront walk of a long row. Shared by the chain executor (chain.hip.hpp), the
// fast wavefront / workgroup round (wave2.hip.hpp) and the record path of the multi-workgroup round (rounds.hip.hpp).
// Part of the gfx950 device code of libecne_hip (see kernels.hip.hpp for the overview).
#pragma once
#include "schedule.hip.hpp"

namespace ecne {

// Loads / stores that are known to hit device memory go through global-address-space pointers: a generic (flat) access
// also counts on the LDS counter, so waiting for a ds_read would wait for every flat load in flight as well.
#define ECNE_GLOBAL __attribute__((address_space(1)))
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <class T> __device__ __forceinline__ ECNE_GLOBAL T* as_global(T* p) { return (ECNE_GLOBAL T*)p; }
template <class T> __device__ __forceinline__ const ECNE_GLOBAL T* as_global(const T* p) { return (const ECNE_GLOBAL T*)p; }

__device__ __forceinline__ uint32_t rdlane(uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); }

// ---- what one pop of a row of the four common shapes does, decided in registers from the row record and the flag bytes
// (x == y rows with a bound of the third kind: from the limbs). Nothing is written: the caller commits FastOut for the rows
// that make it into the prefix. Shared by the fast wavefront / workgroup round and the multi-workgroup round.
struct FastIn {
    uint32_t shape, rx, kpos, kneg, k1, k2, nA, nB, nE;
    uint32_t w[16];                 // row record: w[1 + e] = variable of entry e (A, B, C)
    uint8_t fl[15], fa, fb, fx;     // flag bytes: of the entries (products, sums), of k1 / k2 (x == y), of x (bit check)
    uint8_t flip_in;
    bool live, xy, f2, f4, bigsum;
    bool r4s = false;               // a short binary-decomposition row (R4 shape, l > 2): taken while R4's precondition fails
    bool r3f = false;               // a constant row x = c (R3 shape and none of R4..R6's)
    bool r3x = false;               // ... x = c written as x - 1 = 0 / 1 - x = 0: the R4 (l = 2) and R5 shapes against the constant wire as well
    bool r6f = false;               // 1 = x + y (R6 shape and none of R3..R5's), record = {constant wire, x, y}
    uint32_t validx = 0;            // the row's constants in J.vals (R3: c)
};
struct FastOut {
    uint32_t wva = 0, wvb = 0;      // variables whose flag byte (and maybe bounds) this pop changes
    uint8_t wfa = 0, wfb = 0;
    bool wa = false, wb = false, a01 = false, b01 = false, r2 = false, flip_w = false;
    bool xa_w = false, xb_w = false;            // x == y rows decided on the limbs: new bounds of k1 / k2 (a constant row: of x)
    bool r3v = false;                           // a constant row: values[x] = {xlb0} (:955-961)
    bool v6a = false, v6b = false;              // 1 = x + y: values[k1] / values[k2] = {ub, lb} as written (:1205-1213)
    uint32_t d_h2 = 0, d_h5 = 0;
    fp::u256 xlb0 = fp::make(0), xub0 = fp::make(0), xlb1 = fp::make(0), xub1 = fp::make(0);
    uint8_t flip_new = 0;
    uint32_t ev[5] = {0, 0, 0, 0, 0}, nev = 0;  // REQUEUE events, in the reference's order
    uint32_t d_steps = 0, d_nuniq = 0, d_h0 = 0, d_h1 = 0, d_h3 = 0, d_h4 = 0;
    bool slow = false;              // not settled here: the general executor takes the row
    uint32_t reason = 7;
};
__device__ __forceinline__ void fast_decide(const Job& J, const FastIn& I, FastOut& O) {
    const bool live = I.live, xy = I.xy, f2 = I.f2, f4 = I.f4, bigsum = I.bigsum, r4s = I.r4s;
    const uint32_t shape = I.shape, rx = I.rx, kpos = I.kpos, kneg = I.kneg, k1 = I.k1, k2 = I.k2, nA = I.nA, nB = I.nB, nE = I.nE;
    const uint32_t* const w = I.w;
    const uint8_t* const fl = I.fl;
    uint8_t fa = I.fa, fb = I.fb;
    const uint8_t fx = I.fx, flip_in = I.flip_in;
    uint32_t &wva = O.wva, &wvb = O.wvb, &nev = O.nev, &reason = O.reason;
    uint8_t &wfa = O.wfa, &wfb = O.wfb, &flip_new = O.flip_new;
    bool &wa = O.wa, &wb = O.wb, &a01 = O.a01, &b01 = O.b01, &r2 = O.r2, &flip_w = O.flip_w, &xa_w = O.xa_w, &xb_w = O.xb_w, &slow = O.slow;
    fp::u256 &xlb0 = O.xlb0, &xub0 = O.xub0, &xlb1 = O.xlb1, &xub1 = O.xub1;
    uint32_t* const ev = O.ev;
    uint32_t &d_steps = O.d_steps, &d_nuniq = O.d_nuniq, &d_h0 = O.d_h0, &d_h1 = O.d_h1, &d_h3 = O.d_h3, &d_h4 = O.d_h4;
    auto emit = [&](uint32_t v) {
        if (nev == 0) ev[0] = v; else if (nev == 1) ev[1] = v; else if (nev == 2) ev[2] = v; else if (nev == 3) ev[3] = v; else ev[4] = v;
        ++nev;
    };
    if (live && !slow) {
        if (f2) {
            // R2 check_quadratic (:875-942); errors are the general executor's business
            if (shape & SH_R2_BOUNDSERR) { slow = true; reason = 2; }
            else if ((shape & SH_R2) && !(fx & 2)) {
                if (shape & SH_R2_DIV0) { slow = true; reason = 2; }
                else {
                    wva = rx; wa = true; r2 = true;
                    wfa = (uint8_t)((fx | 2) & ~16u);
                    if (shape & SH_R2_IS01) { wfa = (uint8_t)((wfa & ~12u) | 4u); a01 = true; }
                    emit(rx);
                    d_steps = 1; d_h1 = 1;
                }
            }
        } else if (xy && ((fa | fb) & 8u) && k1 != k2 && nE == 2) {
            // x == y with a bound that is neither [0,1] nor [0,p-1] (a constant wired on, say): the same three rules on the
            // limbs themselves, statement for statement exec_xy_lane() (rules_lane.hip.hpp)
            const bool sw = (shape & SH_R56_SWAP) != 0;
            const uint8_t fa_in = fa, fb_in = fb;
            fp::u256 lb0 = (fa & 8u) ? ld256(J.lb + 4ull * k1) : fp::make(0), ub0 = (fa & 8u) ? ld256(J.ub + 4ull * k1) : ((fa & 4u) ? fp::make(1) : fp::pminus1());
            fp::u256 lb1 = (fb & 8u) ? ld256(J.lb + 4ull * k2) : fp::make(0), ub1 = (fb & 8u) ? ld256(J.ub + 4ull * k2) : ((fb & 4u) ? fp::make(1) : fp::pminus1());
            if (((fa ^ fb) & 1u)) {                               // R1
                if (!(fa & 1)) { fa |= 3; emit(k1); } else { fb |= 3; emit(k2); }
                d_nuniq++; d_steps++; d_h0++;
            }
            {                                                     // R4, l == 2
                flip_new = (uint8_t)(flip_in ^ 1);
                flip_w = true;
                const uint32_t new_key = flip_new ? kneg : kpos;
                const bool n_is_a = new_key == k1;
                uint8_t fn = n_is_a ? fa : fb, fo_ = n_is_a ? fb : fa;
                const fp::u256 lbn = n_is_a ? lb0 : lb1, ubn = n_is_a ? ub0 : ub1;
                if (fo_ & 4) {
                    if (!(fp::is_zero(lbn) && fp::is_one(ubn)) && fp::cmp(ubn, fp::make(1)) > 0) {
                        if (n_is_a) { lb0 = fp::make(0); ub0 = fp::make(1); xa_w = true; } else { lb1 = fp::make(0); ub1 = fp::make(1); xb_w = true; }
                        fn = (uint8_t)((fn & ~12u) | 4u | 2u);
                        d_steps++; d_h3++;
                        emit(new_key);
                    }
                    if ((fn & 1) && !(fo_ & 1)) {
                        fo_ |= 3;
                        d_nuniq++; d_steps++; d_h3++;
                        emit(n_is_a ? k2 : k1);
                    }
                }
                if (n_is_a) { fa = fn; fb = fo_; } else { fb = fn; fa = fo_; }
            }
            if (!fp::eq(ub1, ub0) || !fp::eq(lb1, lb0) || ((fa ^ fb) & 1u)) {      // R5
                bool cha = false, chb = false;
                if ((fa ^ fb) & 1u) { fa |= 3; d_nuniq += 2; cha = chb = true; }
                const fp::u256 mn = fp::cmp(ub0, ub1) <= 0 ? ub0 : ub1;
                const fp::u256 mx = fp::cmp(lb0, lb1) >= 0 ? lb0 : lb1;
                const bool na = fp::cmp(ub0, mn) > 0 || fp::cmp(lb0, mx) < 0, nb = fp::cmp(ub1, mn) > 0 || fp::cmp(lb1, mx) < 0;
                if (na) { lb0 = mx; ub0 = mn; xa_w = true; fa = (uint8_t)((fa & ~12u) | bounds_class_bits(mx, mn) | 2u); }
                if (nb) { lb1 = mx; ub1 = mn; xb_w = true; fb = (uint8_t)((fb & ~12u) | bounds_class_bits(mx, mn) | 2u); }
                cha |= na; chb |= nb;
                const uint32_t nset = (cha ? 1u : 0u) + (chb ? 1u : 0u);
                d_steps += nset;
                if (nset) d_h4++;
                if (sw) { if (chb) emit(k2); if (cha) emit(k1); }
                else { if (cha) emit(k1); if (chb) emit(k2); }
            }
            xlb0 = lb0; xub0 = ub0; xlb1 = lb1; xub1 = ub1;
            wva = k1; wfa = fa; wa = fa != fa_in || xa_w;
            wvb = k2; wfb = fb; wb = fb != fb_in || xb_w;
            // R7 / R8 in reach? R7 with two non-unique, known variables and coefficients +-1 fires iff the first one (C order)
            // has ub <= lb (:1267-1269); R8 needs a group tag on every non-unique variable
            const bool nua = !(fa & 1), nub = !(fb & 1);
            if ((nua || nub) && !((nua && !(fa & 2)) || (nub && !(fb & 2)))) {
                const bool tagged = !((nua && !(fa & 16)) || (nub && !(fb & 16)));
                const bool first_is_a = !sw;
                const bool r7 = nua && nub && (first_is_a ? fp::cmp(ub0, lb0) <= 0 : fp::cmp(ub1, lb1) <= 0);
                if (tagged || r7) { slow = true; reason = 4; }
            }
        } else if (I.r3f) {
            // a constant row c_x * x + c_1 = 0 (R3 check_linear :949-988; R1 first, nothing else has anything to do afterwards:
            // x ends up unique, so R7 / R8 find no non-unique variable). fx = x's flag byte.
            uint8_t f = fx;
            uint32_t cnt = 0;
#pragma unroll
            for (uint32_t e = 0; e < 15; ++e) if (e < nE && !(fl[e] & 1)) ++cnt;
            if (cnt == 1 && !(f & 1)) {                                   // R1 (:827-873): x is the one non-unique variable
                f |= 3; emit(rx);
                O.d_nuniq++; O.d_steps++; O.d_h0++;
            } else if (cnt != 0) { slow = true; reason = 1; }             // (a non-unique constant wire: never seen; general executor)
            if (!slow) {
                // (all loads first; the pop of a constant row whose x already holds c -- every pop after the first -- writes nothing)
                const fp::u256 tv = ld256(J.vals + 4ull * I.validx);
                const uint8_t nv = J.nvalues[rx];
                const fp::u256 va = ld256(J.values + 8ull * rx), lbx = ld256(J.lb + 4ull * rx), ubx = ld256(J.ub + 4ull * rx);
                const bool same = nv == 1 && fp::eq(va, tv);
                const bool bsame = fp::eq(lbx, tv) && fp::eq(ubx, tv);
                bool new_info = false;
                if (!same) { O.d_steps++; O.d_h2++; new_info = true; O.r3v = true; }
                if (!(f & 1)) { O.d_nuniq++; new_info = true; }
                f = (uint8_t)(((f | 3) & ~12u) | bounds_class_bits(tv, tv));
                xlb0 = tv; xub0 = tv; xa_w = !bsame;
                wva = rx; wfa = f; wa = f != fx || !same || !bsame;
                if (new_info) emit(rx);
                if (I.r3x) {
                    // the row also has the x == y shapes, the other variable being the constant wire: R4 (:991-1076, l = 2) negates
                    // the row and then needs the variable that is not the pivot to have bounds exactly [0,1] -- x has [c,c] now, the
                    // constant wire must not have them either; R5 (:1078-1146) does nothing while both bounds and both unique bits
                    // agree. Anything else: the general executor.
                    uint8_t f1w = 0;
                    bool okw = nE == 2 && nA + nB == 0;
#pragma unroll
                    for (uint32_t e = 0; e < 2; ++e) {
                        if (w[1 + e] == 1u) f1w = fl[e];
                        else if (w[1 + e] != rx) okw = false;
                    }
                    okw = okw && rx != 1u && (w[1] == 1u || w[2] == 1u) && (f1w & 3) == 3 && !(f1w & 4);
                    if (okw) {
                        const fp::u256 lb1 = ld256(J.lb + 4ull), ub1 = ld256(J.ub + 4ull);       // bounds of the constant wire (variable 1)
                        okw = fp::eq(lb1, tv) && fp::eq(ub1, tv);
                    }
                    if (!okw) { slow = true; reason = 1; }
                    else { flip_w = true; flip_new = (uint8_t)(flip_in ^ 1); }
                }
            }
        } else if (I.r6f) {
            // 1 = x + y (R6 checkOnePropagateBounds :1148-1232, after R1), on the limbs, statement for statement exec_row()
            // (rules_wave.hip.hpp). The record has to be {constant wire, x, y}, the constant wire unique -- anything else is the
            // general executor's.
            bool ok = nE == 3 && nA + nB == 0 && k1 != k2;
            uint32_t seen = 0;
#pragma unroll
            for (uint32_t e = 0; e < 3; ++e) {
                const uint32_t v = w[1 + e];
                if (v == 1u) { seen |= 1u; if ((fl[e] & 3) != 3) ok = false; }
                else if (v == k1) seen |= 2u;
                else if (v == k2) seen |= 4u;
                else ok = false;
            }
            if (!ok || seen != 7u) { slow = true; reason = 1; }
            else {
                const bool sw = (shape & SH_R56_SWAP) != 0;
                const uint8_t fa_in = fa, fb_in = fb;
                if (((fa ^ fb) & 1u)) {                               // R1 (:827-873): one of the two is the only non-unique variable
                    if (!(fa & 1)) { fa |= 3; emit(k1); } else { fb |= 3; emit(k2); }
                    d_nuniq++; d_steps++; d_h0++;
                }
                fp::u256 lb0 = ld256(J.lb + 4ull * k1), ub0 = ld256(J.ub + 4ull * k1);
                fp::u256 lb1 = ld256(J.lb + 4ull * k2), ub1 = ld256(J.ub + 4ull * k2);
                if (!fp::eq(ub1, ub0) || !fp::eq(lb1, lb0) || ((fa ^ fb) & 1u)) {
                    bool cha = false, chb = false, proceed = true;
                    if ((fa ^ fb) & 1u) { fa |= 3; fb |= 3; d_nuniq += 2; cha = chb = true; }      // (:1186-1189)
                    const fp::u256 mn = fp::cmp(ub0, ub1) <= 0 ? ub0 : ub1;
                    const fp::u256 mx = fp::cmp(lb0, lb1) >= 0 ? lb0 : lb1;
                    if (!fp::is_one(mn) || !fp::is_zero(mx)) proceed = false;                 // (:1196-1199) returns before counting
                    const bool na = proceed && (fp::cmp(ub0, mn) > 0 || fp::cmp(lb0, mx) < 0);
                    const bool nb = proceed && (fp::cmp(ub1, mn) > 0 || fp::cmp(lb1, mx) < 0);
                    if (na) { xlb0 = mx; xub0 = mn; xa_w = true; O.v6a = true; fa = (uint8_t)(((fa | 2u) & ~12u) | bounds_class_bits(mx, mn)); }
                    if (nb) { xlb1 = mx; xub1 = mn; xb_w = true; O.v6b = true; fb = (uint8_t)(((fb | 2u) & ~12u) | bounds_class_bits(mx, mn)); }
                    if (proceed) {
                        cha |= na; chb |= nb;
                        const uint32_t nset = (cha ? 1u : 0u) + (chb ? 1u : 0u);
                        d_steps += nset;
                        if (nset) O.d_h5++;
                        if (sw) { if (chb) emit(k2); if (cha) emit(k1); }
                        else { if (cha) emit(k1); if (chb) emit(k2); }
                    }
                }
                wva = k1; wfa = fa; wa = fa != fa_in || xa_w;
                wvb = k2; wfb = fb; wb = fb != fb_in || xb_w;
                // R7 / R8 (:1235-1348) in reach: every non-unique variable is_known -> the general executor decides
                const bool nua = !(fa & 1), nub = !(fb & 1);
                if ((nua || nub) && !((nua && !(fa & 2)) || (nub && !(fb & 2)))) { slow = true; reason = 4; }
            }
        } else if (xy) {
            if (((fa | fb) & 8u) || k1 == k2 || nE != 2) { slow = true; reason = 3; }
            else {
                const bool sw = (shape & SH_R56_SWAP) != 0;          // C order starts with k2
                const uint8_t fa_in = fa, fb_in = fb;
                // R1 (:827-873)
                if (((fa ^ fb) & 1u)) {
                    if (!(fa & 1)) { fa |= 3; emit(k1); } else { fb |= 3; emit(k2); }
                    d_nuniq++; d_steps++; d_h0++;
                }
                // R4 (:991-1076), l == 2: the row is negated on every visit, the pivot alternates
                {
                    flip_new = (uint8_t)(flip_in ^ 1);
                    flip_w = true;
                    const uint32_t new_key = flip_new ? kneg : kpos;
                    const bool n_is_a = new_key == k1;
                    uint8_t fn = n_is_a ? fa : fb, fo_ = n_is_a ? fb : fa;
                    if (fo_ & 4) {
                        if (!(fn & 4)) {
                            fn = (uint8_t)((fn & ~12u) | 4u | 2u);
                            if (n_is_a) a01 = true; else b01 = true;
                            d_steps++; d_h3++;
                            emit(new_key);
                        }
                        if ((fn & 1) && !(fo_ & 1)) {
                            fo_ |= 3;
                            d_nuniq++; d_steps++; d_h3++;
                            emit(n_is_a ? k2 : k1);
                        }
                    }
                    if (n_is_a) { fa = fn; fb = fo_; } else { fb = fn; fa = fo_; }
                }
                // R5 (:1078-1146): bounds are [0,1] or [0,p-1] here, equal iff the class bits agree
                if (((fa ^ fb) & 4u) || ((fa ^ fb) & 1u)) {
                    bool cha = false, chb = false;
                    if ((fa ^ fb) & 1u) { fa |= 3; d_nuniq += 2; cha = chb = true; }        // key_1 written twice (sic, :1107-1108)
                    const bool na = ((fa ^ fb) & 4u) && !(fa & 4u), nb = ((fa ^ fb) & 4u) && !(fb & 4u);
                    if (na) { fa = (uint8_t)((fa & ~12u) | 4u | 2u); a01 = true; }
                    if (nb) { fb = (uint8_t)((fb & ~12u) | 4u | 2u); b01 = true; }
                    cha |= na; chb |= nb;
                    const uint32_t nset = (cha ? 1u : 0u) + (chb ? 1u : 0u);
                    d_steps += nset;
                    if (nset) d_h4++;
                    if (sw) { if (chb) emit(k2); if (cha) emit(k1); }
                    else { if (cha) emit(k1); if (chb) emit(k2); }
                }
                wva = k1; wfa = fa; wa = fa != fa_in || a01;
                wvb = k2; wfb = fb; wb = fb != fb_in || b01;
                // R7 / R8 (:1235-1348) in reach (see chain.hip.hpp): the general executor decides
                const bool nua = !(fa & 1), nub = !(fb & 1);
                if ((nua || nub) && !((nua && (fa & 18u) != 18u) || (nub && (fb & 18u) != 18u))) { slow = true; reason = 4; }
            }
        } else {
            // products and plain sums: R1 (:827-873)
            bool nuab = false, notknown = false;
            uint32_t cnt = 0, u = 0;
            uint8_t uf = 0;
#pragma unroll
            for (uint32_t e = 0; e < 15; ++e) {
                if (e >= nE) continue;
                const uint8_t f = fl[e];
                if (e < nA + nB) nuab |= !(f & 1);
                else if (!(f & 1)) { if (!cnt) { u = w[1 + e]; uf = f; } ++cnt; if (!(f & 2)) notknown = true; }
            }
            if (bigsum) { if (!(cnt >= 2 && notknown)) { slow = true; reason = 0; } }
            else if (!nuab && cnt == 1) {
                wva = u; wfa = (uint8_t)(uf | 3); wa = true;
                emit(u);
                d_nuniq = 1; d_steps = 1; d_h0 = 1;
            } else if ((f4 || r4s) && cnt > 0 && !notknown) { slow = true; reason = 5; }       // R7 / R8 in reach
            if (r4s && !slow) {
                // R4 checkBinary (:991-1076) on a decomposition of 3..15 terms: nothing happens while some variable other than
                // the pivot lacks bounds exactly [0,1] (:1020-1029) -- the usual state of such a row's pops; otherwise the
                // general executor does the arithmetic
                const uint32_t pivot = (shape & SH_R4_T) ? kpos : kneg;
                bool bad = false;
#pragma unroll
                for (uint32_t e = 0; e < 15; ++e)
                    if (e < nE && w[1 + e] != pivot && !(fl[e] & 4)) bad = true;
                if (!bad) { slow = true; reason = 1; }
            }
        }
    }
}

// what fast_decide() decided, written to the state through the Job's own (generic) pointers; one lane. (The fast round
// commits its prefix with ds_write / global stores of its own; this is for the colder callers.)
__device__ __forceinline__ void fast_commit(const Job& J, const FastOut& D, uint32_t row, uint32_t rx, uint32_t validx, LaneCtr& C) {
    if (D.wa) J.flags[D.wva] = D.wfa;
    if (D.wb) J.flags[D.wvb] = D.wfb;
    if (D.a01) { st256(J.lb + 4ull * D.wva, fp::make(0)); st256(J.ub + 4ull * D.wva, fp::make(1)); }
    if (D.b01) { st256(J.lb + 4ull * D.wvb, fp::make(0)); st256(J.ub + 4ull * D.wvb, fp::make(1)); }
    if (D.xa_w) { st256(J.lb + 4ull * D.wva, D.xlb0); st256(J.ub + 4ull * D.wva, D.xub0); }
    if (D.xb_w) { st256(J.lb + 4ull * D.wvb, D.xlb1); st256(J.ub + 4ull * D.wvb, D.xub1); }
    if (D.r2) {        // make_values (:921-927)
        st256(J.values + 8ull * rx, ld256(J.vals + 4ull * validx));
        st256(J.values + 8ull * rx + 4, ld256(J.vals + 4ull * (validx + 1)));
        J.nvalues[rx] = 2;
        J.abz[rx] = -1;
        J.solved[row] = 1;
    }
    if (D.r3v) { st256(J.values + 8ull * D.wva, D.xlb0); J.nvalues[D.wva] = 1; }
    if (D.v6a) { st256(J.values + 8ull * D.wva, D.xub0); st256(J.values + 8ull * D.wva + 4, D.xlb0); J.nvalues[D.wva] = 2; }
    if (D.v6b) { st256(J.values + 8ull * D.wvb, D.xub1); st256(J.values + 8ull * D.wvb + 4, D.xlb1); J.nvalues[D.wvb] = 2; }
    if (D.flip_w) J.flip3[row] = D.flip_new;
    C.steps += D.d_steps; C.nuniq += D.d_nuniq;
    C.hits[0] += D.d_h0; C.hits[1] += D.d_h1; C.hits[2] += D.d_h2; C.hits[3] += D.d_h3; C.hits[4] += D.d_h4; C.hits[5] += D.d_h5;
}

// ---- one long row (no record: more than 15 terms), walked by a whole wavefront, lanes across its entries, four strides per
// trip (the loads of a trip are in flight together: a 1 025-term sum is 5 dependent round trips instead of 17). Gathers what
// R1 asks (:827-873) -- A and B unique? how many non-unique variables in C, the first of them -- whether one of those is not
// is_known (R7 / R8 out of reach, :1235-1348) and, for a binary decomposition (pivot != 0xFFFFFFFF), whether some variable
// other than the pivot lacks bounds exactly [0,1] (R4 does nothing then, :1020-1029).
//
// Watched terms. Most pops of a long linear row do nothing, for a reason two of its variables witness: two non-unique ones
// (R1 wants exactly one), one of them not is_known (R7 / R8 want all of them known) and -- binary decomposition -- that same
// one, not the pivot, without bounds [0,1] (R4 wants every bit bounded). The walk leaves such a pair in words 1 and 2 of the
// row's (otherwise unused) record line, rec[16 row + 1 / 2]: variable ids, 0xFFFFFFFF = none, 0xFFFFFFFE in word 1 = every
// term of a plain sum is unique, for good. While the pair still looks like that a pop of the row is settled from two flag
// bytes, at any rank of a round and in the chain executor's loop (long_watch_holds); the LAST qualifying term is the one
// watched (the bits of a Num2Bits row get bounded in ascending order, one pop of the row each). A cache, not state: reset by
// every solve's setup. watch: 0 = leave the words alone, 1 = plain sum, 2 = binary decomposition.
// All 64 lanes, row0 / pivot wave-uniform; results wave-uniform.
struct LongWalk { uint32_t nnz, cnt, u, uf; bool nuab, notknown, bad; };
__device__ __forceinline__ bool long_watch_holds(uint8_t f0, uint8_t f1, bool r4) {
    return !(f0 & 1) && !(f0 & 2) && !(f1 & 1) && (!r4 || !(f0 & 4));
}
template <bool LDS>
__device__ __noinline__ void long_row_walk(const Job& J, uint32_t row0, uint32_t pivot, uint32_t watch, LongWalk& R) {
    const int lane = lane_id();
    uint8_t* const Fl = (uint8_t*)(ecne_dyn_lds + (LDS ? J.lds_flags_off : 0u));
    const ECNE_GLOBAL uint8_t* const Fg = as_global(J.flags);
    auto ldF = [&](uint32_t v) -> uint8_t { if constexpr (LDS) return Fl[v]; else return Fg[v]; };
    const ECNE_GLOBAL uint32_t* const rpA = as_global(J.rpA); const ECNE_GLOBAL uint32_t* const rpB = as_global(J.rpB);
    const ECNE_GLOBAL uint32_t* const rpC = as_global(J.rpC);
    const ECNE_GLOBAL uint32_t* const cA = as_global(J.colA); const ECNE_GLOBAL uint32_t* const cB = as_global(J.colB);
    const ECNE_GLOBAL uint32_t* const cC = as_global(J.colC);
    const uint32_t a0 = rpA[row0], a1 = rpA[row0 + 1], b0 = rpB[row0], b1 = rpB[row0 + 1], c0 = rpC[row0], c1 = rpC[row0 + 1];
    bool nu = false;
    for (uint32_t k = a0 + (uint32_t)lane; k < a1; k += 64) nu |= !(ldF(cA[k]) & 1);
    for (uint32_t k = b0 + (uint32_t)lane; k < b1; k += 64) nu |= !(ldF(cB[k]) & 1);
    const bool nuab = __ballot(nu) != 0;
    uint32_t cnt = 0, u = 0, uf = 0;
    uint32_t v_nu2 = 0xFFFFFFFFu, v_w = 0xFFFFFFFFu;      // the second non-unique variable (u is the first); the LAST one fit to be watched
    bool nk = false, bd = false;
    for (uint32_t base = c0; base < c1; base += 256) {
        uint32_t v4[4];
        uint8_t f4_[4];
        bool act4[4];
#pragma unroll
        for (uint32_t t = 0; t < 4; ++t) { const uint32_t k = base + 64u * t + (uint32_t)lane; act4[t] = k < c1; v4[t] = act4[t] ? cC[k] : 1u; }
#pragma unroll
        for (uint32_t t = 0; t < 4; ++t) f4_[t] = act4[t] ? ldF(v4[t]) : (uint8_t)7;
#pragma unroll
        for (uint32_t t = 0; t < 4; ++t) {
            const bool nun = act4[t] && !(f4_[t] & 1);
            const uint64_t m = __ballot(nun);
            const uint64_t mk = __ballot(nun && !(f4_[t] & 2));
            const uint64_t mw = __ballot(nun && !(f4_[t] & 2) && (pivot == 0xFFFFFFFFu || (v4[t] != pivot && !(f4_[t] & 4))));
            if (m) {
                uint64_t mm = m;
                if (cnt == 0) { const int src = __ffsll((long long)mm) - 1; u = rdlane(v4[t], (uint32_t)src); uf = rdlane(f4_[t], (uint32_t)src); mm &= mm - 1; }
                if (mm && v_nu2 == 0xFFFFFFFFu) v_nu2 = rdlane(v4[t], (uint32_t)(__ffsll((long long)mm) - 1));
            }
            if (mw) v_w = rdlane(v4[t], (uint32_t)(63 - __clzll((long long)mw)));
            cnt += (uint32_t)__popcll(m);
            nk |= mk != 0;
            bd |= act4[t] && v4[t] != pivot && !(f4_[t] & 4);
        }
    }
    const bool notknown = __ballot(nk) != 0;
    if (watch && lane == 0 && J.rec != nullptr) {
        ECNE_GLOBAL uint32_t* const hw = as_global(const_cast<uint32_t*>(J.rec)) + 16ull * row0;
        if (cnt >= 2 && v_w != 0xFFFFFFFFu) { hw[1] = v_w; hw[2] = v_w == u ? v_nu2 : u; }
        else hw[1] = 0xFFFFFFFFu;
        if (watch == 1 && (cnt == 0 || (cnt == 1 && !nuab))) hw[1] = 0xFFFFFFFEu;      // (after the R1 the caller performs:) every term unique, for good
        if (watch == 2 && cnt == 0) hw[1] = 0xFFFFFFFEu;                                // a decomposition all of whose terms are unique (long_r4_done)
    }
    R.nnz = (a1 - a0) + (b1 - b0) + (c1 - c0);
    R.cnt = cnt; R.u = u; R.uf = uf; R.nuab = nuab; R.notknown = notknown;
    R.bad = pivot != 0xFFFFFFFFu && __ballot(bd) != 0;
}
// a long row the wavefront walk covers: a plain sum or product, or a binary decomposition (R4 shape, l > 2)
__device__ __forceinline__ bool long_r4(uint32_t shape) {
    return !(shape & (SH_HAS_AB | SH_C_EMPTY | SH_R3 | SH_R5 | SH_R6)) && (((shape & SH_R4_T) != 0) != ((shape & SH_R4_T2) != 0));
}
// A long binary decomposition (long_r4) all of whose terms are unique -- word 1 of its record line (h0) says so, for good: exec_row() and
// long_row_walk() leave 0xFFFFFFFE there -- is popped without effect: R1, R7 and R8 (:827-873, :1235-1348) find no non-unique variable and
// R4 (:991-1076) only has the pivot's bounds left to cut, which it does iff ub(pivot) > 2^(l-1) - 1 (:1035), whatever the other bounds are
// (its second half, :1049-1067, makes variables unique that already are). The pop has then read ub(pivot): the caller treats the
// pivot as read. Secp256k1's 39 decompositions of 87-89 bits are popped ~140 times in that state, 11 us each through the general executor.
__device__ __forceinline__ uint32_t long_r4_pivot(uint32_t shape, uint32_t kpos, uint32_t kneg) { return (shape & SH_R4_T) ? kpos : kneg; }
// (out of line: its callers -- the policy, the crew rounds -- are at their register limits, and this is the rare path)
__device__ __noinline__ bool long_r4_done(const Job& J, uint32_t shape, uint32_t kpos, uint32_t kneg, uint32_t lenC, uint32_t h0) {
    if (h0 != 0xFFFFFFFEu || !long_r4(shape) || (J.lv_off & 4u)) return false;      // (lv_off bit 2: ECNE_R4DONE=0, A/B runs)
    const uint32_t l = lenC;
    if (l == 0 || l - 1 >= 254) return l != 0;
    const fp::u256 nub = ld256(J.ub + 4ull * long_r4_pivot(shape, kpos, kneg));
    fp::u256 ip = fp::make(0), im1;
    ip.w[(l - 1) >> 6] = 1ull << ((l - 1) & 63);
    fp::sub_raw(im1, ip, fp::make(1));
    return fp::cmp(nub, im1) <= 0;
}

// A long binary decomposition whose pivot AND lowest bit (the two terms with coefficients 1 and -1: kpos / kneg) are both not unique is
// popped without effect when (gp, gb: their flag bytes)
//   R1 (:827-873)   two terms are not unique: nothing;
//   R4 (:991-1076)  bit 0 lacks bounds [0,1]: the loop at :1020-1029 returns; else the pivot's bounds are cut iff ub(pivot) > 2^(l-1) - 1
//                   (:1035) -- nothing when it is not; the second half (:1049-1067) wants the pivot unique;
//   R7 (:1235-1298) a term that is not unique and not known: returns (:1252) -- else the two smallest |coefficients| are the pivot's and
//                   bit 0's, both 1: quotient 1, remainder 0, and the rule returns iff 1 <= ub - lb of the first of them (:1262-1266):
//                   bit 0's bounds are [0,1] (gb bit 2), the pivot's are looked at;
//   R8 (:1304-1348) a term that is not unique and carries no group tag: returns.
// secp256k1's 39 decompositions are popped ~60 times in that state (between the pop that cuts the pivot's bounds and the one that finds
// the pivot unique), 11 us each through the general executor and each at the head of the queue, by the whole workgroup. The pop has
// read the state of the pivot and of bit 0: the caller treats both as read. Exact for any state: false = "ask the general executor".
// (f_pos / f_neg: the flag bytes of kpos / kneg; which of the two is the pivot depends on the row's orientation: SH_R4_T -- the pivot is the
//  term with coefficient 1 --, SH_R4_T2 -- the reference negates the row at its first visit, :1001-1011, and the -1 term is the pivot)
__device__ __noinline__ bool long_r4_idle(const Job& J, uint32_t shape, uint32_t f_pos, uint32_t f_neg, uint32_t kpos, uint32_t kneg, uint32_t lenC) {
    const bool t1 = (shape & SH_R4_T) != 0;
    const uint32_t gp = t1 ? f_pos : f_neg, gb = t1 ? f_neg : f_pos, pivot = t1 ? kpos : kneg;
    if (((gp | gb) & 1u) || ((gp & gb) & 16u) || (J.lv_off & 4u)) return false;
    const uint32_t l = lenC;
    if (l < 3 || l - 1 >= 254) return false;
    const bool r7_unknown = !((gp & gb) & 2u);
    if (!(gb & 4u)) return r7_unknown;
    const fp::u256 nub = ld256(J.ub + 4ull * pivot);
    fp::u256 ip = fp::make(0), im1;
    ip.w[(l - 1) >> 6] = 1ull << ((l - 1) & 63);
    fp::sub_raw(im1, ip, fp::make(1));
    if (fp::cmp(nub, im1) > 0) return false;
    if (r7_unknown) return true;
    const fp::u256 nlb = ld256(J.lb + 4ull * pivot);
    return fp::cmp(nub, nlb) > 0;
}

// The two pops of a long binary decomposition that DO something -- both through R4 (:991-1076) alone -- on one wavefront, without the
// general executor's R1 / R7 / R8 passes (exec_row(): 18 and 14 us on an 88-bit row, secp256k1 has 39 of each):
//   (a) pivot and lowest bit not unique, every bit bounded [0,1]: R4 cuts the pivot's bounds to [0, 2^(l-1) - 1] iff ub(pivot) is above
//       (:1031-1048; is_known, one step, REQUEUE(pivot)); R1 (two terms not unique), R7 and R8 do nothing for long_r4_idle()'s reasons --
//       R7 sees the pivot known and bounded [0, 2^(l-1) - 1] after the cut;
//   (b) pivot unique, at least two bits not unique (one: R1's, :827-873), every bit bounded: the cut as in (a), then every bit that is not
//       unique becomes unique, in the row's order, one step and one REQUEUE each (:1049-1067); nothing is left for R7 / R8.
// Anything else: false, nothing written -- the general executor's. All 64 lanes; q.emit as exec_row().
__device__ __noinline__ bool exec_long_r4(const Job& J, QState& q, uint32_t row, const RowInfo& ri, unsigned long long* hits,
                                          unsigned long long& steps, unsigned long long& nuniq) {
    const int lane = lane_id();
    const uint32_t shape = ri.shape, l = ri.lenC;
    if (!long_r4(shape) || l < 3 || l - 1 >= 254 || ri.validx == 0xFFFFFFFFu) return false;
    const bool t1 = (shape & SH_R4_T) != 0;
    const uint32_t piv = t1 ? ri.kpos : ri.kneg, bit0 = t1 ? ri.kneg : ri.kpos;
    const uint32_t gp = J.flags[piv], gb = J.flags[bit0];
    const uint32_t c0 = J.rpC[row], c1 = J.rpC[row + 1];
    bool bad = false, nk = false;
    uint32_t cnt = 0;
    for (uint32_t k = c0 + (uint32_t)lane; k < c1; k += 64) {
        const uint32_t v = J.colC[k];
        const uint32_t f = J.flags[v];
        if (v != piv) { bad |= !(f & 4u); cnt += (f & 1u) ? 0u : 1u; nk |= !(f & 3u); }
    }
    if (__ballot(bad)) return false;
    const bool notknown = __ballot(nk) != 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
    const fp::u256 fub = ld256(J.vals + 4ull * (ri.validx + 1));
    const fp::u256 nlb = ld256(J.lb + 4ull * piv), nub = ld256(J.ub + 4ull * piv);
    fp::u256 ip = fp::make(0), im1;
    ip.w[(l - 1) >> 6] = 1ull << ((l - 1) & 63);
    fp::sub_raw(im1, ip, fp::make(1));
    const bool cut = !(fp::is_zero(nlb) && fp::eq(nub, fub)) && fp::cmp(nub, im1) > 0;
    if (gp & 1u) {
        if (cnt < 2) return false;
    } else {
        if ((gb & 1u) || ((gp & gb) & 16u)) return false;
        const bool r7_unknown = notknown || (!(gp & 2u) && !cut);
        if (!r7_unknown && !cut && fp::cmp(nub, nlb) <= 0) return false;
    }
    if (cut) {
        if (lane == 0) { set_bounds(J, piv, fp::make(0), fub); J.flags[piv] |= 2; }
        wg_fence();
        steps++; hits[3]++;
        requeue(J, q, piv);
    }
    if (gp & 1u) {
        const uint32_t n = uniq_range_and_requeue(J, q, c0, c1, piv);
        nuniq += n; steps += n; hits[3] += n;
        if (lane == 0 && J.rec != nullptr && l > 15) const_cast<uint32_t*>(J.rec)[16ull * row + 1] = 0xFFFFFFFEu;      // (as exec_row(): long_r4_done)
    }
    return true;
}

}  // namespace ecne
