// rules_lane.hip.hpp — one pop executed by ONE lane (rows of at most ECNE_SMALL_ROW entries): statement for statement the rules of rules_wave.hip.hpp, plus the register-resident x == y executor.
// Part of the gfx950 device code of libecne_hip (see kernels.hip.hpp for the overview).
#pragma once
#include "wg_tools.hip.hpp"

namespace ecne {

// ================================================================== chunk-parallel queue schedule
// The reference pops one row at a time. Here the first n queue entries ("chunk", ranks
// 0..n-1) are examined together and the longest prefix of pairwise independent rows is executed in
// parallel, one lane per row, directly on the shared state. Two rows are independent when neither
// can write a variable the other reads or writes; the read/write sets are static supersets derived
// from the row shape:
//     non-linear row, C non-empty : reads A u B u C, may write C            (R1)
//     C empty, bit-check shaped   : reads/writes x                          (R2)
//     C empty, anything else      : touches nothing (no rule can fire)
//     linear row                  : reads/writes C                          (R1, R3..R8)
// (the constant wire's `unique`/`is_known` never change, so it only counts for rows flagged
// SH_TOUCH1). Independent rows commute, so executing the prefix in parallel gives exactly the state
// the sequential pops give; the queue itself is then rebuilt in sequential order by resolving all
// REQUEUE events of the prefix in (rank, emission order, fan-out position) order with the reference's
// in_queue semantics. Rows with more than ECNE_SMALL_ROW entries ("long rows") are handled by a whole
// workgroup, inside the round where possible (big_rows_*), alone otherwise.  DESIGN.md "Schedule" has
// the equivalence argument.

struct LaneCtr {   // per-lane counter deltas of one queue phase (reduced at the end)
    uint32_t steps, nuniq, hits[8];
    uint32_t rank;   // absolute queue position of the row being executed (orders the errors, see raise_ranked)
};

__device__ __forceinline__ uint32_t lane_uniq_range(const Job& J, uint32_t c0, uint32_t c1, uint32_t skip,
                                                    uint32_t* ev, uint32_t& nev) {
    uint32_t n = 0;
    for (uint32_t k = c0; k < c1; ++k) {
        uint32_t v = J.colC[k];
        if (v != skip && !(J.flags[v] & 1)) {
            J.flags[v] |= 3;
            ev[nev++] = v;
            ++n;
        }
    }
    return n;
}

// One queue pop executed by ONE lane (rows with at most ECNE_SMALL_ROW entries). Statement-for-
// statement the same rules as exec_row(); REQUEUE(v) becomes an event appended to ev[].
// R7 then R8 of a small row from the state in memory (no statistics carried over from R1): the closing part
// of exec_row_lane(), also used by the x == y fast path when one of the two could fire.
__device__ __noinline__ void lane_r78_tail(const Job& J, uint32_t c0, uint32_t c1, uint32_t shape, uint32_t* ev, uint32_t& nev, LaneCtr& C) {
    const uint32_t l = c1 - c0;
    if (l == 0) return;
    // R7 (:1235-1298)
    {
        uint32_t nunk = 0;
        bool notknown = false;
        for (uint32_t k = c0; k < c1; ++k) {
            uint8_t f = J.flags[J.colC[k]];
            if (!(f & 1)) { ++nunk; if (!(f & 2)) notknown = true; }
        }
        if (nunk > 0 && !notknown) {
            const bool negated = (shape & SH_R4_T2) && !(shape & SH_R4_T);
            bool fail = false;
            uint32_t prev_k = 0xFFFFFFFFu;
            for (uint32_t s = 0; s < l && !fail; ++s) {
                uint32_t k = c0 + J.csort[c0 + s];
                uint32_t v = J.colC[k];
                if (J.flags[v] & 1) continue;
                if (prev_k != 0xFFFFFFFFu && r7_link_fails(J, k, prev_k, negated)) fail = true;
                prev_k = k;
            }
            if (!fail && r7_top_fits(J, prev_k, negated)) {
                C.steps += nunk; C.hits[6]++;
                C.nuniq += lane_uniq_range(J, c0, c1, 0xFFFFFFFFu, ev, nev);
            }
        }
    }
    // R8 (:1304-1348)
    {
        int group = -1;
        bool bad = false;
        uint32_t cnt = 0;
        for (uint32_t k = c0; k < c1 && !bad; ++k) {
            uint32_t v = J.colC[k];
            if (J.flags[v] & 1) continue;
            int a = J.abz[v];
            if (a == -1) bad = true;
            else if (group == -1) group = a;
            else if (a != group) bad = true;
            ++cnt;
        }
        if (cnt > 0 && !bad) {
            C.hits[7]++;
            uint32_t n = lane_uniq_range(J, c0, c1, 0xFFFFFFFFu, ev, nev);
            C.nuniq += n; C.steps += n;
        }
    }
}

// One pop of a plain x == y row (C = {k1: +-1, k2: -+1}, nothing else; the bulk of an --O0 circuit) on one
// lane: R1, R4 and R5 of exec_row_lane() on a register copy of the two variables' state -- one batch of
// loads, one batch of stores -- instead of a memory round trip per rule step. R7 / R8 run from memory
// afterwards (lane_r78_tail) in the rare case their preconditions hold. Statement for statement the same
// effects, counters and REQUEUE events as the general executor.
__device__ __noinline__ void exec_xy_lane(const Job& J, uint32_t row, const RowInfo& ri, uint32_t* ev, uint32_t& nev, LaneCtr& C) {
    const uint32_t shape = ri.shape;
    const uint32_t kv[2] = {ri.k1, ri.k2};                       // dictionary order (R5's key_1, key_2)
    const bool sw = (shape & SH_R56_SWAP) != 0;                  // the Set / stored order of C starts with k2
    const int o0 = sw ? 1 : 0, o1 = sw ? 0 : 1;                  // indices in C order
    uint8_t f[2] = {J.flags[kv[0]], J.flags[kv[1]]};
    const int ab[2] = {J.abz[kv[0]], J.abz[kv[1]]};
    fp::u256 lb[2] = {ld256(J.lb + 4ull * kv[0]), ld256(J.lb + 4ull * kv[1])};
    fp::u256 ub[2] = {ld256(J.ub + 4ull * kv[0]), ld256(J.ub + 4ull * kv[1])};
    const uint8_t flip = J.flip3[row];
    const uint8_t f_in[2] = {f[0], f[1]};
    bool bdirty[2] = {false, false};
    auto set_b = [&](int i, const fp::u256& nlb, const fp::u256& nub) {   // set_bounds()
        lb[i] = nlb; ub[i] = nub; bdirty[i] = true;
        f[i] = (uint8_t)((f[i] & ~12u) | bounds_class_bits(nlb, nub));
    };
    // R1 (:827-873): no A / B; exactly one non-unique variable of C becomes unique
    {
        const uint32_t cnt = (uint32_t)!(f[o0] & 1) + (uint32_t)!(f[o1] & 1);
        if (cnt == 1) {
            const int u = !(f[o0] & 1) ? o0 : o1;
            f[u] |= 3;
            C.nuniq++; C.steps++; C.hits[0]++;
            ev[nev++] = kv[u];
        }
    }
    // R4 (:991-1076) with l == 2: the row is negated on every visit, the pivot alternates
    {
        const uint8_t o = (uint8_t)(flip ^ 1);
        J.flip3[row] = o;
        const uint32_t new_key = o ? ri.kneg : ri.kpos;
        const int n = new_key == kv[0] ? 0 : 1, ot = 1 - n;
        if (f[ot] & 4) {                                          // the other variable has bounds exactly [0,1]
            if (!(fp::is_zero(lb[n]) && fp::is_one(ub[n]))) {
                if (fp::cmp(ub[n], fp::make(1)) > 0) {             // ub.d > 2^(l-1) - 1
                    set_b(n, fp::make(0), fp::make(1));
                    f[n] |= 2;
                    C.steps++; C.hits[3]++;
                    ev[nev++] = new_key;
                }
            }
            if (f[n] & 1) {                                       // pivot unique: the others become unique, C order
                for (int t = 0; t < 2; ++t) {
                    const int i = t == 0 ? o0 : o1;
                    if (i != n && !(f[i] & 1)) { f[i] |= 3; ev[nev++] = kv[i]; C.nuniq++; C.steps++; C.hits[3]++; }
                }
            }
        }
    }
    // R5 (:1078-1146)
    if (!fp::eq(ub[1], ub[0]) || !fp::eq(lb[1], lb[0]) || ((f[0] ^ f[1]) & 1)) {
        bool ch0 = false, ch1 = false;
        if ((f[0] ^ f[1]) & 1) { f[0] |= 3; C.nuniq += 2; ch0 = ch1 = true; }   // key_1 written twice (sic)
        const fp::u256 mn = fp::cmp(ub[0], ub[1]) <= 0 ? ub[0] : ub[1];
        const fp::u256 mx = fp::cmp(lb[0], lb[1]) >= 0 ? lb[0] : lb[1];
        const bool w0 = fp::cmp(ub[0], mn) > 0 || fp::cmp(lb[0], mx) < 0;
        const bool w1 = fp::cmp(ub[1], mn) > 0 || fp::cmp(lb[1], mx) < 0;
        if (w0) { f[0] |= 2; set_b(0, mx, mn); }
        if (w1) { f[1] |= 2; set_b(1, mx, mn); }
        ch0 |= w0; ch1 |= w1;
        const uint32_t nset = (ch0 ? 1u : 0u) + (ch1 ? 1u : 0u);
        C.steps += nset;
        if (nset) C.hits[4]++;
        if (sw) { if (ch1) ev[nev++] = kv[1]; if (ch0) ev[nev++] = kv[0]; }
        else { if (ch0) ev[nev++] = kv[0]; if (ch1) ev[nev++] = kv[1]; }
    }
    // write back what changed
    for (int i = 0; i < 2; ++i) {
        if (bdirty[i]) { st256(J.lb + 4ull * kv[i], lb[i]); st256(J.ub + 4ull * kv[i], ub[i]); }
        if (f[i] != f_in[i]) J.flags[kv[i]] = f[i];
    }
    // R7 / R8 (:1235-1348): only when one of them could fire
    {
        const bool nu0 = !(f[o0] & 1), nu1 = !(f[o1] & 1);
        if (nu0 || nu1) {
            const bool notknown = (nu0 && !(f[o0] & 2)) || (nu1 && !(f[o1] & 2));
            bool badgroup;
            if (nu0 && nu1) badgroup = ab[o0] == -1 || ab[o1] != ab[o0];
            else badgroup = (nu0 ? ab[o0] : ab[o1]) == -1;
            if (!notknown || !badgroup) {
                const uint32_t c0 = J.rpC[row];
                lane_r78_tail(J, c0, c0 + 2, shape, ev, nev, C);
            }
        }
    }
}

__device__ __noinline__ void exec_row_lane(const Job& J, uint32_t row, uint32_t* ev, uint32_t& nev, LaneCtr& C) {
    const RowInfo ri = J.rinfo[row];
    if ((ri.shape & (SH_R5 | SH_R4_T | SH_R4_T2 | SH_R3)) == (SH_R5 | SH_R4_T | SH_R4_T2)) { exec_xy_lane(J, row, ri, ev, nev, C); return; }
    const uint32_t a0 = J.rpA[row], a1 = J.rpA[row + 1];
    const uint32_t b0 = J.rpB[row], b1 = J.rpB[row + 1];
    const uint32_t c0 = J.rpC[row], c1 = J.rpC[row + 1];
    const uint32_t shape = ri.shape;
    bool st_valid = false, st_notknown = false, st_badgroup = false;
    uint32_t st_cnt = 0;
    int st_group = -2;
    // R1 (:827-873). Entries are fetched four per part at a time -- ids, then flag bytes and group tags --
    // so that the lane waits per batch, not per entry (the constant wire pads the short parts).
    {
        bool nu = false;
        uint32_t cnt = 0, u = 0;
        const bool lin = !(shape & SH_HAS_AB);
        uint32_t n = a1 - a0;
        n = b1 - b0 > n ? b1 - b0 : n;
        n = c1 - c0 > n ? c1 - c0 : n;
        for (uint32_t off = 0; off < n && !nu; off += 4) {
            uint32_t v[12];
            uint8_t fl[12];
            int ab[4];
#pragma unroll
            for (uint32_t i = 0; i < 4; ++i) {
                v[i] = a0 + off + i < a1 ? J.colA[a0 + off + i] : 1u;
                v[4 + i] = b0 + off + i < b1 ? J.colB[b0 + off + i] : 1u;
                v[8 + i] = c0 + off + i < c1 ? J.colC[c0 + off + i] : 1u;
            }
#pragma unroll
            for (uint32_t i = 0; i < 12; ++i) fl[i] = J.flags[v[i]];
#pragma unroll
            for (uint32_t i = 0; i < 4; ++i) ab[i] = lin ? J.abz[v[8 + i]] : -1;
#pragma unroll
            for (uint32_t i = 0; i < 8; ++i) nu |= !(fl[i] & 1);
#pragma unroll
            for (uint32_t i = 0; i < 4; ++i) {
                if (c0 + off + i >= c1) continue;
                const uint8_t f = fl[8 + i];
                if (f & 1) continue;
                if (cnt == 0) u = v[8 + i];
                ++cnt;
                if (lin) {   // the same walk collects what R7 and R8 ask of C's non-unique variables
                    if (!(f & 2)) st_notknown = true;
                    if (st_group == -2) st_group = ab[i];
                    if (ab[i] == -1 || ab[i] != st_group) st_badgroup = true;
                }
            }
        }
        if (!nu) {
            st_cnt = cnt;
            st_valid = lin;
            if (cnt == 1) {
                J.flags[u] |= 3;
                C.nuniq++; C.steps++; C.hits[0]++;
                ev[nev++] = u;
                st_valid = false;
            }
        }
    }
    const uint32_t steps_at_r1 = C.steps, nuniq_at_r1 = C.nuniq;
    // R2 (:875-942)
    if (shape & SH_C_EMPTY) {
        if (shape & SH_R2_BOUNDSERR) { raise_ranked(J, C.rank, K_EBOUNDS); return; }
        if (shape & SH_R2) {
            const uint32_t x = ri.x;
            if (!(J.flags[x] & 2)) {
                if (shape & SH_R2_DIV0) { raise_ranked(J, C.rank, K_EDIVZERO); return; }
                st256(J.values + 8ull * x, ld256(J.vals + 4ull * ri.validx));
                st256(J.values + 8ull * x + 4, ld256(J.vals + 4ull * (ri.validx + 1)));
                J.nvalues[x] = 2;
                J.flags[x] = (uint8_t)((J.flags[x] | 2) & ~16u);   // is_known; the group tag is gone (bit 4)
                J.abz[x] = -1;
                if (shape & SH_R2_IS01) set_bounds(J, x, fp::make(0), fp::make(1));
                J.solved[row] = 1;
                ev[nev++] = x;
                C.steps++; C.hits[1]++;
            }
        }
    }
    if (shape & SH_HAS_AB) return;
    const uint32_t l = c1 - c0;
    // R3 (:949-988)
    if (shape & SH_R3) {
        const uint32_t x = ri.x;
        const fp::u256 tv = ld256(J.vals + 4ull * ri.validx);
        bool new_info = false;
        const bool same = J.nvalues[x] == 1 && fp::eq(ld256(J.values + 8ull * x), tv);
        const uint8_t f = J.flags[x];
        if (!same) { st256(J.values + 8ull * x, tv); J.nvalues[x] = 1; C.steps++; C.hits[2]++; new_info = true; }
        if (!(f & 1)) { C.nuniq++; new_info = true; }
        J.flags[x] = (uint8_t)(f | 3);
        set_bounds(J, x, tv, tv);
        if (new_info) ev[nev++] = x;
    }
    // R4 (:991-1076)
    if ((shape & (SH_R4_T | SH_R4_T2)) && l > 0) {
        uint32_t new_key;
        if ((shape & SH_R4_T) && (shape & SH_R4_T2)) {
            uint8_t o = (uint8_t)(J.flip3[row] ^ 1);
            J.flip3[row] = o;
            new_key = o ? ri.kneg : ri.kpos;
        } else if (shape & SH_R4_T2) new_key = ri.kneg;
        else new_key = ri.kpos;
        bool bad = false;
        for (uint32_t base = c0; base < c1 && !bad; base += 4) {
            uint32_t v[4];
            uint8_t fl[4];
#pragma unroll
            for (uint32_t i = 0; i < 4; ++i) v[i] = base + i < c1 ? J.colC[base + i] : new_key;
#pragma unroll
            for (uint32_t i = 0; i < 4; ++i) fl[i] = J.flags[v[i]];
#pragma unroll
            for (uint32_t i = 0; i < 4; ++i) if (v[i] != new_key && !(fl[i] & 4)) bad = true;
        }
        if (!bad) {
            const fp::u256 fub = ld256(J.vals + 4ull * (ri.validx + 1));
            const fp::u256 nlb = ld256(J.lb + 4ull * new_key), nub = ld256(J.ub + 4ull * new_key);
            if (!(fp::is_zero(nlb) && fp::eq(nub, fub))) {
                bool gt = false;
                if (l - 1 < 254) {
                    fp::u256 ip = fp::make(0);
                    ip.w[(l - 1) >> 6] = 1ull << ((l - 1) & 63);
                    fp::u256 im1;
                    fp::sub_raw(im1, ip, fp::make(1));
                    gt = fp::cmp(nub, im1) > 0;
                }
                if (gt) {
                    set_bounds(J, new_key, fp::make(0), fub);
                    J.flags[new_key] |= 2;
                    C.steps++; C.hits[3]++;
                    ev[nev++] = new_key;
                }
            }
            if (J.flags[new_key] & 1) {
                uint32_t n = lane_uniq_range(J, c0, c1, new_key, ev, nev);
                C.nuniq += n; C.steps += n; C.hits[3] += n;
            }
        }
    }
    // R5 / R6 (:1078-1232)
    if (shape & (SH_R5 | SH_R6)) {
        const bool is6 = (shape & SH_R6) != 0;
        const uint32_t k1 = ri.k1, k2 = ri.k2;
        fp::u256 lb1 = ld256(J.lb + 4ull * k1), ub1 = ld256(J.ub + 4ull * k1);
        fp::u256 lb2 = ld256(J.lb + 4ull * k2), ub2 = ld256(J.ub + 4ull * k2);
        uint8_t f1 = J.flags[k1], f2 = J.flags[k2];
        bool ch1 = false, ch2 = false;
        if (!fp::eq(ub2, ub1) || !fp::eq(lb2, lb1) || ((f1 ^ f2) & 1)) {
            bool proceed = true;
            if ((f1 ^ f2) & 1) {
                f1 |= 3;
                if (is6) f2 |= 3;
                C.nuniq += 2;
                ch1 = ch2 = true;
            }
            fp::u256 mn = fp::cmp(ub1, ub2) <= 0 ? ub1 : ub2;
            fp::u256 mx = fp::cmp(lb1, lb2) >= 0 ? lb1 : lb2;
            if (is6 && (!fp::is_one(mn) || !fp::is_zero(mx))) proceed = false;
            bool w1 = false, w2 = false;
            if (proceed) {
                w1 = fp::cmp(ub1, mn) > 0 || fp::cmp(lb1, mx) < 0;
                w2 = fp::cmp(ub2, mn) > 0 || fp::cmp(lb2, mx) < 0;
            }
            J.flags[k1] = f1;
            J.flags[k2] = f2;
            if (w1) {
                J.flags[k1] |= 2;
                set_bounds(J, k1, mx, mn);
                if (is6) { st256(J.values + 8ull * k1, mn); st256(J.values + 8ull * k1 + 4, mx); J.nvalues[k1] = 2; }
            }
            if (w2) {
                J.flags[k2] |= 2;
                set_bounds(J, k2, mx, mn);
                if (is6) { st256(J.values + 8ull * k2, mn); st256(J.values + 8ull * k2 + 4, mx); J.nvalues[k2] = 2; }
            }
            if (proceed) {
                ch1 |= w1; ch2 |= w2;
                uint32_t nset = (ch1 ? 1u : 0u) + (ch2 ? 1u : 0u);
                C.steps += nset;
                if (nset) C.hits[is6 ? 5 : 4]++;
                const bool sw = (shape & SH_R56_SWAP) != 0;
                uint32_t first = sw ? k2 : k1, second = sw ? k1 : k2;
                bool cf = sw ? ch2 : ch1, cs = sw ? ch1 : ch2;
                if (cf) ev[nev++] = first;
                if (cs) ev[nev++] = second;
            }
        }
    }
    // R7 (:1235-1298)
    if (l > 0) {
        uint32_t nunk = 0;
        bool notknown = false;
        if (st_valid && (C.steps != steps_at_r1 || C.nuniq != nuniq_at_r1)) st_valid = false;   // something fired since R1
        if (st_valid) { nunk = st_cnt; notknown = st_notknown; }
        else
        for (uint32_t k = c0; k < c1; ++k) {
            uint8_t f = J.flags[J.colC[k]];
            if (!(f & 1)) { ++nunk; if (!(f & 2)) notknown = true; }
        }
        if (nunk > 0 && !notknown) {
            const bool negated = (shape & SH_R4_T2) && !(shape & SH_R4_T);
            bool fail = false;
            uint32_t prev_k = 0xFFFFFFFFu;
            for (uint32_t s = 0; s < l && !fail; ++s) {
                uint32_t k = c0 + J.csort[c0 + s];
                uint32_t v = J.colC[k];
                if (J.flags[v] & 1) continue;
                if (prev_k != 0xFFFFFFFFu) {
                    fp::u256 cn = ld256(J.coefC + 4ull * k), cc = ld256(J.coefC + 4ull * prev_k);
                    if (negated) { cn = fp::neg(cn); cc = fp::neg(cc); }
                    cn = r7_abs(cn); cc = r7_abs(cc);
                    fp::u256 qq, rem;
                    fp::divmod(cn, cc, qq, rem);
                    if (!fp::is_zero(rem)) fail = true;
                    else {
                        uint32_t pv = J.colC[prev_k];
                        fp::u256 ub = ld256(J.ub + 4ull * pv), lb = ld256(J.lb + 4ull * pv);
                        if (fp::cmp(ub, lb) >= 0) {
                            fp::u256 diff;
                            fp::sub_raw(diff, ub, lb);
                            if (fp::cmp(qq, diff) <= 0) fail = true;
                        }
                    }
                }
                prev_k = k;
            }
            if (!fail) {
                uint32_t lv = J.colC[prev_k];
                fp::u256 cl = ld256(J.coefC + 4ull * prev_k);
                if (negated) cl = fp::neg(cl);
                cl = r7_abs(cl);
                fp::u256 ub1;
                fp::add_raw(ub1, ld256(J.ub + 4ull * lv), fp::make(1));
                if (!fp::mul_gt_p(cl, ub1)) {
                    C.steps += nunk; C.hits[6]++;
                    C.nuniq += lane_uniq_range(J, c0, c1, 0xFFFFFFFFu, ev, nev);
                }
            }
        }
    }
    // R8 (:1304-1348)
    if (l > 0) {
        int group = -1;
        bool bad = false;
        uint32_t cnt = 0;
        if (st_valid && C.steps == steps_at_r1 && C.nuniq == nuniq_at_r1) { cnt = st_cnt; bad = st_badgroup; }
        else
        for (uint32_t k = c0; k < c1 && !bad; ++k) {
            uint32_t v = J.colC[k];
            if (J.flags[v] & 1) continue;
            int a = J.abz[v];
            if (a == -1) bad = true;
            else if (group == -1) group = a;
            else if (a != group) bad = true;
            ++cnt;
        }
        if (cnt > 0 && !bad) {
            C.hits[7]++;
            uint32_t n = lane_uniq_range(J, c0, c1, 0xFFFFFFFFu, ev, nev);
            C.nuniq += n; C.steps += n;
        }
    }
}

}  // namespace ecne
