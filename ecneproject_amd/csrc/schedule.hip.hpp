// schedule.hip.hpp — the hazard model of the round schedule (access sets, exact no-op test), its LDS state, long rows handled by a whole workgroup, and the ordered REQUEUE resolution.
// Part of the gfx950 device code of libecne_hip (see kernels.hip.hpp for the overview).
#pragma once
#include "rules_lane.hip.hpp"

namespace ecne {

// Access sets of a small row for the conflict test, f(v, read_mask, write_mask) with bit0 = U-class
// (unique / is_known bits) and bit1 = B-class (lb, ub, bounds01 bit, values, abz):
//     non-linear row, C non-empty : reads U of A u B u C, may write U of C                  (R1)
//     C empty, bit-check shaped   : reads U(x), may write U(x) and B(x)                      (R2)
//     C empty, anything else      : touches nothing
//     linear row                  : reads and may write U and B of C                          (R1, R3..R8)
// U-class state of a variable is FINAL once both bits are set (they are only ever set), so U-class
// accesses to such variables are dropped: no row can change them and every reader sees the same value.
// for_row_sets4 also hands out the CONSERVATIVE write mask wrc (4th argument): what the row could write in ANY later state
// reachable before it is popped (finality of U-class state is monotone, so final variables stay dropped) -- the write set of a
// row whose inputs may still change (drain rounds, drain.hip.hpp). wr is a subset of wrc, wrc a subset of what rd | wrc covers.
template <class F>
__device__ __forceinline__ void for_row_sets4(const Job& J, uint32_t row, uint32_t shape, uint32_t x, F f) {
    if (shape & SH_C_EMPTY) {
        if (shape & SH_R2) {
            const bool fin = (J.flags[x] & 3) == 3;
            f(x, fin ? 0u : 1u, fin ? 2u : 3u, fin ? 2u : 3u);
        }
        return;
    }
    const bool lin = !(shape & SH_HAS_AB);
    if (!lin) {
        // Entries are fetched four per part at a time, all variable ids first and all flag bytes second:
        // a lane then waits for two memory round trips per step instead of two per ENTRY (the constant
        // wire, always unique and known, pads the short parts).
        const uint32_t a0 = J.rpA[row], a1 = J.rpA[row + 1], b0 = J.rpB[row], b1 = J.rpB[row + 1];
        const uint32_t c0 = J.rpC[row], c1 = J.rpC[row + 1];
        uint32_t n = a1 - a0;
        n = b1 - b0 > n ? b1 - b0 : n;
        n = c1 - c0 > n ? c1 - c0 : n;
        // Only R1 can write here, and it writes exactly one variable: the single non-unique one of C when
        // every variable of A and B is unique. Decided on the state this row reads -- if an earlier row of
        // the window changes that state the row is blocked anyway, so the observation cannot go stale.
        // Rows that fit one batch (the usual a * b = c) get the exact write set; longer ones the
        // conservative one (any non-final variable of C).
        const bool one_batch = n <= 4;
        for (uint32_t off = 0; off < n; off += 4) {
            uint32_t v[12];
            uint8_t fl[12];
#pragma unroll
            for (uint32_t i = 0; i < 4; ++i) {
                v[i] = a0 + off + i < a1 ? J.colA[a0 + off + i] : 1u;
                v[4 + i] = b0 + off + i < b1 ? J.colB[b0 + off + i] : 1u;
                v[8 + i] = c0 + off + i < c1 ? J.colC[c0 + off + i] : 1u;
            }
#pragma unroll
            for (uint32_t i = 0; i < 12; ++i) fl[i] = J.flags[v[i]];
            bool may_write = true;
            if (one_batch) {
                uint32_t ab = 1, cnt = 0;
#pragma unroll
                for (uint32_t i = 0; i < 8; ++i) ab &= fl[i];
#pragma unroll
                for (uint32_t i = 8; i < 12; ++i) cnt += !(fl[i] & 1);   // (padding is the constant wire: unique)
                may_write = (ab & 1) && cnt == 1;
            }
#pragma unroll
            for (uint32_t i = 0; i < 12; ++i)
                if ((fl[i] & 3) != 3) f(v[i], 1u, (i >= 8 && may_write && !(one_batch && (fl[i] & 1))) ? 1u : 0u, i >= 8 ? 1u : 0u);
        }
        return;
    }
    // linear row. B-class state is only ever WRITTEN by: R3 on x, R4 on its pivot(s), R5/R6 on k1, k2.
    // Where the current state shows that such a write would store what is already there (R3) or
    // would not happen (equal bounds on an x == y / 1 = x + y row), it is not counted as a write;
    // the row still READS that state, so an earlier writer in the chunk blocks it and the
    // observation cannot go stale.
    if ((shape & SH_R5) && !(shape & SH_R3)) {
        // plain x == y row: both variables come from the descriptor, one batch of loads (see row_is_noop)
        const RowInfo ri = J.rinfo[row];
        const uint8_t f1 = J.flags[ri.k1], f2 = J.flags[ri.k2];
        const fp::u256 l1 = ld256(J.lb + 4ull * ri.k1), l2 = ld256(J.lb + 4ull * ri.k2);
        const fp::u256 u1 = ld256(J.ub + 4ull * ri.k1), u2 = ld256(J.ub + 4ull * ri.k2);
        const uint32_t wb = (fp::eq(l1, l2) & fp::eq(u1, u2)) ? 0u : 2u;
        const uint32_t o1 = ((f1 & 3) == 3) ? 0u : 1u, o2 = ((f2 & 3) == 3) ? 0u : 1u;
        f(ri.k1, o1 | 2u, o1 | wb, o1 | 2u);
        f(ri.k2, o2 | 2u, o2 | wb, o2 | 2u);
        return;
    }
    const bool touch1 = (shape & SH_TOUCH1) != 0;
    uint32_t wb0 = 0xFFFFFFFFu, wb1 = 0xFFFFFFFFu, wb2 = 0xFFFFFFFFu;   // variables whose B-state may be written
    uint32_t wc0 = 0xFFFFFFFFu, wc1 = 0xFFFFFFFFu, wc2 = 0xFFFFFFFFu;   // ... in any later state
    fp::u256 tv = fp::make(0);
    if (shape & SH_R3) {
        const RowInfo ri = J.rinfo[row];
        tv = ld256(J.vals + 4ull * ri.validx);
        // (all four loads first: a short-circuit chain would wait for them one after the other)
        const uint8_t nv = J.nvalues[x];
        const fp::u256 va = ld256(J.values + 8ull * x), lbx = ld256(J.lb + 4ull * x), ubx = ld256(J.ub + 4ull * x);
        const bool same = (nv == 1) & fp::eq(va, tv) & fp::eq(lbx, tv) & fp::eq(ubx, tv);
        if (!same) wb0 = x;
        wc0 = x;
    }
    if (shape & (SH_R4_T | SH_R4_T2 | SH_R5 | SH_R6)) {
        const RowInfo ri = J.rinfo[row];
        if (shape & (SH_R5 | SH_R6)) {
            const fp::u256 l1 = ld256(J.lb + 4ull * ri.k1), l2 = ld256(J.lb + 4ull * ri.k2);
            const fp::u256 u1 = ld256(J.ub + 4ull * ri.k1), u2 = ld256(J.ub + 4ull * ri.k2);
            const bool eqb = fp::eq(l1, l2) & fp::eq(u1, u2);
            // x = c written against the constant wire (x - 1 = 0: the R3 and the x == y shapes at once; every "<== 1" of a circuit is
            // one): R3 moves x to [c, c], and when the constant wire already has exactly these bounds (after the first such pop) and
            // no [0,1] class, R4 and R5 find nothing to do (fast_decide's r3x case, fastrow.hip.hpp) -- only x is written. Without
            // this every such row counts as a writer of the constant wire's bounds and they run one behind the other.
            bool partner_settled = false;
            if ((shape & SH_R3) && (shape & SH_R5) && wb0 != 0xFFFFFFFFu && x != 1u && (ri.k1 == x ? ri.k2 : ri.k1) == 1u && (ri.k1 == x || ri.k2 == x)) {
                const uint8_t f1w = J.flags[1];
                const bool k1_is_x = ri.k1 == x;
                partner_settled = (f1w & 3) == 3 && !(f1w & 4) && fp::eq(k1_is_x ? l2 : l1, tv) && fp::eq(k1_is_x ? u2 : u1, tv);
            }
            if ((!eqb || wb0 != 0xFFFFFFFFu) && !partner_settled) { wb1 = ri.k1; wb2 = ri.k2; }   // R3 may first move x's bounds
            wc1 = ri.k1; wc2 = ri.k2;
        } else {
            // binary-decomposition row: only the pivot's bounds can be written
            if ((shape & SH_R4_T) && (shape & SH_R4_T2)) { wb1 = ri.kpos; wb2 = ri.kneg; }
            else if (shape & SH_R4_T2) wb1 = ri.kneg;
            else wb1 = ri.kpos;
            wc1 = wb1; wc2 = wb2;
        }
    }
    const uint32_t c0 = J.rpC[row], c1 = J.rpC[row + 1];
    for (uint32_t base = c0; base < c1; base += 4) {
        uint32_t v[4];
        uint8_t fl[4];
#pragma unroll
        for (uint32_t i = 0; i < 4; ++i) v[i] = base + i < c1 ? J.colC[base + i] : 0xFFFFFFFFu;
#pragma unroll
        for (uint32_t i = 0; i < 4; ++i) fl[i] = v[i] != 0xFFFFFFFFu ? J.flags[v[i]] : (uint8_t)3;
#pragma unroll
        for (uint32_t i = 0; i < 4; ++i) {
            if (v[i] == 0xFFFFFFFFu || (v[i] == 1 && !touch1)) continue;
            const uint32_t u = ((fl[i] & 3) == 3) ? 0u : 1u;
            const uint32_t wb = (v[i] == wb0 || v[i] == wb1 || v[i] == wb2) ? 2u : 0u;
            const uint32_t wc = (v[i] == wc0 || v[i] == wc1 || v[i] == wc2) ? 2u : 0u;
            f(v[i], u | 2u, u | wb, u | wc);
        }
    }
}
template <class F>
__device__ __forceinline__ void for_row_sets(const Job& J, uint32_t row, uint32_t shape, uint32_t x, F f) {
    for_row_sets4(J, row, shape, x, [&](uint32_t v, uint32_t rd, uint32_t wr, uint32_t) { f(v, rd, wr); });
}

// Exact "this pop changes no variable" test against the current state, for rows all of whose
// variables are final. Such a pop only toggles the row's own R4 orientation byte (x == y rows).
// reads_b tells whether the verdict depended on B-class state (then earlier B-writers still block it).
__device__ __noinline__ bool row_is_noop(const Job& J, uint32_t row, const RowInfo& ri, bool& reads_b) {
    const uint32_t shape = ri.shape;
    reads_b = false;
    if (shape & SH_R2_BOUNDSERR) return false;
    if ((shape & SH_R5) && !(shape & (SH_R3 | SH_HAS_AB | SH_C_EMPTY))) {
        // plain x == y row (the bulk of an --O0 circuit): its two variables are in the descriptor, so
        // everything the general test below reads comes back in ONE batch of loads
        const uint8_t f1 = J.flags[ri.k1], f2 = J.flags[ri.k2];
        const fp::u256 l1 = ld256(J.lb + 4ull * ri.k1), l2 = ld256(J.lb + 4ull * ri.k2);
        const fp::u256 u1 = ld256(J.ub + 4ull * ri.k1), u2 = ld256(J.ub + 4ull * ri.k2);
        if ((f1 & f2 & 3) != 3) return false;
        reads_b = true;
        return fp::eq(l1, l2) & fp::eq(u1, u2);
    }
    {
        // (batched like for_row_sets: ids of up to four entries per part, then their flag bytes)
        const uint32_t a0 = J.rpA[row], a1 = J.rpA[row + 1], b0 = J.rpB[row], b1 = J.rpB[row + 1];
        const uint32_t c0 = J.rpC[row], c1 = J.rpC[row + 1];
        uint32_t n = a1 - a0;
        n = b1 - b0 > n ? b1 - b0 : n;
        n = c1 - c0 > n ? c1 - c0 : n;
        for (uint32_t off = 0; off < n; off += 4) {
            uint32_t v[12];
#pragma unroll
            for (uint32_t i = 0; i < 4; ++i) {
                v[i] = a0 + off + i < a1 ? J.colA[a0 + off + i] : 1u;
                v[4 + i] = b0 + off + i < b1 ? J.colB[b0 + off + i] : 1u;
                v[8 + i] = c0 + off + i < c1 ? J.colC[c0 + off + i] : 1u;
            }
            uint32_t all = 3;
#pragma unroll
            for (uint32_t i = 0; i < 12; ++i) all &= J.flags[v[i]];
            if (all != 3) return false;
        }
    }
    if (shape & SH_HAS_AB) return true;            // R1 needs a non-unique variable; R2 needs !is_known(x)
    if (shape & SH_C_EMPTY) return true;
    // linear row, every variable unique and known: R1, R7, R8 cannot fire. R3 / R4 / R5 / R6 may still
    // move bounds or values.
    const bool r4 = (shape & (SH_R4_T | SH_R4_T2)) != 0;
    const bool r56 = (shape & (SH_R5 | SH_R6)) != 0;
    if (r4 && !(shape & SH_R5)) return false;       // binary-decomposition rows are always executed
    if (shape & SH_R3) {
        reads_b = true;
        const uint32_t x = ri.x;
        const fp::u256 tv = ld256(J.vals + 4ull * ri.validx);
        const uint8_t nv = J.nvalues[x];
        const fp::u256 va = ld256(J.values + 8ull * x), lbx = ld256(J.lb + 4ull * x), ubx = ld256(J.ub + 4ull * x);
        if (!(nv == 1 && fp::eq(va, tv) && fp::eq(lbx, tv) && fp::eq(ubx, tv))) return false;
    }
    if (r56) {
        reads_b = true;   // equal bounds (and equal unique bits, given above): R5/R6 return at their first test,
        // and R4 on an x == y row finds either a non-[0,1] partner or already-equal [0,1] bounds
        const fp::u256 l1 = ld256(J.lb + 4ull * ri.k1), l2 = ld256(J.lb + 4ull * ri.k2);
        const fp::u256 u1 = ld256(J.ub + 4ull * ri.k1), u2 = ld256(J.ub + 4ull * ri.k2);
        if (!(fp::eq(l1, l2) && fp::eq(u1, u2))) return false;
    }
    return true;
}

// The three walks of a round over one small row's access set, with the write-marks in device memory
// (multi-workgroup rounds, wide single-workgroup rounds). Out of line on purpose: the access-set walk is
// a few KB of code and a round calls it for up to four rows per lane in three passes -- inlined a dozen
// times it made the queue loop several times larger than the instruction cache.
__device__ __noinline__ void row_mark_global(const Job& J, uint32_t row, uint32_t shape, uint32_t x, uint32_t rank) {
    // only WRITE sets are marked: the readers find write-after-read hazards themselves (row_check_global)
    for_row_sets(J, row, shape, x, [&](uint32_t v, uint32_t rd, uint32_t wr) {
        if (wr & 1) atomicMin(&J.wmarkU[v], rank);
        if (wr & 2) atomicMin(&J.wmarkB[v], rank);
    });
}
// returns the lowest rank at which the prefix has to end because of this row: its own rank when an earlier
// row may write what it reads or writes (blocked), else the lowest later rank that may overwrite what it
// reads, else 0xFFFFFFFF. Marks are updated with device-scope atomics (performed at L2): read past the L1.
__device__ __noinline__ uint32_t row_check_global(const Job& J, uint32_t row, uint32_t shape, uint32_t x, uint32_t rank) {
    bool blocked = false;
    uint32_t mycut = 0xFFFFFFFFu;
    for_row_sets(J, row, shape, x, [&](uint32_t v, uint32_t rd, uint32_t wr) {
        if ((rd | wr) & 1) {
            const uint32_t m = ld_agent(&J.wmarkU[v]);
            if (m < rank) blocked = true; else if (m > rank && m < mycut) mycut = m;
        }
        if ((rd | wr) & 2) {
            const uint32_t m = ld_agent(&J.wmarkB[v]);
            if (m < rank) blocked = true; else if (m > rank && m < mycut) mycut = m;
        }
    });
    return blocked ? rank : mycut;
}
__device__ __noinline__ void row_unmark_global(const Job& J, uint32_t row, uint32_t shape, uint32_t x) {
    for_row_sets(J, row, shape, x, [&](uint32_t v, uint32_t rd, uint32_t wr) {
        if (wr & 1) J.wmarkU[v] = 0xFFFFFFFFu;
        if (wr & 2) J.wmarkB[v] = 0xFFFFFFFFu;
    });
}
// a no-op row that looked at B-class state: blocked if an earlier row may write it
__device__ __noinline__ bool row_noop_blocked_global(const Job& J, uint32_t row, uint32_t rank) {
    bool blocked = false;
    for (uint32_t k = J.rpC[row]; k < J.rpC[row + 1]; ++k)
        if (ld_agent(&J.wmarkB[J.colC[k]]) < rank) blocked = true;
    return blocked;
}

#define ECNE_HSLOTS 4096
#define ECNE_ASET 6
struct ChunkShared {   // LDS of the chunked queue phase
    uint32_t cut;
    uint32_t bases[ECNE_WG + 1];
    uint32_t scan[ECNE_NWAVES + 2];
    unsigned long long acc[12];   // steps, nuniq, hits[0..7], pops, pop_nnz
    uint32_t head, tail, fallback, nbig, flag7;
    uint32_t nbigev, bigev_v[64], bigev_a[64], bigev_b[64];   // high-fan-out events expanded cooperatively
    uint32_t bt[ECNE_BIGTAB];   // lowest candidate index per big target row of this expansion (slot = tbig[row] - 1)
    // small rounds (at most one row per lane): write-marks in an exact LDS hash table instead of device
    // memory, and every lane's access set kept here between the mark and the check pass
    uint32_t hkey[ECNE_HSLOTS], hrank[ECNE_HSLOTS];   // key = 1 + 2 * variable + class (0 = empty); lowest writer rank
    uint32_t aset[ECNE_WG][ECNE_ASET];                // variable | rd << 28 | wr << 30
    uint32_t small_ovf;
    // long rows (> ECNE_SMALL_ROW entries) riding along in a round, at most ECNE_BIGK per workgroup: marked,
    // checked and executed by the whole workgroup, lanes across the row's entries
    uint32_t bl_n, bl_any, bl_rank[ECNE_BIGK], bl_row[ECNE_BIGK], bl_nev[ECNE_BIGK], bl_deg[ECNE_BIGK], bl_base[ECNE_BIGK];
    uint32_t bl_tmp[8];
    uint32_t dr_st[ECNE_BIGK], depoch, dcut;   // drain rounds: state of the registered long rows, this workgroup's copy of the mark epoch, lowest demoted rank
    uint32_t hasbig;
    unsigned long long sd[16];  // schedule diagnostics (ecne_summary.sched)
    unsigned long long mt[8];   // diagnostics of multi-workgroup rounds (master only)
    unsigned long long qt[8];   // diagnostics: 100 MHz ticks in head / mark / check+unmark / exec / flatten / resolve / big / n
};

// ---- exact LDS hash table of write-marks (small rounds). hmark: record that `rank` may write (v, cls);
// hlook: lowest rank that may write it, 0xFFFFFFFF if nobody. Linear probing; the table is wiped as a
// whole after every round. A probe sequence longer than 64 raises small_ovf (the round then falls back
// to the marks in device memory).
__device__ __forceinline__ void hmark(ChunkShared& S, uint32_t v, uint32_t cls, uint32_t rank) {
    const uint32_t key = 1u + 2u * v + cls;
    uint32_t s = (key * 2654435761u) >> (32 - 12);
    for (int probe = 0; probe < 64; ++probe) {
        const uint32_t k = atomicCAS(&S.hkey[s], 0u, key);
        if (k == 0u || k == key) { atomicMin(&S.hrank[s], rank); return; }
        s = (s + 1) & (ECNE_HSLOTS - 1);
    }
    S.small_ovf = 1;
}
__device__ __forceinline__ uint32_t hlook(const ChunkShared& S, uint32_t v, uint32_t cls) {
    const uint32_t key = 1u + 2u * v + cls;
    uint32_t s = (key * 2654435761u) >> (32 - 12);
    for (int probe = 0; probe < 64; ++probe) {
        const uint32_t k = S.hkey[s];
        if (k == key) return S.hrank[s];
        if (k == 0u) return 0xFFFFFFFFu;
        s = (s + 1) & (ECNE_HSLOTS - 1);
    }
    return 0xFFFFFFFFu;   // unreachable when no insertion overflowed (overflow abandons the small path)
}

// A big row (> ECNE_SMALL_ROW entries) popped alone, executed by the whole workgroup instead of one
// wavefront: a 1 025-term sum row costs 2-3 dependent memory round trips instead of 17. Covers the shapes
// long rows have in practice -- R1 on any row with a non-empty C, and R8 / "nothing fires" on a plain
// linear sum -- with the same statistics exec_row() gathers in its fused R1 walk. Returns false, having
// written nothing, when the row may need another rule (R2..R6 shapes, or R7's precondition holds); the
// caller then runs exec_row() on one wavefront. REQUEUE events go to ev[] in the reference's order.
__device__ __noinline__ bool exec_big_row_wg(const Job& J, ChunkShared& S, uint32_t row, uint32_t* ev, uint32_t* nev_out) {
    const int tid = threadIdx.x;
    const uint32_t shape = J.rinfo[row].shape;
    if (shape & (SH_C_EMPTY | SH_R2 | SH_R3 | SH_R4_T | SH_R4_T2 | SH_R5 | SH_R6)) return false;   // uniform
    const uint32_t a0 = J.rpA[row], a1 = J.rpA[row + 1];
    const uint32_t b0 = J.rpB[row], b1 = J.rpB[row + 1];
    const uint32_t c0 = J.rpC[row], c1 = J.rpC[row + 1];
    const uint32_t l = c1 - c0;
    uint32_t* sh = S.bases;   // [0] A/B non-unique, [1] count, [2] the variable, [3] not-known, [4] min tag, [5] max tag
    if (tid == 0) { sh[0] = 0; sh[1] = 0; sh[2] = 0; sh[3] = 0; sh[4] = 0xFFFFFFFFu; sh[5] = 0; }
    __syncthreads();
    bool nuab = false;
    for (uint32_t k = a0 + tid; k < a1; k += ECNE_WG) nuab |= !(J.flags[J.colA[k]] & 1);
    for (uint32_t k = b0 + tid; k < b1; k += ECNE_WG) nuab |= !(J.flags[J.colB[k]] & 1);
    // C in contiguous blocks per thread, so that a thread's events are contiguous in row order
    const uint32_t per = (l + ECNE_WG - 1) / ECNE_WG;
    const uint32_t k0 = c0 + ((uint32_t)tid * per < l ? (uint32_t)tid * per : l);
    const uint32_t k1 = c0 + (((uint32_t)tid + 1) * per < l ? ((uint32_t)tid + 1) * per : l);
    uint32_t cnt = 0, u = 0, amin = 0xFFFFFFFFu, amax = 0;
    bool notknown = false;
    // (four entries per trip: the ids, then the flag bytes, then the group tags of the non-unique ones -- three trips for a thread's share
    //  of a 1 025-term row instead of three per entry)
    for (uint32_t kb = k0; kb < k1; kb += 4) {
        uint32_t v4[4];
        uint8_t f4[4];
        uint32_t a4[4];
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) v4[j] = kb + j < k1 ? J.colC[kb + j] : 0u;
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) f4[j] = kb + j < k1 ? J.flags[v4[j]] : (uint8_t)1;
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) a4[j] = (kb + j < k1 && !(f4[j] & 1)) ? (uint32_t)J.abz[v4[j]] : 0u;   // -1 (no group) is the largest value
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) {
            if (kb + j >= k1 || (f4[j] & 1)) continue;
            if (!cnt) u = v4[j];
            ++cnt;
            if (!(f4[j] & 2)) notknown = true;
            amin = a4[j] < amin ? a4[j] : amin;
            amax = a4[j] > amax ? a4[j] : amax;
        }
    }
    if (nuab) sh[0] = 1;
    if (cnt) {
        atomicAdd(&sh[1], cnt);
        sh[2] = u;                    // only read when the total is 1
        if (notknown) sh[3] = 1;
        atomicMin(&sh[4], amin);
        atomicMax(&sh[5], amax);
    }
    __syncthreads();
    const uint32_t tot = sh[1];
    const bool ab_unique = sh[0] == 0, any_notknown = sh[3] != 0;
    const bool badgroup = sh[4] != sh[5] || sh[5] == 0xFFFFFFFFu;
    const uint32_t the_u = sh[2];
    __syncthreads();                  // sh[] is free again (S.bases is scratch of the scans below)
    uint32_t nev = 0;
    if (ab_unique && tot == 1) {      // R1 (:827-873); nothing is left for R7 / R8 afterwards
        if (tid == 0) {
            J.flags[the_u] |= 3;
            ev[0] = the_u;
            S.acc[0] += 1; S.acc[1] += 1; S.acc[2 + 0] += 1;
        }
        nev = 1;
    } else if (!(shape & SH_HAS_AB) && tot > 0) {
        bool fire7 = false;
        if (!any_notknown) {
            // R7 (:1235-1298) over the sorted order csort[]: contiguous sorted positions per thread; the link
            // across a thread boundary is checked by the later thread against the nearest earlier
            // thread's last non-unique entry (S.bases[t], 0xFFFFFFFF = none)
            const uint32_t s0 = (uint32_t)tid * per < l ? (uint32_t)tid * per : l;
            const uint32_t s1 = ((uint32_t)tid + 1) * per < l ? ((uint32_t)tid + 1) * per : l;
            uint32_t firstk = 0xFFFFFFFFu, lastk = 0xFFFFFFFFu;
            bool fail = false;
            for (uint32_t sp = s0; sp < s1; ++sp) {
                const uint32_t k = c0 + J.csort[c0 + sp];
                if (J.flags[J.colC[k]] & 1) continue;
                if (lastk != 0xFFFFFFFFu) { if (!fail && r7_link_fails(J, k, lastk, false)) fail = true; }
                else firstk = k;
                lastk = k;
            }
            S.bases[tid] = lastk;
            if (tid == 0) S.flag7 = 0;
            __syncthreads();
            if (firstk != 0xFFFFFFFFu && !fail) {
                int t = tid - 1;
                while (t >= 0 && S.bases[t] == 0xFFFFFFFFu) --t;
                if (t >= 0 && r7_link_fails(J, firstk, S.bases[t], false)) fail = true;
            }
            if (fail) S.flag7 = 1;
            __syncthreads();
            if (!S.flag7) {
                // the largest entry: the last thread that saw a non-unique variable holds it
                if (tid == 0) {
                    int t = ECNE_WG - 1;
                    while (t >= 0 && S.bases[t] == 0xFFFFFFFFu) --t;
                    S.flag7 = r7_top_fits(J, S.bases[t], false) ? 2u : 1u;
                }
                __syncthreads();
            }
            fire7 = S.flag7 == 2;
            __syncthreads();
        }
        if (fire7 || !badgroup) {                 // R7, else R8 (:1304-1348): every non-unique variable, in row order
            uint32_t total;
            uint32_t o = wg_exclusive_scan(cnt, S.scan, &total);
            for (uint32_t k = k0; k < k1; ++k) {
                const uint32_t v = J.colC[k];
                if (J.flags[v] & 1) continue;
                J.flags[v] |= 3;
                ev[o++] = v;
            }
            if (tid == 0) { S.acc[0] += tot; S.acc[1] += tot; S.acc[2 + (fire7 ? 6 : 7)] += 1; }
            nev = tot;
        }
    }
    // a linear row all of whose terms are unique now stays that way: its later pops are recognised without a walk (word 1 of its
    // record line, see long_row_walk in fastrow.hip.hpp)
    if (tid == 0 && J.rec != nullptr && !(shape & SH_HAS_AB) && (tot == 0 || nev != 0)) const_cast<uint32_t*>(J.rec)[16ull * row + 1] = 0xFFFFFFFEu;
    if (tid == 0) *nev_out = nev;
    __syncthreads();
    return true;
}

// ---- long rows inside a round. Only "plain" long rows qualify (the shapes exec_big_row_wg executes:
// R1 on anything with a non-empty C, R7 / R8 on a linear sum); the others still end the prefix.
__device__ __forceinline__ bool big_plain(uint32_t shape) {
    return !(shape & (SH_C_EMPTY | SH_R2 | SH_R3 | SH_R4_T | SH_R4_T2 | SH_R5 | SH_R6));
}
// a lane registers its long row; false = no slot left (the caller cuts the prefix there). Slot 0 is kept
// for the row at rank 0, which must never be refused: the prefix always contains rank 0.
__device__ __forceinline__ bool big_register(ChunkShared& S, uint32_t row, uint32_t rank) {
    uint32_t slot = 0;
    if (rank != 0) {
        slot = 1 + atomicAdd(&S.bl_n, 1u);
        if (slot >= ECNE_BIGK) return false;
    }
    S.bl_rank[slot] = rank;
    S.bl_row[slot] = row;
    S.bl_nev[slot] = 0;
    S.bl_any = 1;
    return true;
}
// (thread 0, between rounds) forget the registrations
__device__ __forceinline__ void big_reset(ChunkShared& S) {
    S.bl_n = 0; S.bl_any = 0; S.hasbig = 0;
    for (int k = 0; k < ECNE_BIGK; ++k) S.bl_rank[k] = 0xFFFFFFFFu;
}
// write-marks of the registered long rows: U class of every non-final variable of C (R1 / R7 / R8 may set it)
__device__ __noinline__ void big_rows_mark(const Job& J, ChunkShared& S) {
    for (uint32_t k = 0; k < ECNE_BIGK; ++k) {
        const uint32_t row = S.bl_row[k], rank = S.bl_rank[k];
        if (rank == 0xFFFFFFFFu) continue;   // (uniform) empty slot
        for (uint32_t e = J.rpC[row] + threadIdx.x; e < J.rpC[row + 1]; e += ECNE_WG) {
            const uint32_t v = J.colC[e];
            if ((J.flags[v] & 3) != 3) atomicMin(&J.wmarkU[v], rank);
        }
    }
}
__device__ __noinline__ void big_rows_unmark(const Job& J, ChunkShared& S) {
    for (uint32_t k = 0; k < ECNE_BIGK; ++k) {
        const uint32_t row = S.bl_row[k];
        if (S.bl_rank[k] == 0xFFFFFFFFu) continue;   // (uniform) empty slot
        for (uint32_t e = J.rpC[row] + threadIdx.x; e < J.rpC[row + 1]; e += ECNE_WG) {
            const uint32_t v = J.colC[e];
            if ((J.flags[v] & 3) != 3) J.wmarkU[v] = 0xFFFFFFFFu;
        }
    }
}
// hazards of the registered long rows against the marks (same rule as for a lane's row: a lower mark
// blocks it, a higher one cuts the prefix there). The row reads U of all its variables and B (bounds,
// group tag) of C's non-unique ones. All threads of the workgroup; updates S.cut.
__device__ __noinline__ void big_rows_check(const Job& J, ChunkShared& S) {
    for (uint32_t k = 0; k < ECNE_BIGK; ++k) {
        const uint32_t row = S.bl_row[k], rank = S.bl_rank[k];
        if (rank == 0xFFFFFFFFu) continue;   // (uniform) empty slot
        bool blocked = false;
        uint32_t cutm = 0xFFFFFFFFu;
        auto see = [&](uint32_t m) { if (m < rank) blocked = true; else if (m > rank && m < cutm) cutm = m; };
        for (uint32_t e = J.rpA[row] + threadIdx.x; e < J.rpA[row + 1]; e += ECNE_WG) {
            const uint32_t v = J.colA[e];
            if ((J.flags[v] & 3) != 3) see(ld_agent(&J.wmarkU[v]));
        }
        for (uint32_t e = J.rpB[row] + threadIdx.x; e < J.rpB[row + 1]; e += ECNE_WG) {
            const uint32_t v = J.colB[e];
            if ((J.flags[v] & 3) != 3) see(ld_agent(&J.wmarkU[v]));
        }
        for (uint32_t e = J.rpC[row] + threadIdx.x; e < J.rpC[row + 1]; e += ECNE_WG) {
            const uint32_t v = J.colC[e];
            const uint8_t f = J.flags[v];
            if ((f & 3) != 3) see(ld_agent(&J.wmarkU[v]));
            if (!(f & 1)) see(ld_agent(&J.wmarkB[v]));
        }
        if (blocked) atomicMin(&S.cut, rank);
        else if (cutm != 0xFFFFFFFFu) atomicMin(&S.cut, cutm);
    }
}
// Events of a long row executed inside a round go to a slot of the job's pool (a long row can emit one
// event per term, more than a rank's regular event list holds): [0, maxrow) events, then their
// candidate offsets (multi-workgroup rounds). One slot per (workgroup, registration index).
__device__ __forceinline__ uint32_t* big_ev(const Job& J, uint32_t wgrank, uint32_t k) {
    return J.bigpool + (size_t)(wgrank * ECNE_BIGK + k) * J.bigstride;
}
__device__ __forceinline__ uint32_t* big_off(const Job& J, uint32_t wgrank, uint32_t k) {
    return big_ev(J, wgrank, k) + J.bigstride / 2;
}
// execute the registered long rows that made it into the prefix (rank < c); events go to their pool slot
__device__ __noinline__ void big_rows_exec(const Job& J, ChunkShared& S, uint32_t c, uint32_t wgrank) {
    for (uint32_t k = 0; k < ECNE_BIGK; ++k) {
        const uint32_t row = S.bl_row[k], rank = S.bl_rank[k];
        if (rank == 0xFFFFFFFFu) continue;   // (uniform) empty slot
        if (rank >= c) continue;                                   // uniform
        exec_big_row_wg(J, S, row, big_ev(J, wgrank, k), &S.bl_nev[k]);
    }
    __syncthreads();
}
// number of events a lane's long row emitted (0 if it is none of the registered ones)
__device__ __forceinline__ int big_slot_of(const ChunkShared& S, uint32_t rank) {
    for (uint32_t k = 0; k < ECNE_BIGK; ++k) if (S.bl_rank[k] == rank) return (int)k;
    return -1;
}

// one push candidate: event of rank a wants to push row t as candidate j (see resolve_pushes)
__device__ __forceinline__ void expand_candidate(const Job& J, ChunkShared& S, uint32_t t, uint32_t j, uint32_t a, bool multi) {
    const uint32_t st = J.inq[t];
    const uint32_t bslot = J.tbig[t];
    bool elig;
    if (multi) elig = st == 0 || (st == 2 && J.prank[t] <= a);     // 2 = being popped in this multi round
    else elig = st == 0 || (st >= 2 && st - 2 <= a);              // rank + 2 = being popped at that rank
    J.cand[j] = t | (elig ? 0x80000000u : 0u);
    // many candidates of one round can target the same row (a 1 000-term sum row is pushed by each of
    // its terms): look before the atomic, most of them have already lost
    // A long row is the target of up to one candidate per term (a 1 000-term sum row is pushed by each of
    // its terms in the same round): those meet in an LDS slot first and one atomic per workgroup goes to
    // memory (flush_big_targets); otherwise a thousand same-address atomics serialise at the L2.
    if (!elig) return;
    if (bslot) { if (S.bt[bslot - 1] > j) atomicMin(&S.bt[bslot - 1], j); }
    else if (ld_agent(&J.best[t]) > j) atomicMin(&J.best[t], j);
}
// expand event (v, rank a, candidate base b0): small fan-outs inline, big ones go to the workgroup list
#ifndef ECNE_HUGE_EVENT
#define ECNE_HUGE_EVENT 4096   // REQUEUE events with at least this many rows are expanded by the whole team (rounds on teams of workgroups only)
#endif
#define ECNE_HUGE_SLOTS 8
__device__ __forceinline__ void expand_event(const Job& J, ChunkShared& S, uint32_t v, uint32_t a, uint32_t b0, bool multi) {
    const uint32_t f0 = J.fo_ptr[v], f1 = J.fo_ptr[v + 1];
    if (multi && J.nwg > 1 && f1 - f0 >= ECNE_HUGE_EVENT) {
        // the constant wire's bounds move, say, and every row that mentions it is re-queued (53 250 rows of an ecdsa-scale circuit):
        // 104 strides of ONE workgroup were 0.45 ms of one round; the team takes slices after the expansion barrier (multi_finish)
        const uint32_t slot = atomicAdd(&J.ctr->q_nhuge, 1u);
        if (slot < ECNE_HUGE_SLOTS) { J.ctr->q_huge[slot][0] = v; J.ctr->q_huge[slot][1] = a; J.ctr->q_huge[slot][2] = b0; return; }
    }
    if (f1 - f0 > 48) {
        const uint32_t slot = atomicAdd(&S.nbigev, 1u);
        if (slot < 64) { S.bigev_v[slot] = v; S.bigev_a[slot] = a; S.bigev_b[slot] = b0; return; }
    }
    // four candidates at a time, stage by stage: the loads of one stage are independent of each other, so
    // a lane waits for one memory round trip per stage and not per candidate
    for (uint32_t k = f0; k < f1; k += 4) {
        const uint32_t nn = f1 - k < 4 ? f1 - k : 4;
        uint32_t t[4], st[4], bs[4], pre[4];
        bool el[4];
#pragma unroll
        for (uint32_t i = 0; i < 4; ++i) t[i] = i < nn ? J.fo_rows[k + i] : 0;
#pragma unroll
        for (uint32_t i = 0; i < 4; ++i) { st[i] = i < nn ? J.inq[t[i]] : 1; bs[i] = i < nn ? J.tbig[t[i]] : 0; }
#pragma unroll
        for (uint32_t i = 0; i < 4; ++i) {
            if (multi) el[i] = st[i] == 0 || (st[i] == 2 && J.prank[t[i]] <= a);
            else el[i] = st[i] == 0 || (st[i] >= 2 && st[i] - 2 <= a);
            el[i] = el[i] && i < nn;
            pre[i] = (el[i] && !bs[i]) ? ld_agent(&J.best[t[i]]) : 0;
        }
#pragma unroll
        for (uint32_t i = 0; i < 4; ++i) {
            if (i >= nn) continue;
            const uint32_t j = b0 + (k - f0) + i;
            J.cand[j] = t[i] | (el[i] ? 0x80000000u : 0u);
            if (!el[i]) continue;
            if (bs[i]) { if (S.bt[bs[i] - 1] > j) atomicMin(&S.bt[bs[i] - 1], j); }
            else if (pre[i] > j) atomicMin(&J.best[t[i]], j);
        }
    }
}
// the workgroup's minima for big target rows (S.bt, filled by expand_candidate) go to best[]; the slots are left empty again
__device__ __forceinline__ void flush_big_targets(const Job& J, ChunkShared& S) {
    const uint32_t nb_rows = J.nBigRows < ECNE_BIGTAB ? J.nBigRows : ECNE_BIGTAB;
    for (uint32_t i = threadIdx.x; i < nb_rows; i += ECNE_WG) {
        const uint32_t j = S.bt[i];
        if (j != 0xFFFFFFFFu) { atomicMin(&J.best[J.bigrows[i]], j); S.bt[i] = 0xFFFFFFFFu; }
    }
}
// all threads of the workgroup: expand the listed big events, lanes across fan-out positions
__device__ __forceinline__ void expand_big_events(const Job& J, ChunkShared& S, bool multi) {
    __syncthreads();
    const uint32_t nb = S.nbigev < 64 ? S.nbigev : 64;
    for (uint32_t i = 0; i < nb; ++i) {
        const uint32_t v = S.bigev_v[i], a = S.bigev_a[i], b0 = S.bigev_b[i];
        const uint32_t f0 = J.fo_ptr[v], f1 = J.fo_ptr[v + 1];
        for (uint32_t k = f0 + threadIdx.x; k < f1; k += ECNE_WG) expand_candidate(J, S, J.fo_rows[k], b0 + (k - f0), a, multi);
    }
    __syncthreads();
    if (threadIdx.x == 0) S.nbigev = 0;
    flush_big_targets(J, S);
}
// the team's share of the events published as huge (expand_event): every workgroup takes every (team size)-th stride of each list
__device__ __forceinline__ void expand_huge_events(const Job& J, ChunkShared& S, uint32_t wgrank, uint32_t nh) {
    const uint32_t T = J.nwg * ECNE_WG, g = wgrank * ECNE_WG + threadIdx.x;
    for (uint32_t h = 0; h < nh && h < ECNE_HUGE_SLOTS; ++h) {
        const uint32_t v = ld_agent(&J.ctr->q_huge[h][0]), a = ld_agent(&J.ctr->q_huge[h][1]), b0 = ld_agent(&J.ctr->q_huge[h][2]);
        const uint32_t f0 = J.fo_ptr[v], f1 = J.fo_ptr[v + 1];
        for (uint32_t k = f0 + g; k < f1; k += T) expand_candidate(J, S, J.fo_rows[k], b0 + (k - f0), a, true);
    }
    __syncthreads();
    flush_big_targets(J, S);
}

// Ordered multi-source REQUEUE by the whole workgroup. Input: a flat list of N events (variables) in
// the order the reference would issue REQUEUE(v), each tagged with the rank of the queue entry that
// emitted it (rank_of: J.frank[e] when `ranks` is true, else 0). The result is exactly what calling
// REQUEUE for every event in order leaves in the queue and in inq[]. head >= 0 means ranks are the
// queue entries head.. being popped right now (their inq[] holds rank + 2: a push may re-queue a row
// popped at the same or a lower rank, never one still waiting at a higher rank); head < 0: nothing is
// being popped (sweep phases). Returns the new tail. All threads of the workgroup must call it.
__device__ __noinline__ uint32_t resolve_pushes(const Job& J, ChunkShared& S, const uint32_t* fvar, bool ranks, uint32_t N,
                                   long long head, uint32_t nranks, uint32_t tail, unsigned long long* n_fallback) {
    const int tid = threadIdx.x, lane = lane_id(), w = wave_id();
    if (N == 0) return tail;   // uniform: nothing was emitted
    // candidate base of every event = exclusive scan of the fan-out sizes
    uint32_t M = 0;
    for (uint32_t eb = 0; eb < N; eb += ECNE_WG) {
        const uint32_t e = eb + tid;
        uint32_t d = 0;
        if (e < N) { const uint32_t v = fvar[e]; d = J.fo_ptr[v + 1] - J.fo_ptr[v]; }
        uint32_t tot;
        const uint32_t off = wg_exclusive_scan(d, S.scan, &tot);
        if (e < N) J.fbase[e] = M + off;
        M += tot;
        if (M > ECNE_CANDCAP) break;   // uniform: M and tot are workgroup-wide values
    }
    __syncthreads();
    uint32_t new_tail = tail;
    if (M > ECNE_CANDCAP) {
        // rare (a variable with a huge fan-out): replay the events sequentially on wave 0
        if (n_fallback) (*n_fallback)++;
        if (w == 0) {
            QState qq;
            qq.head = 0; qq.tail = tail; qq.evout = nullptr; qq.nev = 0; qq.emit = 0;
            uint32_t popped = 0;   // ranks < popped have been popped
            for (uint32_t e = 0; e < N; ++e) {
                const uint32_t a = ranks ? J.frank[e] : 0;
                if (head >= 0)
                    for (; popped <= a && popped < nranks; ++popped) {
                        const uint32_t rr = J.queue[((uint32_t)head + popped) & J.qmask];
                        if (lane == 0) J.inq[rr] = 0;
                        wg_fence();
                    }
                requeue(J, qq, fvar[e]);
            }
            if (head >= 0)
                for (; popped < nranks; ++popped) {
                    const uint32_t rr = J.queue[((uint32_t)head + popped) & J.qmask];
                    if (lane == 0 && J.inq[rr] >= 2) J.inq[rr] = 0;
                }
            if (lane == 0) S.tail = qq.tail;
        }
        __syncthreads();
        new_tail = S.tail;
    } else if (M > 0) {
        // event-parallel expansion. A candidate (rank a, target t) may push iff t is not queued
        // "as of rank a": inq[t] == 0, or t is itself being popped at a rank <= a
        if (tid == 0) S.nbigev = 0;
        __syncthreads();
        for (uint32_t e = tid; e < N; e += ECNE_WG)
            expand_event(J, S, fvar[e], ranks ? J.frank[e] : 0, J.fbase[e], false);
        expand_big_events(J, S, false);
        __syncthreads();
        // the earliest eligible candidate of each target wins; winners keep candidate order
        for (uint32_t jb = 0; jb < M; jb += ECNE_WG) {
            const uint32_t j = jb + tid;
            uint32_t t = 0, win = 0;
            if (j < M) {
                const uint32_t cw = J.cand[j];
                t = cw & 0x7FFFFFFFu;
                win = (cw & 0x80000000u) && ld_agent(&J.best[t]) == j;
            }
            uint32_t tot;
            const uint32_t off = wg_exclusive_scan(win, S.scan, &tot);
            if (win) J.queue[(new_tail + off) & J.qmask] = t;
            if (j < M) J.cand[j] = t | (win ? 0x80000000u : 0u);
            new_tail += tot;
        }
        __syncthreads();
        // winners are queued again; forget the per-target minima
        for (uint32_t j = tid; j < M; j += ECNE_WG) {
            const uint32_t cw = J.cand[j];
            const uint32_t t = cw & 0x7FFFFFFFu;
            if (cw & 0x80000000u) J.inq[t] = 1;
            J.best[t] = 0xFFFFFFFFu;
        }
    }
    __syncthreads();
    return new_tail;
}

}  // namespace ecne
