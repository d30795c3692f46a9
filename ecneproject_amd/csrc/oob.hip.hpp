// oob.hip.hpp — the reference's LAZY BoundsError for rows and specials that name a variable id above num_variables.
// Part of the gfx950 device code of libecne_hip (see kernels.hip.hpp for the overview).
//
// Reference: `variable_states = [... for i = 1:num_variables]` (src/R1CSConstraintSolver.jl:681), indexed by every rule that looks
// at a variable's state (:829, :835, :841, :881, :1022, :1095, ...). `variable_to_indices` is a DefaultDict (:628), so setup does
// not raise: the run dies at the FIRST read of such a state -- which depends on the order a rule walks a row's variables in and on
// where its early exits are. Such a variable w can never change (every write is preceded by a read), so up to that read the run
// is the run of the widened system the engine solves anyway (w: never unique, never known, initial bounds). What is left to
// decide is, per pop / sweep visit of a row that names such an id, whether the reference reads the state before it leaves the
// rule. That is a function of the flag bytes of the variables IN FRONT of the first such id in the walk's order; the host lays
// those prefixes down once (ecne_engine.hip, build_oob_tables -- the orders are Julia Set orders, jlorder.hpp) and the device only
// evaluates them:
//   pop of a linear row (A, B empty):           always raises (R7 reads `unique` of every C variable, :1240; R1, R4-R6 at the latest there)
//   R1 (:827-873), any other row:               B then A are walked while unique (strict part), C until the second non-unique
//   R2 (:875-942), C empty:                     getVariables order until the second variable that is not is_known
//   P3 (:1357-1417) visit of the row:           getVariables order; only a non-unique variable of A n B in front ends the walk (:1366)
//   P4 (:1425-1483) visit of a row, C empty:   A in nonzeroKeys order while unique (:1431-1436); B's only key (:1443)
//   P5 (:1492-1550) at row i:                   A of row i while unique (:1503); then `variable_states[var_key]` (:1533)
//   P1 (:718-747), P2 (:750-800):               the specials' id lists, directly (the k-loop's dsu roots and `same_set` are static: per-pair codes)
// A system with such ids is solved by ONE workgroup with strictly sequential pops (queue_mode 1), so the first raise is the
// reference's. Malformed input only: none of this is on the path of a well-formed file (Job.oob == nullptr).
#pragma once
#include "rules_wave.hip.hpp"

namespace ecne {

// blob header words (host: OobBlob in ecne_engine.hip)
enum : uint32_t { OOB_NROWS = 0, OOB_NP5 = 1, OOB_NVREF = 2, OOB_DSU_SIZE = 3, OOB_OFF_ROWIDS = 4, OOB_OFF_ROWREC = 5, OOB_OFF_P5 = 6, OOB_OFF_P2 = 7, OOB_HDR = 8 };
// row record: [flags, nS, nCn, nR2, nP3, nP4, vars ...]
enum : uint32_t { OOBF_LINEAR = 1u, OOBF_STRICT_HAS = 2u, OOBF_C_EMPTY = 4u, OOBF_P4_A_HAS = 8u, OOBF_P4_B = 16u, OOB_REC_HDR = 6u };

__device__ __forceinline__ bool oob_all_unique(const Job& J, const uint32_t* v, uint32_t n) {
    for (uint32_t i = 0; i < n; ++i) if (!(J.flags[v[i]] & 1)) return false;
    return true;
}
__device__ __forceinline__ uint32_t oob_count_not(const Job& J, const uint32_t* v, uint32_t n, uint8_t bit) {
    uint32_t c = 0;
    for (uint32_t i = 0; i < n; ++i) c += !(J.flags[v[i]] & bit);
    return c;
}
// record of `row`, or nullptr (binary search over the ascending row ids)
__device__ __noinline__ const uint32_t* oob_find(const Job& J, uint32_t row) {
    const uint32_t* b = J.oob;
    const uint32_t* ids = b + b[OOB_OFF_ROWIDS];
    uint32_t lo = 0, hi = b[OOB_NROWS];
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (ids[mid] < row) lo = mid + 1; else hi = mid;
    }
    if (lo >= b[OOB_NROWS] || ids[lo] != row) return nullptr;
    return b + (b + b[OOB_OFF_ROWREC])[lo];
}
// the pop of `row` (not solved): 0 = the row names no such id (the caller executes it), 1 = it does and the pop does nothing,
// 2 = it does and the reference raises BoundsError
__device__ __noinline__ int oob_pop(const Job& J, uint32_t row) {
    const uint32_t* r = oob_find(J, row);
    if (!r) return 0;
    const uint32_t fl = r[0], nS = r[1], nCn = r[2], nR2 = r[3];
    if (fl & OOBF_LINEAR) return 2;
    const uint32_t* v = r + OOB_REC_HDR;
    // R1: strict part (B, then A) while unique; C until the second non-unique
    if (oob_all_unique(J, v, nS)) {
        if (fl & OOBF_STRICT_HAS) return 2;
        if (!(fl & OOBF_C_EMPTY) && oob_count_not(J, v + nS, nCn, 1) <= 1) return 2;
    }
    // R2 (C empty): the walk over getVariables ends at the second variable that is not is_known
    if ((fl & OOBF_C_EMPTY) && oob_count_not(J, v + nS + nCn, nR2, 2) <= 1) return 2;
    return 1;
}
// P3: the lowest row >= f whose visit raises, or 0xFFFFFFFF
__device__ __noinline__ uint32_t oob_p3_first(const Job& J, uint32_t f) {
    const uint32_t* b = J.oob;
    const uint32_t* ids = b + b[OOB_OFF_ROWIDS];
    const uint32_t* rec = b + b[OOB_OFF_ROWREC];
    for (uint32_t k = 0; k < b[OOB_NROWS]; ++k) {
        if (ids[k] < f) continue;
        const uint32_t* r = b + rec[k];
        if (oob_all_unique(J, r + OOB_REC_HDR + r[1] + r[2] + r[3], r[4])) return ids[k];
    }
    return 0xFFFFFFFFu;
}
// P4 (:1425-1483): the lowest row whose visit raises, or 0xFFFFFFFF. A row with C empty: `unique_a` reads the states of A's keys in
// Set order up to the first one that is not unique (:1431-1436 -- the flag is never used, the reads happen); then, B holding at most
// one key, that key's state (:1443). P4 makes nothing unique, so one evaluation in front of the sweep holds for all of it.
__device__ __noinline__ uint32_t oob_p4_first(const Job& J) {
    const uint32_t* b = J.oob;
    const uint32_t* ids = b + b[OOB_OFF_ROWIDS];
    const uint32_t* rec = b + b[OOB_OFF_ROWREC];
    for (uint32_t k = 0; k < b[OOB_NROWS]; ++k) {
        const uint32_t* r = b + rec[k];
        if (!(r[0] & (OOBF_P4_A_HAS | OOBF_P4_B))) continue;
        if ((r[0] & OOBF_P4_B) || oob_all_unique(J, r + OOB_REC_HDR + r[1] + r[2] + r[3] + r[4], r[5])) return ids[k];
    }
    return 0xFFFFFFFFu;
}
// P5: entries [row, kind, n, vars...] from `*cur` on whose row is <= upto (0xFFFFFFFF: all that are left); true = one of them raises
__device__ __noinline__ bool oob_p5_upto(const Job& J, uint32_t* cur, uint32_t* idx, uint32_t upto) {
    const uint32_t* b = J.oob;
    while (*idx < b[OOB_NP5]) {
        const uint32_t* e = b + b[OOB_OFF_P5] + *cur;
        if (upto != 0xFFFFFFFFu && e[0] > upto) break;
        *cur += 3 + e[2];
        ++*idx;
        if (oob_all_unique(J, e + 3, e[2])) return true;
    }
    return false;
}

// ---- the master's sweeps over the specials and the isZero pairs, one element at a time (wavefront 0, lanes in lockstep)
// P1 (:718-747): inputs are read in order until the first one that is not unique (:723-728); a fired special reads every output (:733)
__device__ __noinline__ void oob_p1(const Job& J, QState& q, unsigned long long* hits, unsigned long long& steps) {
    const uint32_t nVref = J.oob[OOB_NVREF];
    for (uint32_t i = 0; i < J.nSp; ++i) {
        if (J.fired[J.nC + i]) continue;
        bool ok = true;
        for (uint32_t e = J.sp_in_ptr[i]; e < J.sp_in_ptr[i + 1]; ++e) {
            const uint32_t v = J.sp_in[e];
            if (v > nVref) { raise(J, K_EBOUNDS); return; }
            if (!(J.flags[v] & 1)) { ok = false; break; }
        }
        if (!ok) continue;
        if (lane_id() == 0) J.fired[J.nC + i] = 1;
        steps++; hits[8]++;
        for (uint32_t e = J.sp_out_ptr[i]; e < J.sp_out_ptr[i + 1]; ++e)
            if (J.sp_out[e] > nVref) { raise(J, K_EBOUNDS); return; }
        p1_fire_outputs(J, q, i);
    }
}
// P2 (:750-800). Per (BigMultModP i, BigLessThan j) pair the host has decided what is static: bit 0 -- the k-loop's
// `find_root` leaves the dsu (:762); bit 1 -- same_set and constraint_j[3][1] is such an id (:767); bit 2 -- same_set and one of
// constraint_i[3], constraint_i[2][1,2,3,7,8,9] is (:769-783, read iff that output's values are [1])
__device__ __noinline__ void oob_p2(const Job& J, QState& q, unsigned long long* hits) {
    const uint32_t nVref = J.oob[OOB_NVREF];
    const uint32_t* codes = J.oob + J.oob[OOB_OFF_P2];
    for (uint32_t a = 0; a < J.nK1; ++a) {
        const uint32_t i = J.k1_list[a];
        for (uint32_t bj = 0; bj < J.nK2; ++bj) {
            const uint32_t j = J.k2_list[bj];
            if (!J.secp_solve) { raise(J, K_EUNDEF_DSU); return; }
            const uint32_t ni = J.sp_in_ptr[i + 1] - J.sp_in_ptr[i], nj = J.sp_in_ptr[j + 1] - J.sp_in_ptr[j];
            if (ni < 9 || nj < 6) { raise(J, K_EBOUNDS); return; }
            const uint32_t code = codes[a * J.nK2 + bj];
            if (code & 3u) { raise(J, K_EBOUNDS); return; }
            if (code & 4u) {
                const uint32_t o = J.sp_out[J.sp_out_ptr[j]];
                if (J.nvalues[o] == 1 && fp::eq(ld256(J.values + 8ull * o), fp::make(1))) { raise(J, K_EBOUNDS); return; }
            }
            hits[9]++;
            for (uint32_t t = 0; t < 3; ++t) {
                const uint32_t v = J.sp_in[J.sp_in_ptr[j] + t];
                if (v > nVref) { raise(J, K_EBOUNDS); return; }
                if (J.flags[v] & 1) continue;
                mark_unique(J, v);
                requeue(J, q, v);
            }
        }
    }
}
// P5 (:1492-1550): the static candidates and the rows whose visit can raise, merged in row order
__device__ __noinline__ void oob_p5(const Job& J, QState& q, unsigned long long* hits, unsigned long long& steps) {
    uint32_t cur = 0, idx = 0;
    for (uint32_t i = 0; i < J.nP5; ++i) {
        const uint32_t r = J.p5_rows[i], y = J.p5_y[i];
        if (oob_p5_upto(J, &cur, &idx, r)) { raise(J, K_EBOUNDS); return; }
        if (J.flags[y] & 1) continue;
        bool can = true;
        for (uint32_t e = J.rpA[r]; e < J.rpA[r + 1] && can; ++e) can = (J.flags[J.colA[e]] & 1) != 0;
        if (!can) continue;
        mark_unique(J, y);
        if (lane_id() == 0) { J.solved[r] = 1; J.solved[r + 1] = 1; }
        wg_fence();
        steps++; hits[12]++;
        requeue(J, q, y);
    }
    if (oob_p5_upto(J, &cur, &idx, 0xFFFFFFFFu)) raise(J, K_EBOUNDS);
}

}  // namespace ecne
