// rules_wave.hip.hpp — the queue cursor, REQUEUE and the wave-cooperative executor of one pop (rules R1..R8, :824-1348): sequential mode, bursts, long rows of the R2..R6 shapes.
// Part of the gfx950 device code of libecne_hip (see kernels.hip.hpp for the overview).
#pragma once
#include "dev_common.hip.hpp"

namespace ecne {

// ====================================================================================== solver
struct QState {   // FIFO cursors, wave-uniform registers of the wave that drives the queue
    uint32_t head, tail;
    // "emit" mode: REQUEUE(v) appends v to evout[] instead of pushing; the pushes are resolved later,
    // in the same order, by the whole workgroup (resolve_pushes)
    uint32_t* evout;
    uint32_t nev;
    uint32_t emit;
};

__device__ __forceinline__ void raise(const Job& J, int code) { atomicCAS(&J.ctr->error, 0, code); }
// An error raised by a queue pop. The reference stops at the FIRST pop that raises; rows of one parallel
// prefix raise concurrently and the error word is only polled every few rounds, so besides setting it
// (which stops the solve) every pop records (its absolute queue position, code) in a min-word: the lowest
// position is the pop the sequential run would have died on. The host reports that code.
__device__ __forceinline__ void raise_ranked(const Job& J, uint32_t queue_pos, int code) {
    atomicMin(&J.ctr->err_key, ((unsigned long long)queue_pos << 8) | (unsigned long long)(-code));
    raise(J, code);
}

// flag bit 4 (16): the variable carries a group tag, abz != -1 (set by P4 :1425-1483, cleared by R2's make_values).
// flag bits 2 and 3 summarise a variable's bounds: 4 = exactly [0,1] (what R4 asks, :1020-1029), 0 = still the
// initial [0, p-1], 8 = anything else -- so that the common x == y pop decides "equal bounds?" from the flag
// bytes alone (chain.hip.hpp); lb / ub themselves are only read when a bound of the third kind is involved.
__device__ __forceinline__ uint32_t bounds_class_bits(const fp::u256& lb, const fp::u256& ub) {
    if (fp::is_zero(lb)) {
        if (fp::is_one(ub)) return 4u;
        if (fp::eq(ub, fp::pminus1())) return 0u;
    }
    return 8u;
}
__device__ __forceinline__ void set_bounds(const Job& J, uint32_t v, const fp::u256& lb, const fp::u256& ub) {
    st256(J.lb + 4ull * v, lb);
    st256(J.ub + 4ull * v, ub);
    uint8_t f = J.flags[v];
    f = (uint8_t)((f & ~12u) | bounds_class_bits(lb, ub));
    J.flags[v] = f;
}

// REQUEUE(v): for each row r of variable_to_indices[v], ascending: push r unless already queued.
// Wave-cooperative; exactly the sequential order because the rows of one list are distinct.
__device__ __noinline__ void requeue(const Job& J, QState& q, uint32_t v) {
    const int lane = lane_id();
    if (q.emit) {
        if (lane == 0) q.evout[q.nev] = v;
        q.nev++;
        return;
    }
    const uint32_t beg = J.fo_ptr[v], end = J.fo_ptr[v + 1];
    for (uint32_t base = beg; base < end; base += 64) {
        uint32_t k = base + lane;
        bool act = k < end;
        uint32_t r = act ? J.fo_rows[k] : 0;
        bool push = act && J.inq[r] == 0;
        uint64_t m = __ballot(push);
        if (push) {
            uint32_t pos = q.tail + (uint32_t)__popcll(m & lanes_below());
            J.queue[pos & J.qmask] = r;
            J.inq[r] = 1;
        }
        q.tail += (uint32_t)__popcll(m);
    }
    wg_fence();
}

// make `v` unique + known (lane 0 writes), wave-uniform
__device__ __forceinline__ void mark_unique(const Job& J, uint32_t v) {
    if (lane_id() == 0) J.flags[v] |= 3;
    wg_fence();
}

// P1 (:735-746): the outputs of special `is` that are not unique yet become unique and known and are re-queued, in order. The
// wavefront takes up to 64 outputs at a time: ids, flag bytes and fan-out bounds in one round trip each, then all push
// candidates of the batch in (output, fan-out position) order -- what one REQUEUE after the other leaves in the queue (a row is
// pushed by the first candidate that finds it out of the queue). One output at a time this was six dependent round trips per
// output, ~50 us per outer iteration on an ECDSA-scale circuit (one adder fires per iteration, six outputs).
__device__ __noinline__ void p1_fire_outputs(const Job& J, QState& q, uint32_t is) {
    const uint32_t lane = (uint32_t)lane_id();
    const uint32_t o0 = J.sp_out_ptr[is], o1 = J.sp_out_ptr[is + 1];
    for (uint32_t ob = o0; ob < o1; ob += 64) {
        const uint32_t nb = o1 - ob < 64u ? o1 - ob : 64u;
        const bool act = lane < nb;
        const uint32_t v = act ? J.sp_out[ob + lane] : 1u;
        const uint8_t f = J.flags[v];
        const uint32_t f0 = J.fo_ptr[v], f1 = J.fo_ptr[v + 1];
        bool dup = false;           // the same variable earlier in the list: that occurrence makes it unique, this one finds it so
        for (uint32_t e = 0; e < nb; ++e) { const uint32_t ve = (uint32_t)__shfl((int)v, (int)e, 64); if (e < lane && ve == v) dup = true; }
        const bool todo = act && !(f & 1) && !dup;
        if (todo) J.flags[v] = (uint8_t)(f | 3);
        const uint32_t deg = todo ? f1 - f0 : 0u;
        uint32_t M;
        const uint32_t cb = wave_excl_scan(deg, &M);
        for (uint32_t jb = 0; jb < M; jb += 64) {
            const uint32_t j = jb + lane;
            const bool have = j < M;
            uint32_t k = 0;
            for (uint32_t e = 0; e < nb; ++e) {
                const uint32_t be = (uint32_t)__shfl((int)cb, (int)e, 64), de = (uint32_t)__shfl((int)deg, (int)e, 64), fe = (uint32_t)__shfl((int)f0, (int)e, 64);
                if (have && j >= be && j - be < de) k = fe + (j - be);
            }
            const uint32_t r = have ? J.fo_rows[k] : 0u;
            bool push = have && J.inq[r] == 0;
            const uint32_t nj = M - jb < 64u ? M - jb : 64u;
            for (uint32_t e = 0; e + 1 < nj; ++e) {      // the same row twice in this batch: the first candidate pushes it
                const uint32_t re = (uint32_t)__shfl((int)r, (int)e, 64);
                const int pe = __shfl((int)push, (int)e, 64);
                if (e < lane && pe && re == r) push = false;
            }
            const uint64_t m = __ballot(push);
            if (push) {
                const uint32_t pos = q.tail + (uint32_t)__popcll(m & lanes_below());
                J.queue[pos & J.qmask] = r;
                J.inq[r] = 1;
            }
            q.tail += (uint32_t)__popcll(m);
            wg_fence();
        }
        wg_fence();
    }
}

// walk C entries [c0,c1) in stored (= reference Set) order; every non-unique variable other than
// `skip` becomes unique and is re-queued, in order. Returns how many.
__device__ __noinline__ uint32_t uniq_range_and_requeue(const Job& J, QState& q, uint32_t c0, uint32_t c1, uint32_t skip) {
    const int lane = lane_id();
    uint32_t n = 0;
    for (uint32_t base = c0; base < c1; base += 64) {
        uint32_t k = base + lane;
        bool act = k < c1;
        uint32_t v = act ? J.colC[k] : 0;
        bool todo = act && v != skip && !(J.flags[v] & 1);
        uint64_t m = __ballot(todo);
        if (!m) continue;
        // REQUEUE never reads flags, so marking this chunk's variables first and re-queueing them
        // afterwards, in order, is the reference's mark-one/requeue-one sequence
        if (todo) J.flags[v] |= 3;
        wg_fence();
        n += (uint32_t)__popcll(m);
        if (q.emit) {
            if (todo) q.evout[q.nev + (uint32_t)__popcll(m & lanes_below())] = v;
            q.nev += (uint32_t)__popcll(m);
        } else {
            while (m) {
                int src = __ffsll((long long)m) - 1;
                m &= m - 1;
                requeue(J, q, __shfl(v, src, 64));
            }
        }
    }
    return n;
}

// R7 (:1257-1272): entry k follows entry prev_k among the row's non-unique variables in sorted-|coefficient|
// order. The link holds when |c_k| is a multiple of |c_prev| and the ratio exceeds the range of prev's variable.
__device__ __noinline__ bool r7_link_fails(const Job& J, uint32_t k, uint32_t prev_k, bool negated) {
    fp::u256 cn = ld256(J.coefC + 4ull * k), cc = ld256(J.coefC + 4ull * prev_k);
    if (negated) { cn = fp::neg(cn); cc = fp::neg(cc); }
    cn = r7_abs(cn); cc = r7_abs(cc);
    fp::u256 qq, rem;
    fp::divmod(cn, cc, qq, rem);
    if (!fp::is_zero(rem)) return true;
    const uint32_t pv = J.colC[prev_k];
    const fp::u256 ub = ld256(J.ub + 4ull * pv), lb = ld256(J.lb + 4ull * pv);
    if (fp::cmp(ub, lb) >= 0) {
        fp::u256 diff;
        fp::sub_raw(diff, ub, lb);
        if (fp::cmp(qq, diff) <= 0) return true;
    }
    return false;
}
// R7's closing test (:1274): |c_last| * (ub(last) + 1) <= p
__device__ __noinline__ bool r7_top_fits(const Job& J, uint32_t last_k, bool negated) {
    const uint32_t lv = J.colC[last_k];
    fp::u256 cl = ld256(J.coefC + 4ull * last_k);
    if (negated) cl = fp::neg(cl);
    cl = r7_abs(cl);
    fp::u256 ub1;
    fp::add_raw(ub1, ld256(J.ub + 4ull * lv), fp::make(1));
    return !fp::mul_gt_p(cl, ub1);
}

#ifdef ECNE_POPPROF
// developer aid: where the time of a strictly sequential pop goes (queue_mode 1), 100 MHz ticks per stage
struct PopProf { unsigned long long last, acc[8]; };
__device__ __forceinline__ PopProf& pop_prof() { __shared__ PopProf p; return p; }
#ifdef ECNE_AVAILHIST
#define ECNE_PT(k) do { } while (0)
#else
#define ECNE_PT(k) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); if (threadIdx.x == 0) { PopProf& pp_ = pop_prof(); const unsigned long long t_ = wall_clock64(); pp_.acc[k] += t_ - pp_.last; pp_.last = t_; } } while (0)
#endif
#else
#define ECNE_PT(k) do { } while (0)
#endif

// R2 check_quadratic (:875-942) as the reference states it, for the one situation the static shapes do not cover: a caller's
// known_variables WITHOUT the constant wire (Job.lv_off bit 3) while variable 1 is not is_known yet. The shapes (SH_R2 / SH_R2_BOUNDSERR /
// SH_R2_DIV0, RowInfo.x, the two precomputed roots) are laid down for "variable 1 is known from the setup on"; here the variable that
// is not known may be the wire itself. Strictly sequential pops only (such a system runs in queue_mode 1: the pop loop of k_solve calls this INSTEAD of
// exec_row() for a row without C -- R1 needs a non-unique variable of C, and behind R2 such a row has nothing left to do, :944-946 or an empty C
// in R3..R8 --; inside exec_row() the call cost every caller of exec_row() registers across it), every lane the same walk.
// Returns true when the pop raised.
__device__ __noinline__ bool r2_constant_wire_free(const Job& J, QState& q, uint32_t row, unsigned long long* hits, unsigned long long& steps) {
    const uint32_t a0 = J.rpA[row], a1 = J.rpA[row + 1], b0 = J.rpB[row], b1 = J.rpB[row + 1];
    uint32_t u = 0, cnt = 0;                                  // `unknown_var` (:880-889): the one variable that is not is_known
    for (uint32_t e = a0; e < a1; ++e) { const uint32_t v = J.colA[e]; if (!(J.flags[v] & 2)) { if (cnt == 0) { u = v; cnt = 1; } else if (v != u) cnt = 2; } }
    for (uint32_t e = b0; e < b1; ++e) { const uint32_t v = J.colB[e]; if (!(J.flags[v] & 2)) { if (cnt == 0) { u = v; cnt = 1; } else if (v != u) cnt = 2; } }
    if (cnt >= 2) return false;
    fp::u256 sa = fp::make(0), ia = fp::make(0), sb = fp::make(0), ib = fp::make(0);
    for (uint32_t e = a0; e < a1; ++e) {                      // :892-899 (`i == unknown_var` is tested first)
        const uint32_t v = J.colA[e];
        if (cnt == 1 && v == u) sa = ld256(J.coefA + 4ull * e);
        else if (v == 1) ia = ld256(J.coefA + 4ull * e);
        else return false;
    }
    for (uint32_t e = b0; e < b1; ++e) {                      // :902-909
        const uint32_t v = J.colB[e];
        if (cnt == 1 && v == u) sb = ld256(J.coefB + 4ull * e);
        else if (v == 1) ib = ld256(J.coefB + 4ull * e);
        else return false;
    }
    if (cnt == 0) { raise_ranked(J, q.head - 1, K_EBOUNDS); return true; }                       // variable_states[-1] (:916, read before the divisions)
    if (fp::is_zero(sa) || fp::is_zero(sb)) { raise_ranked(J, q.head - 1, K_EDIVZERO); return true; }   // :919-920
    const fp::u256 ra = fp::mul(fp::neg(ia), fp::inv(sa)), rb = fp::mul(fp::neg(ib), fp::inv(sb));
    if (lane_id() == 0) {
        st256(J.values + 8ull * u, ra);
        st256(J.values + 8ull * u + 4, rb);
        J.nvalues[u] = 2;
        J.flags[u] = (uint8_t)((J.flags[u] | 2) & ~16u);      // make_values: is_known, abz reset by the constructor (:158)
        J.abz[u] = -1;
        if ((fp::is_zero(ra) && fp::is_one(rb)) || (fp::is_one(ra) && fp::is_zero(rb))) set_bounds(J, u, fp::make(0), fp::make(1));   // :923-927
        J.solved[row] = 1;
    }
    wg_fence();
    requeue(J, q, u);
    steps++; hits[1]++;
    return false;
}

// ---- one queue pop: rules R1..R8 on row `row`, in the reference's order (:824-1348)
__device__ __noinline__ void exec_row(const Job& J, QState& q, uint32_t row, unsigned long long* hits,
                         unsigned long long& steps, unsigned long long& nuniq) {
    const int lane = lane_id();
    const RowInfo ri = J.rinfo[row];
    const uint32_t a0 = J.rpA[row], a1 = J.rpA[row + 1];
    const uint32_t b0 = J.rpB[row], b1 = J.rpB[row + 1];
    const uint32_t c0 = J.rpC[row], c1 = J.rpC[row + 1];
    const uint32_t shape = ri.shape;

    // C is walked once: the R1 pass also gathers the statistics R7 / R8 need,
    // valid as long as no rule in between changes the state (R1 / R3..R6 firing invalidates them).
    const bool fuse = true;
    ECNE_PT(2);
    bool st_valid = false, st_notknown = false, st_badgroup = false;
    uint32_t st_cnt = 0;
    int st_group = -2;
    // R1 check_unique (:827-873)
    {
        bool nu = false;
        for (uint32_t k = a0 + lane; k < a1; k += 64) nu |= !(J.flags[J.colA[k]] & 1);
        for (uint32_t k = b0 + lane; k < b1; k += 64) nu |= !(J.flags[J.colB[k]] & 1);
        ECNE_PT(3);
        if (!__ballot(nu)) {
            uint32_t cnt = 0, u = 0;
            for (uint32_t base = c0; base < c1; base += 64) {
                uint32_t k = base + lane;
                bool act = k < c1;
                uint32_t v = act ? J.colC[k] : 0;
                const uint8_t f = act ? J.flags[v] : 1;
                bool x = act && !(f & 1);
                uint64_t m = __ballot(x);
                if (m && cnt == 0) u = __shfl(v, __ffsll((long long)m) - 1, 64);
                cnt += (uint32_t)__popcll(m);
                if (fuse && m) {
                    // the same walk also collects what R7 and R8 ask of C's non-unique variables
                    if (x && !(f & 2)) st_notknown = true;
                    int a = x ? J.abz[v] : -1;
                    if (st_group == -2) st_group = __shfl(a, __ffsll((long long)m) - 1, 64);
                    if (x && (a == -1 || a != st_group)) st_badgroup = true;
                }
            }
            st_cnt = cnt;
            st_valid = fuse;
            ECNE_PT(4);
            if (cnt == 1) {
                mark_unique(J, u);
                nuniq++; steps++; hits[0]++;
                requeue(J, q, u);
                st_valid = false;
                ECNE_PT(5);
            }
        }
    }
    const unsigned long long steps_at_r1 = steps, nuniq_at_r1 = nuniq;
    // R2 check_quadratic (:875-942)
    if (shape & SH_C_EMPTY) {
        if (shape & SH_R2_BOUNDSERR) { raise_ranked(J, q.head - 1, K_EBOUNDS); return; }   // (the row was popped just before)
        if (shape & SH_R2) {
            const uint32_t x = ri.x;
            if (!(J.flags[x] & 2)) {
                if (shape & SH_R2_DIV0) { raise_ranked(J, q.head - 1, K_EDIVZERO); return; }
                if (lane == 0) {
                    // make_values: is_known, values; abz reset by the constructor (:158)
                    st256(J.values + 8ull * x, ld256(J.vals + 4ull * ri.validx));
                    st256(J.values + 8ull * x + 4, ld256(J.vals + 4ull * (ri.validx + 1)));
                    J.nvalues[x] = 2;
                    J.flags[x] = (uint8_t)((J.flags[x] | 2) & ~16u);   // is_known; the group tag is gone (bit 4)
                    J.abz[x] = -1;
                    if (shape & SH_R2_IS01) set_bounds(J, x, fp::make(0), fp::make(1));   // make_bounds (:923-927)
                    J.solved[row] = 1;
                }
                wg_fence();
                requeue(J, q, x);
                steps++; hits[1]++;
            }
        }
    }
    ECNE_PT(6);
    if (shape & SH_HAS_AB) return;   // (:944-946)
    const uint32_t l = c1 - c0;

    // R3 check_linear (:949-988)
    if (shape & SH_R3) {
        const uint32_t x = ri.x;
        const fp::u256 tv = ld256(J.vals + 4ull * ri.validx);
        bool new_info = false;
        bool same = J.nvalues[x] == 1 && fp::eq(ld256(J.values + 8ull * x), tv);
        uint8_t f = J.flags[x];
        if (!same) { steps++; hits[2]++; new_info = true; }
        if (!(f & 1)) { nuniq++; new_info = true; }
        if (lane == 0) {
            if (!same) { st256(J.values + 8ull * x, tv); J.nvalues[x] = 1; }
            J.flags[x] = (uint8_t)(f | 3);
            set_bounds(J, x, tv, tv);
        }
        wg_fence();
        if (new_info) requeue(J, q, x);
    }
    // R4 checkBinary (:991-1076)
    if ((shape & (SH_R4_T | SH_R4_T2)) && l > 0) {
        uint32_t new_key;
        if ((shape & SH_R4_T) && (shape & SH_R4_T2)) {   // l == 2: the row is negated on every visit
            uint8_t o = (uint8_t)(J.flip3[row] ^ 1);
            if (lane == 0) J.flip3[row] = o;
            new_key = o ? ri.kneg : ri.kpos;
        } else if (shape & SH_R4_T2) {
            new_key = ri.kneg;   // negated once; the -1 entry is the 1 entry from then on
        } else {
            new_key = ri.kpos;
        }
        bool bad = false;   // every other variable needs bounds exactly [0,1] (:1020-1029)
        for (uint32_t k = c0 + lane; k < c1; k += 64) {
            uint32_t v = J.colC[k];
            if (v != new_key && !(J.flags[v] & 4)) bad = true;
        }
        if (!__ballot(bad)) {
            const fp::u256 fub = ld256(J.vals + 4ull * (ri.validx + 1));
            const fp::u256 nlb = ld256(J.lb + 4ull * new_key), nub = ld256(J.ub + 4ull * new_key);
            if (!(fp::is_zero(nlb) && fp::eq(nub, fub))) {
                bool gt = false;   // integer compare ub.d > 2^(l-1) - 1 (:1035)
                if (l - 1 < 254) {
                    fp::u256 ip = fp::make(0);
                    ip.w[(l - 1) >> 6] = 1ull << ((l - 1) & 63);
                    fp::u256 im1;
                    fp::sub_raw(im1, ip, fp::make(1));
                    gt = fp::cmp(nub, im1) > 0;
                }
                if (gt) {
                    if (lane == 0) { set_bounds(J, new_key, fp::make(0), fub); J.flags[new_key] |= 2; }
                    wg_fence();
                    steps++; hits[3]++;
                    requeue(J, q, new_key);
                }
            }
            if (J.flags[new_key] & 1) {   // (:1049-1067)
                uint32_t n = uniq_range_and_requeue(J, q, c0, c1, new_key);
                nuniq += n; steps += n; hits[3] += n;
            }
        }
    }
    // R5 checkpropagateBounds (:1078-1146) and R6 checkOnePropagateBounds (:1148-1232)
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
                // `!=` between a mutable struct and a fresh copy is identity, so both branches run.
                // R5 writes key_1 twice (:1107-1108, sic); R6 writes key_2 (:1188-1189).
                f1 |= 3;
                if (is6) f2 |= 3;
                nuniq += 2;
                ch1 = ch2 = true;
            }
            fp::u256 mn = fp::cmp(ub1, ub2) <= 0 ? ub1 : ub2;
            fp::u256 mx = fp::cmp(lb1, lb2) >= 0 ? lb1 : lb2;
            if (is6 && (!fp::is_one(mn) || !fp::is_zero(mx))) proceed = false;   // (:1196-1199) returns before counting
            bool w1 = false, w2 = false;
            if (proceed) {
                w1 = fp::cmp(ub1, mn) > 0 || fp::cmp(lb1, mx) < 0;
                w2 = fp::cmp(ub2, mn) > 0 || fp::cmp(lb2, mx) < 0;
            }
            if (lane == 0) {
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
            }
            wg_fence();
            if (proceed) {
                ch1 |= w1; ch2 |= w2;
                uint32_t nset = (k1 == k2) ? ((ch1 || ch2) ? 1u : 0u) : ((ch1 ? 1u : 0u) + (ch2 ? 1u : 0u));
                steps += nset;
                if (nset) hits[is6 ? 5 : 4]++;
                // for j in Set(changed_vars): hash order of the (at most two) keys
                uint32_t first = (shape & SH_R56_SWAP) ? k2 : k1, second = (shape & SH_R56_SWAP) ? k1 : k2;
                bool cf = (shape & SH_R56_SWAP) ? ch2 : ch1, cs = (shape & SH_R56_SWAP) ? ch1 : ch2;
                if (cf) requeue(J, q, first);
                if (cs && second != first) requeue(J, q, second);
            }
        }
    }
    // R7 checkModularArithmetic (:1235-1298)
    // (statistics from the R1 walk stay valid only if nothing fired since: R3..R6 always count a step
    //  or a new unique variable when they change anything, except R4's flip byte which no rule reads)
    if (st_valid && (steps != steps_at_r1 || nuniq != nuniq_at_r1)) st_valid = false;
    if (l > 0) {
        uint32_t nunk = 0;
        bool notknown = false;
        if (st_valid) { nunk = st_cnt; notknown = st_notknown; }
        else
        for (uint32_t base = c0; base < c1; base += 64) {
            uint32_t k = base + lane;
            bool act = k < c1;
            uint8_t f = act ? J.flags[J.colC[k]] : 1;
            uint64_t m = __ballot(act && !(f & 1));
            nunk += (uint32_t)__popcll(m);
            if (act && !(f & 1) && !(f & 2)) notknown = true;
        }
        // a long binary decomposition all of whose terms are unique stays that way: its later pops are recognised from word 1 of its
        // (otherwise unused) record line (long_r4_done, fastrow.hip.hpp)
        const bool long_dec = l > 15 && J.rec != nullptr && !(shape & (SH_HAS_AB | SH_C_EMPTY | SH_R3 | SH_R5 | SH_R6))
                              && (((shape & SH_R4_T) != 0) != ((shape & SH_R4_T2) != 0));
        if (nunk == 0 && long_dec && lane == 0) const_cast<uint32_t*>(J.rec)[16ull * row + 1] = 0xFFFFFFFEu;
        if (nunk > 0 && !__ballot(notknown)) {
            const bool negated = (shape & SH_R4_T2) && !(shape & SH_R4_T);
            bool fail = false;
            // previous non-unique entry in sorted order, carried across chunks (wave-uniform)
            uint32_t carry_k = 0xFFFFFFFFu;
            for (uint32_t sb = 0; sb < l; sb += 64) {
                uint32_t s = sb + lane;
                bool act = s < l;
                uint32_t k = act ? c0 + J.csort[c0 + s] : 0;
                uint32_t v = act ? J.colC[k] : 0;
                bool nu = act && !(J.flags[v] & 1);
                uint64_t m = __ballot(nu);
                uint64_t below = m & lanes_below();
                // every lane executes the shuffle (uniform control flow); lanes without an in-chunk
                // predecessor fall back to the carried one
                const int psrc = below ? 63 - __clzll((long long)below) : 0;
                const uint32_t pk = __shfl(k, psrc, 64);
                const uint32_t prev_k = below ? pk : carry_k;
                if (nu && prev_k != 0xFFFFFFFFu && r7_link_fails(J, k, prev_k, negated)) fail = true;
                if (m) carry_k = __shfl(k, 63 - __clzll((long long)m), 64);
                if (__ballot(fail)) { fail = true; break; }   // one broken link settles it: R7 does not fire
            }
            if (!__ballot(fail)) {
                // coeffs[last] * (ub(last) + 1) <= p  (:1274)
                if (r7_top_fits(J, carry_k, negated)) {
                    steps += nunk; hits[6]++;
                    uint32_t n = uniq_range_and_requeue(J, q, c0, c1, 0xFFFFFFFFu);
                    nuniq += n;
                    if (long_dec && lane == 0) const_cast<uint32_t*>(J.rec)[16ull * row + 1] = 0xFFFFFFFEu;      // (every term is unique now)
                }
            }
        }
    }
    // R8 checkAllButOneZeroGroup (:1304-1348)
    if (l > 0) {
        int group = -1;
        bool bad = false;
        uint32_t cnt = 0;
        if (st_valid && steps == steps_at_r1 && nuniq == nuniq_at_r1) { cnt = st_cnt; bad = st_badgroup; }
        else
        for (uint32_t base = c0; base < c1; base += 64) {
            uint32_t k = base + lane;
            bool act = k < c1;
            uint32_t v = act ? J.colC[k] : 0;
            bool nu = act && !(J.flags[v] & 1);
            int a = nu ? J.abz[v] : -1;
            uint64_t m = __ballot(nu);
            if (m) {
                if (group == -1) group = __shfl(a, __ffsll((long long)m) - 1, 64);
                if (nu && (a == -1 || a != group)) bad = true;
                cnt += (uint32_t)__popcll(m);
            }
        }
        // a first non-unique variable with abz == -1 leaves group == -1 and bad == true
        if (cnt > 0 && !__ballot(bad)) {
            hits[7]++;
            uint32_t n = uniq_range_and_requeue(J, q, c0, c1, 0xFFFFFFFFu);
            nuniq += n; steps += n;
        }
    }
}

// R7 then R8 of one row from the state in memory (the closing part of exec_row() without statistics carried over
// from R1): what the chain executor calls in the rare case one of the two could fire after its own fast rules.
__device__ __noinline__ void exec_r78_wave(const Job& J, QState& q, uint32_t row, unsigned long long* hits,
                                           unsigned long long& steps, unsigned long long& nuniq) {
    const int lane = lane_id();
    const uint32_t shape = J.rinfo[row].shape;
    const uint32_t c0 = J.rpC[row], c1 = J.rpC[row + 1];
    const uint32_t l = c1 - c0;
    if (l == 0) return;
    {   // R7 (:1235-1298)
        uint32_t nunk = 0;
        bool notknown = false;
        for (uint32_t base = c0; base < c1; base += 64) {
            uint32_t k = base + lane;
            bool act = k < c1;
            uint8_t f = act ? J.flags[J.colC[k]] : 1;
            uint64_t m = __ballot(act && !(f & 1));
            nunk += (uint32_t)__popcll(m);
            if (act && !(f & 1) && !(f & 2)) notknown = true;
        }
        if (nunk > 0 && !__ballot(notknown)) {
            const bool negated = (shape & SH_R4_T2) && !(shape & SH_R4_T);
            bool fail = false;
            uint32_t carry_k = 0xFFFFFFFFu;
            for (uint32_t sb = 0; sb < l; sb += 64) {
                uint32_t s = sb + lane;
                bool act = s < l;
                uint32_t k = act ? c0 + J.csort[c0 + s] : 0;
                uint32_t v = act ? J.colC[k] : 0;
                bool nu = act && !(J.flags[v] & 1);
                uint64_t m = __ballot(nu);
                uint64_t below = m & lanes_below();
                const int psrc = below ? 63 - __clzll((long long)below) : 0;
                const uint32_t pk = __shfl(k, psrc, 64);
                const uint32_t prev_k = below ? pk : carry_k;
                if (nu && prev_k != 0xFFFFFFFFu && r7_link_fails(J, k, prev_k, negated)) fail = true;
                if (m) carry_k = __shfl(k, 63 - __clzll((long long)m), 64);
                if (__ballot(fail)) { fail = true; break; }
            }
            if (!__ballot(fail) && r7_top_fits(J, carry_k, negated)) {
                steps += nunk; hits[6]++;
                nuniq += uniq_range_and_requeue(J, q, c0, c1, 0xFFFFFFFFu);
            }
        }
    }
    {   // R8 (:1304-1348)
        int group = -1;
        bool bad = false;
        uint32_t cnt = 0;
        for (uint32_t base = c0; base < c1; base += 64) {
            uint32_t k = base + lane;
            bool act = k < c1;
            uint32_t v = act ? J.colC[k] : 0;
            bool nu = act && !(J.flags[v] & 1);
            int a = nu ? J.abz[v] : -1;
            uint64_t m = __ballot(nu);
            if (m) {
                if (group == -1) group = __shfl(a, __ffsll((long long)m) - 1, 64);
                if (nu && (a == -1 || a != group)) bad = true;
                cnt += (uint32_t)__popcll(m);
            }
        }
        if (cnt > 0 && !__ballot(bad)) {
            hits[7]++;
            uint32_t n = uniq_range_and_requeue(J, q, c0, c1, 0xFFFFFFFFu);
            nuniq += n; steps += n;
        }
    }
}

}  // namespace ecne
