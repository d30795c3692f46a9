// kernels.hip.hpp — gfx950 device code of the Ecne propagation engine (included by ecne_engine.hip).
//
// Kernels
//   k_classify_rows   two passes (one lane per short row, one wavefront per longer row): streams the
//                     row's (col, coeff) pairs once, decides the static shape of the row (which of the
//                     reference's rules R2..R8 it can ever feed), computes the rule constants that need
//                     field arithmetic (the two roots of a bit-check row, the value of a
//                     single-variable linear row, the power-of-two bound of a binary-decomposition
//                     row) and the |coefficient| order rule R7 walks. HBM-bound streaming pass: ~36 B
//                     per non-zero in, 32 B per row + 4 B per C non-zero out.  Reference: the pattern
//                     tests re-done on every queue visit at src/R1CSConstraintSolver.jl:875-927,
//                     :949-964, :999-1013, :1245-1265.
//   k_solve           the whole fixed point (outer loop :706-1556) in one persistent launch: per system
//                     1..96 cooperating workgroups of 512 threads (SPMD, hand-rolled job barrier), FIFO
//                     worklist in HBM/L2, rules R1-R8, batch phases P1-P5, verdict counts.  All
//                     ordering-sensitive steps follow the reference's sequential order exactly (see
//                     DESIGN.md "Schedule").
//   k_fp_selftest     field-arithmetic known-answer vectors on the device.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "engine_types.hpp"
#define ECNE_FINE_TICKS 1   // in-kernel phase clocks (measured: no effect on the solve time)
#include "fp256.hpp"

namespace ecne {

#define ECNE_WG 512
#define ECNE_NWAVES (ECNE_WG / 64)

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }
__device__ __forceinline__ uint64_t lanes_below() { return (1ull << lane_id()) - 1ull; }
__device__ __forceinline__ void wg_fence() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); }

__device__ __forceinline__ fp::u256 ld256(const uint64_t* p) { return fp::make(p[0], p[1], p[2], p[3]); }
__device__ __forceinline__ void st256(uint64_t* p, const fp::u256& v) {
    p[0] = v.w[0]; p[1] = v.w[1]; p[2] = v.w[2]; p[3] = v.w[3];
}
__device__ __forceinline__ fp::u256 shfl256(const fp::u256& v, int src) {
    fp::u256 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        unsigned lo = __shfl((unsigned)(v.w[i] & 0xffffffffu), src, 64);
        unsigned hi = __shfl((unsigned)(v.w[i] >> 32), src, 64);
        r.w[i] = ((uint64_t)hi << 32) | lo;
    }
    return r;
}

// abs(flip_coeffs(x)) of rule R7 (:1245-1259): values above the literal threshold are taken as
// negative numbers. The literal is NOT p-1 (it is ~1e75 below it) and is kept exactly.
__device__ __forceinline__ fp::u256 r7_abs(const fp::u256& c) {
    const fp::u256 thr = fp::make(0x43e1f593f0000000ULL, 0x9c41be16bb2a8891ULL, 0x045fcd3eea44076aULL,
                                  0x2e2e53955f6f1dfeULL);
    if (fp::cmp(c, thr) > 0) {
        fp::u256 t;
        fp::sub_raw(t, fp::modulus(), c);
        return t;
    }
    return c;
}

__device__ __forceinline__ uint32_t ld_agent(const uint32_t* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

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
            val = fp::mul(fp::neg(icpt), fp::inv(slope));
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
            st256(J.vals + 4ull * ri.validx, fp::mul(fp::neg(c1v), fp::inv(cx)));
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
                for (uint32_t base = c0; base < c1; base += 64) {
                    uint32_t k = base + lane;
                    bool act = k < c1;
                    fp::u256 c = act ? ld256(J.coefC + 4ull * k) : fp::make(2);
                    uint32_t v = act ? J.colC[k] : 0;
                    uint64_t m1 = __ballot(act && fp::is_one(c)), m2 = __ballot(act && fp::is_one(fp::neg(c)));
                    n_one += __popcll(m1);
                    n_mone += __popcll(m2);
                    if (m1) kpos = __shfl(v, __ffsll((long long)m1) - 1, 64);
                    if (m2) kneg = __shfl(v, __ffsll((long long)m2) - 1, 64);
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
            // long sum rows usually carry one |coefficient| (all +-1): the stable order is then the stored one
            bool all_same = true;
            {
                fp::u256 first = ld256(J.coefC + 4ull * c0);
                if (negated) first = fp::neg(first);
                first = r7_abs(first);
                for (uint32_t base = c0; base < c1 && all_same; base += 64) {
                    uint32_t k = base + lane;
                    bool diff = false;
                    if (k < c1) {
                        fp::u256 c = ld256(J.coefC + 4ull * k);
                        if (negated) c = fp::neg(c);
                        diff = !fp::eq(r7_abs(c), first);
                    }
                    if (__ballot(diff)) all_same = false;
                }
            }
            if (all_same)
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

// Rows with at most ECNE_CLS_LANE entries in C (almost all of them) are classified by ONE lane each:
// 64 rows per wavefront, so the field inversions of bit-check / single-variable rows run on full
// SIMDs instead of one lane of a wave. Same results as classify_row, written serially.
#define ECNE_CLS_LANE 8
__device__ void classify_row_lane(const Job& J, uint32_t row) {
    RowInfo ri = J.rinfo[row];
    const uint32_t c0 = J.rpC[row], c1 = J.rpC[row + 1];
    const uint32_t l = c1 - c0;
    uint32_t shape = ri.shape;
    if ((shape & SH_R2) && !(shape & SH_R2_DIV0)) {
        fp::u256 val[2];
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
            val[part] = fp::mul(fp::neg(icpt), fp::inv(slope));
            st256(J.vals + 4ull * (ri.validx + part), val[part]);
        }
        if ((fp::is_zero(val[0]) && fp::is_one(val[1])) || (fp::is_one(val[0]) && fp::is_zero(val[1]))) shape |= SH_R2_IS01;
    }
    if (!(shape & SH_HAS_AB) && l > 0) {
        if (shape & SH_R3) {
            fp::u256 c1v = fp::make(0), cx = fp::make(0);
            for (uint32_t k = c0; k < c1; ++k) {
                uint32_t v = J.colC[k];
                if (v == 1) c1v = ld256(J.coefC + 4ull * k);
                else if (v == ri.x) cx = ld256(J.coefC + 4ull * k);
            }
            st256(J.vals + 4ull * ri.validx, fp::mul(fp::neg(c1v), fp::inv(cx)));
        }
        fp::u256 key[ECNE_CLS_LANE];
        if (!(shape & SH_CZERO)) {
            uint32_t n_one = 0, n_mone = 0, kpos = 0, kneg = 0;
            uint32_t maskT = 0, maskT2 = 0;     // exponents seen (l <= 8: exponents 0..6)
            bool okT = true, okT2 = true;
            for (uint32_t k = c0; k < c1; ++k) {
                const fp::u256 c = ld256(J.coefC + 4ull * k);
                const fp::u256 nc = fp::neg(c);
                const bool one = fp::is_one(c), mone = fp::is_one(nc);
                if (one) { ++n_one; if (n_one == 1) kpos = J.colC[k]; }
                if (mone) { ++n_mone; if (n_mone == 1) kneg = J.colC[k]; }
                if (!one) {
                    int e = (popc256(nc) == 1) ? ctz256(nc) : 999;
                    if (e > (int)l - 2 || (maskT >> e & 1)) okT = false; else maskT |= 1u << e;
                }
                if (!mone) {
                    int e = (popc256(c) == 1) ? ctz256(c) : 999;
                    if (e > (int)l - 2 || (maskT2 >> e & 1)) okT2 = false; else maskT2 |= 1u << e;
                }
            }
            const bool isT = okT && n_one == 1, isT2 = okT2 && n_mone == 1;
            if (isT) shape |= SH_R4_T;
            if (isT2) shape |= SH_R4_T2;
            if (isT || isT2) {
                ri.kpos = kpos;
                ri.kneg = kneg;
                fp::u256 pw = fp::make(1);
                for (uint32_t s = 0; s + 1 < l; ++s) pw = fp::add(pw, pw);
                st256(J.vals + 4ull * (ri.validx + 1), fp::sub(pw, fp::make(1)));
            }
        }
        // R7 order: stable insertion sort of the (at most 8) entries by |signed coefficient|
        {
            const bool negated = (shape & SH_R4_T2) && !(shape & SH_R4_T);
            uint32_t idx[ECNE_CLS_LANE];
            for (uint32_t k = 0; k < l; ++k) {
                fp::u256 c = ld256(J.coefC + 4ull * (c0 + k));
                if (negated) c = fp::neg(c);
                c = r7_abs(c);
                uint32_t pos = k;
                while (pos > 0 && fp::cmp(key[pos - 1], c) > 0) { key[pos] = key[pos - 1]; idx[pos] = idx[pos - 1]; --pos; }
                key[pos] = c;
                idx[pos] = k;
            }
            for (uint32_t k = 0; k < l; ++k) J.csort[c0 + k] = idx[k];
            shape |= SH_R7_SORTED;
        }
    }
    if ((shape & SH_C_HAS1) && (shape & (SH_R4_T | SH_R4_T2 | SH_R5))) shape |= SH_TOUCH1;
    ri.shape = shape;
    J.rinfo[row] = ri;
}

// pass 0: one lane per row for rows with lenC <= ECNE_CLS_LANE; pass 1: one wavefront per remaining row
__global__ __launch_bounds__(256) void k_classify_rows(const Job* jobs, uint32_t job_index, uint32_t pass) {
    __shared__ uint32_t scratch[4][16];
    __shared__ Job sJ;
    if (threadIdx.x < sizeof(Job) / 4) ((uint32_t*)&sJ)[threadIdx.x] = ((const uint32_t*)&jobs[job_index])[threadIdx.x];
    __syncthreads();
    if (pass == 0) {
        for (uint32_t row = blockIdx.x * 256 + threadIdx.x; row < sJ.nC; row += gridDim.x * 256)
            if (sJ.rinfo[row].lenC <= ECNE_CLS_LANE) classify_row_lane(sJ, row);
        return;
    }
    const uint32_t wave = threadIdx.x >> 6;
    // the long rows are listed by the host (big_list): one wavefront each
    for (uint32_t i = blockIdx.x * 4 + wave; i < sJ.nBigCls; i += gridDim.x * 4) classify_row(sJ, sJ.cls_list[i], scratch[wave]);
}

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

__device__ __forceinline__ void set_bounds(const Job& J, uint32_t v, const fp::u256& lb, const fp::u256& ub) {
    st256(J.lb + 4ull * v, lb);
    st256(J.ub + 4ull * v, ub);
    uint8_t f = J.flags[v];
    f = (uint8_t)((f & ~4u) | ((fp::is_zero(lb) && fp::is_one(ub)) ? 4u : 0u));
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
    bool st_valid = false, st_notknown = false, st_badgroup = false;
    uint32_t st_cnt = 0;
    int st_group = -2;
    // R1 check_unique (:827-873)
    {
        bool nu = false;
        for (uint32_t k = a0 + lane; k < a1; k += 64) nu |= !(J.flags[J.colA[k]] & 1);
        for (uint32_t k = b0 + lane; k < b1; k += 64) nu |= !(J.flags[J.colB[k]] & 1);
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
            if (cnt == 1) {
                mark_unique(J, u);
                nuniq++; steps++; hits[0]++;
                requeue(J, q, u);
                st_valid = false;
            }
        }
    }
    const unsigned long long steps_at_r1 = steps, nuniq_at_r1 = nuniq;
    // R2 check_quadratic (:875-942)
    if (shape & SH_C_EMPTY) {
        if (shape & SH_R2_BOUNDSERR) { raise(J, K_EBOUNDS); return; }
        if (shape & SH_R2) {
            const uint32_t x = ri.x;
            if (!(J.flags[x] & 2)) {
                if (shape & SH_R2_DIV0) { raise(J, K_EDIVZERO); return; }
                if (lane == 0) {
                    // make_values: is_known, values; abz reset by the constructor (:158)
                    st256(J.values + 8ull * x, ld256(J.vals + 4ull * ri.validx));
                    st256(J.values + 8ull * x + 4, ld256(J.vals + 4ull * (ri.validx + 1)));
                    J.nvalues[x] = 2;
                    J.flags[x] |= 2;
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


// ---------------------------------------------------------------------------------- workgroup tools
// exclusive prefix sum of one value per thread over the workgroup; returns the
// thread's offset, *total receives the sum. lds: ECNE_NWAVES + 1 words.
__device__ uint32_t wg_exclusive_scan(uint32_t x, uint32_t* lds, uint32_t* total) {
    const int lane = lane_id(), w = wave_id();
    uint32_t incl = x;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t t = __shfl_up(incl, d, 64);
        if (lane >= d) incl += t;
    }
    __syncthreads();
    if (lane == 63) lds[w] = incl;
    __syncthreads();
    if (w == 0) {
        uint32_t v = lane < ECNE_NWAVES ? lds[lane] : 0, inc = v;
#pragma unroll
        for (int d = 1; d < ECNE_NWAVES; d <<= 1) {
            uint32_t t = __shfl_up(inc, d, 64);
            if (lane >= d) inc += t;
        }
        if (lane < ECNE_NWAVES) lds[lane] = inc - v;
        if (lane == ECNE_NWAVES - 1) lds[ECNE_NWAVES] = inc;
    }
    __syncthreads();
    uint32_t off = lds[w] + incl - x;
    *total = lds[ECNE_NWAVES];
    __syncthreads();
    return off;
}

// Ordered multi-source REQUEUE: events[0..n) are variables in the reference's order; equivalent to
// calling requeue() for each in turn. Driven by one wavefront (the caller passes the queue cursor).
__device__ void requeue_events(const Job& J, QState& q, const uint32_t* events, uint32_t n) {
    for (uint32_t e = 0; e < n; ++e) requeue(J, q, events[e]);
}

// 128-bit commutative hash of a set of variable ids (P3 group key: the sorted unknown tuple, :1386-1387)
__device__ __forceinline__ uint64_t mixA(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}
__device__ __forceinline__ uint64_t mixB(uint64_t x) {
    x += 0x9e3779b97f4a7c15ULL; x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ULL; x = (x ^ (x >> 27)) * 0x94d049bb133111ebULL;
    return x ^ (x >> 31);
}

// P3 eligibility of one row (one lane per row): no non-unique variable in A or B; k = number of
// non-unique variables of C; h/h2 = commutative hash of that set (:1360-1386).
__device__ __noinline__ void p3_eval(const Job& J, uint32_t row, uint32_t& k, uint64_t& h, uint64_t& h2) {
    k = 0; h = 0; h2 = 0;
    // entries four per part at a time: ids first, then flag bytes (the constant wire pads short parts)
    const uint32_t a0 = J.rpA[row], a1 = J.rpA[row + 1], b0 = J.rpB[row], b1 = J.rpB[row + 1];
    const uint32_t c0 = J.rpC[row], c1 = J.rpC[row + 1];
    uint32_t n = a1 - a0;
    n = b1 - b0 > n ? b1 - b0 : n;
    n = c1 - c0 > n ? c1 - c0 : n;
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
        uint32_t ab = 1;
#pragma unroll
        for (uint32_t i = 0; i < 8; ++i) ab &= fl[i];
        if (!(ab & 1)) { k = 0xFFFFFFFFu; return; }
#pragma unroll
        for (uint32_t i = 8; i < 12; ++i)
            if (!(fl[i] & 1)) { ++k; h += mixA(v[i]); h2 += mixB(v[i]); }
    }
    h = mixA(h + k);   // never 0-sensitive: empty sets are not inserted
}
// Open-addressing table keyed by the 64-bit half of the group hash; the other half is recorded with
// a second CAS by every visitor, so two different keys that agree on 64 bits are DETECTED (the solve
// stops with ECNE_ECAPACITY) instead of being merged. No lane ever spins on another lane.
__device__ __forceinline__ uint32_t ht_slot(const Job& J, uint64_t h, uint64_t h2, bool insert, bool* created = nullptr) {
    const unsigned long long key = (unsigned long long)(h | 1ull), key2 = (unsigned long long)(h2 | 1ull);
    uint32_t s = (uint32_t)((h >> 1) & J.htmask);
    for (uint32_t probe = 0; probe <= J.htmask; ++probe) {
        unsigned long long cur = __hip_atomic_load((unsigned long long*)&J.ht_key[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur == 0ull) {
            if (!insert) return 0xFFFFFFFFu;
            cur = atomicCAS((unsigned long long*)&J.ht_key[s], 0ull, key);
            if (cur == 0ull) { cur = key; if (created) *created = true; }
        }
        if (cur == key) {
            unsigned long long o2 = atomicCAS((unsigned long long*)&J.ht_key2[s], 0ull, key2);
            if (o2 != 0ull && o2 != key2) { raise(J, K_ECAPACITY); return 0xFFFFFFFFu; }
            return s;
        }
        s = (s + 1) & J.htmask;
    }
    raise(J, K_ECAPACITY);
    return 0xFFFFFFFFu;
}

// coefficient of variable v in row's C part (0 if absent)
__device__ fp::u256 c_coef(const Job& J, uint32_t row, uint32_t v) {
    for (uint32_t e = J.rpC[row]; e < J.rpC[row + 1]; ++e)
        if (J.colC[e] == v) return ld256(J.coefC + 4ull * e);
    return fp::make(0);
}

// slow_det (:1389-1400): sum over ODD permutations only (Combinatorics.parity is 0 for even and
// 1 for odd permutations and is used as a factor). Wave-parallel over permutation indices.
// rows[0..k) in arrival order, vars[0..k) ascending. Returns non-zero?
__device__ __noinline__ bool p3_odd_perm_sum_nonzero(const Job& J, const uint32_t* rows, const uint32_t* vars, uint32_t k) {
    const int lane = lane_id();
    uint64_t nperm = 1;
    for (uint32_t i = 2; i <= k; ++i) nperm *= i;
    fp::u256 acc = fp::make(0);
    for (uint64_t pi = lane; pi < nperm; pi += 64) {
        // decode permutation number pi (factoradic) into perm[], count inversions
        uint32_t perm[10], avail[10];
        for (uint32_t i = 0; i < k; ++i) avail[i] = i;
        uint64_t rem = pi;
        uint64_t f = nperm;
        uint32_t inv = 0;
        for (uint32_t i = 0; i < k; ++i) {
            f /= (k - i);
            uint32_t d = (uint32_t)(rem / f);
            rem -= (uint64_t)d * f;
            perm[i] = avail[d];
            inv += d;
            for (uint32_t j = d; j + 1 < k - i; ++j) avail[j] = avail[j + 1];
        }
        if (inv & 1) {
            fp::u256 term = fp::make(1);
            for (uint32_t j = 0; j < k; ++j) term = fp::mul(term, c_coef(J, rows[j], vars[perm[j]]));
            acc = fp::add(acc, term);
        }
    }
    // wave reduction (field addition)
    for (int d = 32; d >= 1; d >>= 1) {
        fp::u256 o = shfl256(acc, (lane + d) & 63);
        if (lane < d) acc = fp::add(acc, o);
    }
    acc = shfl256(acc, 0);
    return !fp::is_zero(acc);
}

// ---------------------------------------------------------------------------------------- k_solve
// workgroup-uniform view of the device error word (every thread takes the same branch)
__device__ __forceinline__ int wg_error(const Job& J, int* s_err) {
    __syncthreads();
    if (threadIdx.x == 0) *s_err = atomicAdd(&J.ctr->error, 0);
    __syncthreads();
    return *s_err;
}

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
        f[i] = (uint8_t)((f[i] & ~4u) | ((fp::is_zero(nlb) && fp::is_one(nub)) ? 4u : 0u));
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
        if (shape & SH_R2_BOUNDSERR) { raise(J, K_EBOUNDS); return; }
        if (shape & SH_R2) {
            const uint32_t x = ri.x;
            if (!(J.flags[x] & 2)) {
                if (shape & SH_R2_DIV0) { raise(J, K_EDIVZERO); return; }
                st256(J.values + 8ull * x, ld256(J.vals + 4ull * ri.validx));
                st256(J.values + 8ull * x + 4, ld256(J.vals + 4ull * (ri.validx + 1)));
                J.nvalues[x] = 2;
                J.flags[x] |= 2;
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

// Access sets of a small row for the conflict test, f(v, read_mask, write_mask) with bit0 = U-class
// (unique / is_known bits) and bit1 = B-class (lb, ub, bounds01 bit, values, abz):
//     non-linear row, C non-empty : reads U of A u B u C, may write U of C                  (R1)
//     C empty, bit-check shaped   : reads U(x), may write U(x) and B(x)                      (R2)
//     C empty, anything else      : touches nothing
//     linear row                  : reads and may write U and B of C                          (R1, R3..R8)
// U-class state of a variable is FINAL once both bits are set (they are only ever set), so U-class
// accesses to such variables are dropped: no row can change them and every reader sees the same value.
template <class F>
__device__ __forceinline__ void for_row_sets(const Job& J, uint32_t row, uint32_t shape, uint32_t x, F f) {
    if (shape & SH_C_EMPTY) {
        if (shape & SH_R2) {
            const bool fin = (J.flags[x] & 3) == 3;
            f(x, fin ? 0u : 1u, fin ? 2u : 3u);
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
                if ((fl[i] & 3) != 3) f(v[i], 1u, (i >= 8 && may_write && !(one_batch && (fl[i] & 1))) ? 1u : 0u);
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
        f(ri.k1, o1 | 2u, o1 | wb);
        f(ri.k2, o2 | 2u, o2 | wb);
        return;
    }
    const bool touch1 = (shape & SH_TOUCH1) != 0;
    uint32_t wb0 = 0xFFFFFFFFu, wb1 = 0xFFFFFFFFu, wb2 = 0xFFFFFFFFu;   // variables whose B-state may be written
    if (shape & SH_R3) {
        const RowInfo ri = J.rinfo[row];
        const fp::u256 tv = ld256(J.vals + 4ull * ri.validx);
        // (all four loads first: a short-circuit chain would wait for them one after the other)
        const uint8_t nv = J.nvalues[x];
        const fp::u256 va = ld256(J.values + 8ull * x), lbx = ld256(J.lb + 4ull * x), ubx = ld256(J.ub + 4ull * x);
        const bool same = (nv == 1) & fp::eq(va, tv) & fp::eq(lbx, tv) & fp::eq(ubx, tv);
        if (!same) wb0 = x;
    }
    if (shape & (SH_R4_T | SH_R4_T2 | SH_R5 | SH_R6)) {
        const RowInfo ri = J.rinfo[row];
        if (shape & (SH_R5 | SH_R6)) {
            const fp::u256 l1 = ld256(J.lb + 4ull * ri.k1), l2 = ld256(J.lb + 4ull * ri.k2);
            const fp::u256 u1 = ld256(J.ub + 4ull * ri.k1), u2 = ld256(J.ub + 4ull * ri.k2);
            const bool eqb = fp::eq(l1, l2) & fp::eq(u1, u2);
            if (!eqb || wb0 != 0xFFFFFFFFu) { wb1 = ri.k1; wb2 = ri.k2; }   // R3 may first move x's bounds
        } else {
            // binary-decomposition row: only the pivot's bounds can be written
            if ((shape & SH_R4_T) && (shape & SH_R4_T2)) { wb1 = ri.kpos; wb2 = ri.kneg; }
            else if (shape & SH_R4_T2) wb1 = ri.kneg;
            else wb1 = ri.kpos;
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
            f(v[i], u | 2u, u | wb);
        }
    }
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
    uint32_t acnt[ECNE_WG];                           // entries cached; ECNE_ASET + 1 = too many, walk the row again
    uint32_t small_ovf;
    // long rows (> ECNE_SMALL_ROW entries) riding along in a round, at most ECNE_BIGK per workgroup: marked,
    // checked and executed by the whole workgroup, lanes across the row's entries
    uint32_t bl_n, bl_any, bl_rank[ECNE_BIGK], bl_row[ECNE_BIGK], bl_nev[ECNE_BIGK], bl_deg[ECNE_BIGK], bl_base[ECNE_BIGK];
    uint32_t bl_tmp[8];
    uint32_t hasbig;
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
    for (uint32_t k = k0; k < k1; ++k) {
        const uint32_t v = J.colC[k];
        const uint8_t f = J.flags[v];
        if (f & 1) continue;
        if (!cnt) u = v;
        ++cnt;
        if (!(f & 2)) notknown = true;
        const uint32_t a = (uint32_t)J.abz[v];   // -1 (no group) is the largest value
        amin = a < amin ? a : amin;
        amax = a > amax ? a : amax;
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
__device__ __forceinline__ void expand_event(const Job& J, ChunkShared& S, uint32_t v, uint32_t a, uint32_t b0, bool multi) {
    const uint32_t f0 = J.fo_ptr[v], f1 = J.fo_ptr[v + 1];
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
    // the workgroup's minima for big target rows go to best[]; the slots are left empty again
    const uint32_t nb_rows = J.nBigRows < ECNE_BIGTAB ? J.nBigRows : ECNE_BIGTAB;
    for (uint32_t i = threadIdx.x; i < nb_rows; i += ECNE_WG) {
        const uint32_t j = S.bt[i];
        if (j != 0xFFFFFFFFu) { atomicMin(&J.best[J.bigrows[i]], j); S.bt[i] = 0xFFFFFFFFu; }
    }
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

// ------------------------------------------------------------------------------------ job barrier
// A job (one constraint system) is run by J.nwg co-resident workgroups: workgroup 0 (the "master")
// executes everything whose order matters (P1, P2, the queue, the decisions of P3, P5, all REQUEUEs);
// the others join for the row-parallel passes of the whole-system sweeps P3 / P4, the setup and the
// verdict count. They meet at this barrier: sense-reversing counter, agent-scope release before
// arriving (writes back this XCD's dirty L2 lines) and agent-scope acquire after leaving (drops
// stale L1/L2 lines) — per-XCD L2s are not coherent with each other on MI355X. The last arriver
// snapshots the job's error word, so every workgroup leaves with the SAME view of it and takes the
// same branch. Spins are bounded.
__device__ __forceinline__ uint32_t my_xcc_id() {
    // HW_REG_XCC_ID (hwreg 20), bits [3:0]: which of the 8 XCDs this wave runs on
    return __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20) & 7u;
}

// What a workgroup remembers between job barriers (LDS): the generation it waits for next and, once the
// first barrier of the launch has established them, its XCD's member count and the number of XCDs in use
// -- so that a barrier costs one atomic per level and one polled word, no other memory round trips.
struct BarLocal { unsigned gen, members, nxcd, ready; };
__device__ __forceinline__ BarLocal& bar_local() {
    __shared__ BarLocal b;
    return b;
}
__device__ __forceinline__ void job_barrier_init() {   // thread 0, once per launch (the device words are zeroed by the host)
    BarLocal& b = bar_local();
    b.gen = 0; b.members = 0; b.nxcd = 0; b.ready = 0;
}

__device__ int job_barrier(const Job& J, int* s_err) {
    __syncthreads();
    if (threadIdx.x == 0) {
        Counters* c = J.ctr;
        if (J.nwg == 1) {
            *s_err = __hip_atomic_load(&c->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            // XCD-hierarchical: workgroups of one XCD share its L2, so only the last of them to arrive
            // (the XCD leader) pays for the agent-scope release (L2 write-back) before arriving at
            // the top-level counter; everybody waits on one generation word and then drops its stale
            // L1 lines. The first barrier of a launch is flat and establishes the XCD membership.
            // The generation word carries the generation in its upper bits and "an error was raised" in
            // bit 0, so the waiters learn both from the one word they poll.
            BarLocal& b = bar_local();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // my stores have reached my XCD's L2
            const unsigned g = b.gen;
            const bool hier = b.ready != 0;
            bool arrive_top = true;
            const unsigned x = my_xcc_id();
            if (hier) {
                const unsigned a = __hip_atomic_fetch_add(&c->xcd_count[x][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (a == b.members - 1)
                    __hip_atomic_store(&c->xcd_count[x][0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else arrive_top = false;
            } else {
                __hip_atomic_fetch_add(&c->xcd_members[x], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (arrive_top) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const unsigned expect = hier ? b.nxcd : J.nwg;
                const unsigned arrived = __hip_atomic_fetch_add(&c->bar_count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (arrived == expect - 1) {
                    if (!hier) {
                        unsigned na = 0;
                        for (int i = 0; i < 8; ++i)
                            na += __hip_atomic_load(&c->xcd_members[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
                        __hip_atomic_store(&c->n_xcd_active, na, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    const int e = __hip_atomic_load(&c->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&c->bar_count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&c->bar_gen, ((g + 1u) << 1) | (e != 0 ? 1u : 0u), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            unsigned spins = 0, w;
            while (((w = __hip_atomic_load(&c->bar_gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 1) == g) {
                __builtin_amdgcn_s_sleep(8);
                if (++spins > (1u << 28)) { raise(J, K_ECAPACITY); w = 1; break; }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            b.gen = g + 1;
            if (!hier) {   // the first barrier of the launch just completed: remember the XCD layout
                b.members = __hip_atomic_load(&c->xcd_members[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                b.nxcd = __hip_atomic_load(&c->n_xcd_active, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                b.ready = 1;
            }
            *s_err = (w & 1u) ? __hip_atomic_load(&c->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
        }
    }
    __syncthreads();
    return *s_err;
}

// ----------------------------------------------------------------- multi-workgroup queue round
// Same round as in queue_phase_chunked, but executed by ALL workgroups of the job on a window of up
// to nwg * 512 * 2 queue entries — for the thousand-row-wide frontiers of large circuits. Global
// thread g owns ranks g*rpl .. g*rpl + rpl - 1. Cross-workgroup steps use job_barrier (6 per round)
// and two job-wide scans; everything a lane needs later (its rows, its events) it produced itself,
// except cand[] / best[] / inq[] / wmark, which are read after a barrier. Returns nonzero on error.
__device__ uint32_t team_exclusive_scan(const Job& J, ChunkShared& S, uint32_t wgrank, uint32_t x, uint32_t buf,
                                        uint32_t* total, int* s_err, int* err_out) {
    uint32_t wgtot;
    const uint32_t local = wg_exclusive_scan(x, S.scan, &wgtot);
    if (threadIdx.x == 0) __hip_atomic_store(&J.ctr->q_part[buf][wgrank], wgtot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *err_out = job_barrier(J, s_err);
    if (threadIdx.x < J.nwg) S.bases[threadIdx.x] = ld_agent(&J.ctr->q_part[buf][threadIdx.x]);
    __syncthreads();
    uint32_t pre = 0, tot = 0;
    for (uint32_t i = 0; i < J.nwg; ++i) { const uint32_t v = S.bases[i]; if (i < wgrank) pre += v; tot += v; }
    __syncthreads();
    *total = tot;
    return pre + local;
}

// team scan that also works for a single workgroup (no job barrier needed then)
__device__ uint32_t team_exclusive_scan_any(const Job& J, ChunkShared& S, uint32_t wgrank, uint32_t x, uint32_t* total,
                                            int* s_err, int* err_out) {
    if (J.nwg == 1) { *err_out = 0; return wg_exclusive_scan(x, S.scan, total); }
    return team_exclusive_scan(J, S, wgrank, x, 0, total, s_err, err_out);
}

__device__ __noinline__ int queue_round_multi(const Job& J, ChunkShared& S, uint32_t wgrank, uint32_t head, uint32_t tail,
                                             uint32_t n, LaneCtr& C, uint32_t& my_pops, uint32_t& my_nnz, int* s_err,
                                             uint32_t* out_c, uint32_t* out_tail) {
    const int tid = threadIdx.x, lane = lane_id(), w = wave_id();
    Counters* const ctr = J.ctr;
    const uint32_t T = J.nwg * ECNE_WG, g = wgrank * ECNE_WG + tid;
    const uint32_t rpl = (n + T - 1) / T;               // <= 2 by the caller's choice of n
    const uint32_t r0 = g * rpl;
    uint32_t row[2], shape[2], xv[2];
    uint32_t live = 0, noop = 0, noop_b = 0;
    int err;
    unsigned long long mt_last = wall_clock64();
#ifdef ECNE_FINE_TICKS
#define MTICK(slot) do { if (g == 0) { unsigned long long t_ = wall_clock64(); S.mt[slot] += t_ - mt_last; mt_last = t_; } } while (0)
#else
#define MTICK(slot) do { } while (0)
#endif
#pragma unroll
    for (uint32_t sl = 0; sl < 2; ++sl) {
        row[sl] = 0; shape[sl] = 0; xv[sl] = 0;
        if (sl < rpl && r0 + sl < n) {
            row[sl] = J.queue[(head + r0 + sl) & J.qmask];
            const RowInfo ri = J.rinfo[row[sl]];
            shape[sl] = ri.shape;
            xv[sl] = ri.x;
            if (!J.solved[row[sl]]) live |= 1u << sl;
        }
    }
    if (tid == 0) S.cut = 0xFFFFFFFFu;
    __syncthreads();
    // ---- mark (write sets)
#pragma unroll
    for (uint32_t sl = 0; sl < 2; ++sl) {
        if (sl >= rpl || r0 + sl >= n) continue;
        const uint32_t rank = r0 + sl;
        if (!(live & (1u << sl))) continue;
        if (shape[sl] & SH_BIG) {   // plain long rows ride along, handled by this workgroup as a whole (see big_rows_*)
            if (!big_plain(shape[sl]) || !big_register(S, row[sl], rank)) atomicMin(&S.cut, rank);
            continue;
        }
        const RowInfo ri = J.rinfo[row[sl]];
        bool nb = false;
        if (row_is_noop(J, row[sl], ri, nb)) { noop |= 1u << sl; if (nb) noop_b |= 1u << sl; continue; }
        for_row_sets(J, row[sl], shape[sl], xv[sl], [&](uint32_t v, uint32_t rd, uint32_t wr) {
            if (wr & 1) atomicMin(&J.wmarkU[v], rank);
            if (wr & 2) atomicMin(&J.wmarkB[v], rank);
        });
    }
    __syncthreads();
    if (S.bl_any) big_rows_mark(J, S);
    if ((err = job_barrier(J, s_err))) return err;
    MTICK(0);
    // ---- check
#pragma unroll
    for (uint32_t sl = 0; sl < 2; ++sl) {
        if (sl >= rpl || r0 + sl >= n || !(live & (1u << sl)) || (shape[sl] & SH_BIG)) continue;
        const uint32_t rank = r0 + sl;
        bool blocked = false;
        if (noop & (1u << sl)) {
            if (noop_b & (1u << sl))
                for (uint32_t k = J.rpC[row[sl]]; k < J.rpC[row[sl] + 1]; ++k)
                    if (ld_agent(&J.wmarkB[J.colC[k]]) < rank) blocked = true;
        } else {
            for_row_sets(J, row[sl], shape[sl], xv[sl], [&](uint32_t v, uint32_t rd, uint32_t wr) {
                    if ((rd | wr) & 1) {
                        const uint32_t m = ld_agent(&J.wmarkU[v]);
                        if (m < rank) blocked = true; else if (m > rank && m != 0xFFFFFFFFu) atomicMin(&S.cut, m);
                    }
                    if ((rd | wr) & 2) {
                        const uint32_t m = ld_agent(&J.wmarkB[v]);
                        if (m < rank) blocked = true; else if (m > rank && m != 0xFFFFFFFFu) atomicMin(&S.cut, m);
                    }
                });
        }
        if (blocked) atomicMin(&S.cut, rank);
    }
    if (S.bl_any) big_rows_check(J, S);
    // one global update per workgroup (thousands of lanes on one word would serialise)
    __syncthreads();
    if (tid == 0 && S.cut != 0xFFFFFFFFu) atomicMin(&ctr->q_cut, S.cut);
    if ((err = job_barrier(J, s_err))) return err;
    MTICK(1);
    uint32_t c = ld_agent(&ctr->q_cut);         // >= 1 (the master checked that rank 0 is not a big row)
    if (c > n) c = n;                             // nobody blocked: the whole window commits
    // ---- unmark, tag, execute my ranks below the cut
    if (S.bl_any) big_rows_unmark(J, S);
    uint32_t nev[2], mycand = 0, bigsl = 0;
#pragma unroll
    for (uint32_t sl = 0; sl < 2; ++sl) {
        nev[sl] = 0;
        if (sl >= rpl || r0 + sl >= n) continue;
        if ((live & (1u << sl)) && !(shape[sl] & SH_BIG) && !(noop & (1u << sl)))
            for_row_sets(J, row[sl], shape[sl], xv[sl], [&](uint32_t v, uint32_t rd, uint32_t wr) {
                if (wr & 1) J.wmarkU[v] = 0xFFFFFFFFu;
                if (wr & 2) J.wmarkB[v] = 0xFFFFFFFFu;
            });
        if (r0 + sl >= c) continue;
        // rank tags must fit inq's 16 bits: multi rounds tag with the rank's low part plus a flag that
        // the row is in the current prefix; the exact rank is recovered from prank[] (see below)
        J.inq[row[sl]] = (uint16_t)2;
        my_pops++;
        my_nnz += (J.rpA[row[sl] + 1] - J.rpA[row[sl]]) + (J.rpB[row[sl] + 1] - J.rpB[row[sl]]) + (J.rpC[row[sl] + 1] - J.rpC[row[sl]]);
        J.prank[row[sl]] = r0 + sl;      // rank of a row being popped in this round
        if (live & (1u << sl)) {
            if (noop & (1u << sl)) { if ((shape[sl] & SH_R4_T) && (shape[sl] & SH_R4_T2)) J.flip3[row[sl]] ^= 1; }
            else if (shape[sl] & SH_BIG) { bigsl |= 1u << sl; continue; }   // executed below by the whole workgroup
            else exec_row_lane(J, row[sl], J.evbuf + (size_t)(r0 + sl) * ECNE_EVCAP, nev[sl], C);
        }
        uint32_t* ev = J.evbuf + (size_t)(r0 + sl) * ECNE_EVCAP;
        for (uint32_t e = 0; e < nev[sl]; ++e) mycand += J.fo_ptr[ev[e] + 1] - J.fo_ptr[ev[e]];
        ev[ECNE_EVCAP - 1] = nev[sl];    // for the sequential replay fallback
    }
    if (S.bl_any) {   // (uniform per workgroup) long rows of the prefix: execute, then candidate offsets of their events
        big_rows_exec(J, S, c, wgrank);
        for (uint32_t k = 0; k < ECNE_BIGK; ++k) {
            if (S.bl_rank[k] >= c) continue;
            const uint32_t* ev = big_ev(J, wgrank, k);
            uint32_t* off = big_off(J, wgrank, k);
            const uint32_t ne = S.bl_nev[k];
            uint32_t run = 0;
            for (uint32_t eb = 0; eb < ne; eb += ECNE_WG) {          // (uniform trip count)
                const uint32_t e = eb + tid;
                uint32_t d = 0;
                if (e < ne) { const uint32_t v = ev[e]; d = J.fo_ptr[v + 1] - J.fo_ptr[v]; }
                uint32_t tot;
                const uint32_t o = wg_exclusive_scan(d, S.scan, &tot);
                if (e < ne) off[e] = run + o;
                run += tot;
            }
            if (tid == 0) {
                S.bl_deg[k] = run;
                // the rank's regular slot only says where the events are (for the sequential replay fallback)
                uint32_t* slot = J.evbuf + (size_t)S.bl_rank[k] * ECNE_EVCAP;
                slot[ECNE_EVCAP - 1] = 0x80000000u | (wgrank * ECNE_BIGK + k);
                slot[ECNE_EVCAP - 2] = ne;
            }
        }
        __syncthreads();
#pragma unroll
        for (uint32_t sl = 0; sl < 2; ++sl)
            if (bigsl & (1u << sl)) {
                const int k = big_slot_of(S, r0 + sl);
                nev[sl] = S.bl_nev[k];
                mycand += S.bl_deg[k];
            }
    }
    uint32_t M;
    const uint32_t cbase = team_exclusive_scan(J, S, wgrank, mycand, 0, &M, s_err, &err);
    MTICK(2);
    if (err) return err;
    if (M > J.candcap) {
        // a variable with a huge fan-out: the master replays all events sequentially (rare)
        if (wgrank == 0) {
            if (w == 0) {
                QState qq;
                qq.head = 0; qq.tail = tail; qq.evout = nullptr; qq.nev = 0; qq.emit = 0;
                // event counts live in the executing lanes' registers: recount from the fan-out lists is not
                // possible, so each rank's count was also stored behind its events (slot ECNE_EVCAP - 1)
                for (uint32_t r = 0; r < c; ++r) {
                    const uint32_t rr = J.queue[(head + r) & J.qmask];
                    if (lane == 0) J.inq[rr] = 0;
                    wg_fence();
                    uint32_t ne = J.evbuf[(size_t)r * ECNE_EVCAP + ECNE_EVCAP - 1];
                    const uint32_t* evs = J.evbuf + (size_t)r * ECNE_EVCAP;
                    if (ne & 0x80000000u) {   // a long row: its events are in the pool
                        evs = J.bigpool + (size_t)(ne & 0x7FFFFFFFu) * J.bigstride;
                        ne = J.evbuf[(size_t)r * ECNE_EVCAP + ECNE_EVCAP - 2];
                    }
                    for (uint32_t e = 0; e < ne; ++e) requeue(J, qq, evs[e]);
                }
                if (lane == 0) { ctr->q_tail_out = qq.tail; ctr->q_c_out = c; ctr->q_cut = 0xFFFFFFFFu; }
            }
            __syncthreads();
        }
        if ((err = job_barrier(J, s_err))) return err;
        *out_c = c;
        *out_tail = ld_agent(&ctr->q_tail_out);
        return 0;
    }
    // ---- expansion of my own events: candidate index = cbase + running offset
    {
        if (tid == 0) S.nbigev = 0;
        __syncthreads();
        uint32_t j = cbase;
#pragma unroll
        for (uint32_t sl = 0; sl < 2; ++sl) {
            if (sl >= rpl || r0 + sl >= c) continue;
            const uint32_t a = r0 + sl;
            if (bigsl & (1u << sl)) {   // a long row's events are expanded by the whole workgroup, below
                const int k = big_slot_of(S, a);
                S.bl_base[k] = j;
                j += S.bl_deg[k];
                continue;
            }
            const uint32_t* ev = J.evbuf + (size_t)a * ECNE_EVCAP;
            for (uint32_t e = 0; e < nev[sl]; ++e) {
                const uint32_t v = ev[e];
                expand_event(J, S, v, a, j, true);
                j += J.fo_ptr[v + 1] - J.fo_ptr[v];
            }
        }
        if (S.bl_any) {
            __syncthreads();
            for (uint32_t k = 0; k < ECNE_BIGK; ++k) {
                if (S.bl_rank[k] >= c) continue;
                const uint32_t* ev = big_ev(J, wgrank, k);
                const uint32_t* off = big_off(J, wgrank, k);
                for (uint32_t e = tid; e < S.bl_nev[k]; e += ECNE_WG) expand_event(J, S, ev[e], S.bl_rank[k], S.bl_base[k] + off[e], true);
            }
        }
        expand_big_events(J, S, true);
    }
    if ((err = job_barrier(J, s_err))) return err;
    MTICK(3);
    // ---- the prefix rows leave the queue (tags no longer needed); then winners in candidate order
#pragma unroll
    for (uint32_t sl = 0; sl < 2; ++sl)
        if (sl < rpl && r0 + sl < c) J.inq[row[sl]] = 0;
    const uint32_t per = (M + T - 1) / T;
    const uint32_t j0 = g * per < M ? g * per : M, j1 = (g + 1) * per < M ? (g + 1) * per : M;
    uint32_t nwin = 0;
    for (uint32_t j = j0; j < j1; ++j) {
        const uint32_t cw = J.cand[j];
        const uint32_t t = cw & 0x7FFFFFFFu;
        const bool win = (cw & 0x80000000u) && ld_agent(&J.best[t]) == j;
        J.cand[j] = t | (win ? 0x80000000u : 0u);
        nwin += win;
    }
    uint32_t W;
    const uint32_t wbase = team_exclusive_scan(J, S, wgrank, nwin, 1, &W, s_err, &err);
    MTICK(4);
    if (err) return err;
    {
        uint32_t o = tail + wbase;
        for (uint32_t j = j0; j < j1; ++j) {
            const uint32_t cw = J.cand[j];
            const uint32_t t = cw & 0x7FFFFFFFu;
            if (cw & 0x80000000u) { J.queue[o & J.qmask] = t; J.inq[t] = 1; ++o; }
            J.best[t] = 0xFFFFFFFFu;
        }
    }
    if (g == 0) ctr->q_cut = 0xFFFFFFFFu;     // ready for the next multi round
    if (tid == 0) big_reset(S);
    if ((err = job_barrier(J, s_err))) return err;
    MTICK(5);
    *out_c = c;
    *out_tail = tail + W;
    return 0;
}

// ------------------------------------------------------------------------------- wavefront round
// The same round as in queue_phase_chunked for a window of at most 64 queue entries, executed by ONE
// wavefront (lane = rank) without a single workgroup barrier: narrow dependency levels (a dozen rows
// wide) are the bulk of the rounds of a deep circuit and a workgroup round costs them ~20 us of barriers
// and idle lanes. Write-marks use the first 1024 slots of the LDS hash table (wiped afterwards), the
// REQUEUE events are resolved with wave scans. Returns the number of committed rows, or 0xFFFFFFFF
// without having touched anything when the window starts with a live long row (the caller's general path
// takes it). Wave 0 only, all 64 lanes.
#define ECNE_WSLOTS 1024
__device__ __forceinline__ void lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ uint32_t wave_excl_scan(uint32_t x, uint32_t* total) {
    uint32_t incl = x;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(incl, d, 64); if (lane_id() >= d) incl += y; }
    *total = __shfl(incl, 63, 64);
    return incl - x;
}
__device__ __forceinline__ uint32_t wave_min(uint32_t x) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { const uint32_t y = __shfl_xor(x, d, 64); x = y < x ? y : x; }
    return x;
}
__device__ __noinline__ uint32_t queue_round_wave(const Job& J, ChunkShared& S, uint32_t head, uint32_t tail, uint32_t n,
                                                  LaneCtr& C, uint32_t& my_pops, uint32_t& my_nnz, uint32_t* out_tail,
                                                  unsigned long long* n_fallback) {
    const int lane = lane_id();
    const uint32_t rank = (uint32_t)lane;
    const bool mine = rank < n;
    uint32_t row = 0, shape = 0, xv = 0;
    bool live = false;
    if (mine) {
        row = J.queue[(head + rank) & J.qmask];
        const RowInfo ri = J.rinfo[row];
        shape = ri.shape;
        xv = ri.x;
        live = !J.solved[row];
    }
    const uint64_t bigm = __ballot(mine && live && (shape & SH_BIG));
    if (bigm & 1ull) return 0xFFFFFFFFu;
    uint32_t cut = bigm ? (uint32_t)(__ffsll((long long)bigm) - 1) : n;    // a long row ends the prefix
    // ---- mark (LDS hash, slots [0, ECNE_WSLOTS)); the access set is cached for the check
    bool noop = false, noop_b = false;
    uint32_t acnt = 0;
    auto wmark = [&](uint32_t v, uint32_t cls) {
        const uint32_t key = 1u + 2u * v + cls;
        uint32_t sl = (key * 2654435761u) >> (32 - 10);
        for (int probe = 0; probe < ECNE_WSLOTS; ++probe) {
            const uint32_t k = atomicCAS(&S.hkey[sl], 0u, key);
            if (k == 0u || k == key) { atomicMin(&S.hrank[sl], rank); return; }
            sl = (sl + 1) & (ECNE_WSLOTS - 1);
        }
    };
    auto wlook = [&](uint32_t v, uint32_t cls) -> uint32_t {
        const uint32_t key = 1u + 2u * v + cls;
        uint32_t sl = (key * 2654435761u) >> (32 - 10);
        for (int probe = 0; probe < ECNE_WSLOTS; ++probe) {
            const uint32_t k = S.hkey[sl];
            if (k == key) return S.hrank[sl];
            if (k == 0u) return 0xFFFFFFFFu;
            sl = (sl + 1) & (ECNE_WSLOTS - 1);
        }
        return 0xFFFFFFFFu;
    };
    uint32_t nmarks = 0;
    if (mine && live && rank < cut) {
        const RowInfo ri = J.rinfo[row];
        bool nb = false;
        if (row_is_noop(J, row, ri, nb)) {
            noop = true;
            if (nb) {
                noop_b = true;
                for (uint32_t k = J.rpC[row]; k < J.rpC[row + 1]; ++k) {
                    if (acnt < ECNE_ASET) S.aset[lane][acnt] = J.colC[k] | (2u << 28);
                    ++acnt;
                }
            }
        } else {
            for_row_sets(J, row, shape, xv, [&](uint32_t v, uint32_t rd, uint32_t wr) {
                if (acnt < ECNE_ASET) S.aset[lane][acnt] = v | (rd << 28) | (wr << 30);
                ++acnt;
                nmarks += (wr & 1) + ((wr >> 1) & 1);
            });
        }
    }
    // a window that would load the table beyond a quarter is cut down to the rows that fit (rank 0 always does:
    // a small row has at most 2 * 64 marks)
    {
        uint32_t tot;
        const uint32_t before = wave_excl_scan(nmarks, &tot);
        if (tot > ECNE_WSLOTS / 4) {
            const uint64_t over = __ballot(before + nmarks > ECNE_WSLOTS / 4);
            const uint32_t first = over ? (uint32_t)(__ffsll((long long)over) - 1) : n;
            if (first < cut) cut = first < 1 ? 1 : first;
        }
    }
    if (mine && live && rank < cut && !noop) {
        if (acnt <= ECNE_ASET) {
            for (uint32_t i = 0; i < acnt; ++i) { const uint32_t e = S.aset[lane][i]; if ((e >> 30) & 1) wmark(e & 0x0FFFFFFFu, 0); if (e >> 31) wmark(e & 0x0FFFFFFFu, 1); }
        } else {
            for_row_sets(J, row, shape, xv, [&](uint32_t v, uint32_t rd, uint32_t wr) { if (wr & 1) wmark(v, 0); if (wr & 2) wmark(v, 1); });
        }
    }
    lds_fence();
    // ---- check
    uint32_t mycut = 0xFFFFFFFFu;
    if (mine && live && rank < cut) {
        bool blocked = false;
        auto test = [&](uint32_t v, uint32_t rd, uint32_t wr) {
            if ((rd | wr) & 1) { const uint32_t m = wlook(v, 0); if (m < rank) blocked = true; else if (m > rank && m < mycut) mycut = m; }
            if ((rd | wr) & 2) { const uint32_t m = wlook(v, 1); if (m < rank) blocked = true; else if (m > rank && m < mycut) mycut = m; }
        };
        if (noop) {
            if (noop_b) {
                if (acnt <= ECNE_ASET) { for (uint32_t i = 0; i < acnt; ++i) if (wlook(S.aset[lane][i] & 0x0FFFFFFFu, 1) < rank) blocked = true; }
                else for (uint32_t k = J.rpC[row]; k < J.rpC[row + 1]; ++k) if (wlook(J.colC[k], 1) < rank) blocked = true;
            }
        } else if (acnt <= ECNE_ASET) {
            for (uint32_t i = 0; i < acnt; ++i) { const uint32_t e = S.aset[lane][i]; test(e & 0x0FFFFFFFu, (e >> 28) & 3u, e >> 30); }
        } else for_row_sets(J, row, shape, xv, test);
        if (blocked) mycut = rank;
    }
    {
        const uint32_t m = wave_min(mycut);
        if (m < cut) cut = m;
    }
    const uint32_t c = cut;    // >= 1
    lds_fence();
    for (uint32_t i = lane; i < ECNE_WSLOTS; i += 64) { S.hkey[i] = 0; S.hrank[i] = 0xFFFFFFFFu; }
    // ---- tag, execute
    uint32_t nev = 0;
    if (mine && rank < c) {
        J.inq[row] = (uint16_t)(rank + 2);
        my_pops++;
        my_nnz += (J.rpA[row + 1] - J.rpA[row]) + (J.rpB[row + 1] - J.rpB[row]) + (J.rpC[row + 1] - J.rpC[row]);
    }
    wg_fence();
    if (mine && rank < c && live) {
        if (noop) { if ((shape & SH_R4_T) && (shape & SH_R4_T2)) J.flip3[row] ^= 1; }
        else exec_row_lane(J, row, J.evbuf + (size_t)rank * ECNE_EVCAP, nev, C);
    }
    wg_fence();
    // ---- REQUEUE resolution in sequential order (rank, emission index), see resolve_pushes
    uint32_t new_tail = tail;
    uint32_t Nev;
    wave_excl_scan(nev, &Nev);
    if (Nev) {
        const uint32_t* ev = J.evbuf + (size_t)rank * ECNE_EVCAP;
        uint32_t deg = 0;
        for (uint32_t e = 0; e < nev; ++e) deg += J.fo_ptr[ev[e] + 1] - J.fo_ptr[ev[e]];
        uint32_t M;
        const uint32_t cbase = wave_excl_scan(deg, &M);
        if (M > ECNE_CANDCAP) {
            // (a variable with a huge fan-out) replay the events one by one, ranks leaving the queue in order
            if (n_fallback) (*n_fallback)++;
            QState qq;
            qq.head = 0; qq.tail = tail; qq.evout = nullptr; qq.nev = 0; qq.emit = 0;
            for (uint32_t r = 0; r < c; ++r) {
                const uint32_t rr = J.queue[(head + r) & J.qmask];
                if (lane == 0) J.inq[rr] = 0;
                wg_fence();
                const uint32_t ne = __shfl(nev, (int)r, 64);
                for (uint32_t e = 0; e < ne; ++e) requeue(J, qq, J.evbuf[(size_t)r * ECNE_EVCAP + e]);
            }
            *out_tail = qq.tail;
            return c;
        }
        // expansion: my events, in emission order; long fan-out lists are shared out across the lanes afterwards
        auto cand1 = [&](uint32_t t, uint32_t j, uint32_t a) {
            const uint32_t st = J.inq[t];
            const bool elig = st == 0 || (st >= 2 && st - 2 <= a);
            J.cand[j] = t | (elig ? 0x80000000u : 0u);
            if (elig && ld_agent(&J.best[t]) > j) atomicMin(&J.best[t], j);
        };
        uint32_t nlong = 0, lv = 0, lb = 0;      // at most one long-fan-out event per lane is deferred
        {
            uint32_t j = cbase;
            for (uint32_t e = 0; e < nev; ++e) {
                const uint32_t v = ev[e];
                const uint32_t f0 = J.fo_ptr[v], f1 = J.fo_ptr[v + 1];
                if (f1 - f0 > 64 && !nlong) { nlong = 1; lv = v; lb = j; }
                else for (uint32_t k = f0; k < f1; ++k) cand1(J.fo_rows[k], j + (k - f0), rank);
                j += f1 - f0;
            }
        }
        for (uint64_t lm = __ballot(nlong != 0); lm; lm &= lm - 1) {
            const int src = __ffsll((long long)lm) - 1;
            const uint32_t v = __shfl(lv, src, 64), b0 = __shfl(lb, src, 64);
            const uint32_t f0 = J.fo_ptr[v], f1 = J.fo_ptr[v + 1];
            for (uint32_t k = f0 + lane; k < f1; k += 64) cand1(J.fo_rows[k], b0 + (k - f0), (uint32_t)src);
        }
        wg_fence();
        // winners, in candidate order
        for (uint32_t jb = 0; jb < M; jb += 64) {
            const uint32_t j = jb + lane;
            uint32_t t = 0;
            bool win = false;
            if (j < M) {
                const uint32_t cw = J.cand[j];
                t = cw & 0x7FFFFFFFu;
                win = (cw & 0x80000000u) && ld_agent(&J.best[t]) == j;
            }
            const uint64_t wm = __ballot(win);
            if (win) { J.queue[(new_tail + (uint32_t)__popcll(wm & lanes_below())) & J.qmask] = t; J.inq[t] = 1; }
            new_tail += (uint32_t)__popcll(wm);
        }
        // (best[] is reset only now: every candidate above was judged against the same minima)
        wg_fence();
        for (uint32_t j = lane; j < M; j += 64) J.best[J.cand[j] & 0x7FFFFFFFu] = 0xFFFFFFFFu;
        wg_fence();
    }
    // rows of the prefix that nobody re-queued are out of the queue now
    if (mine && rank < c && J.inq[row] >= 2) J.inq[row] = 0;
    wg_fence();
    *out_tail = new_tail;
    return c;
}

// The whole QUEUE phase (:805-1349) as the master workgroup sees it. q is kept identical in every thread.
// A single-workgroup round examines up to ECNE_RPL * ECNE_WG queue entries; lane t owns the consecutive ranks
// t*rpl .. t*rpl + rpl - 1, so that per-lane totals scanned once give rank-ordered offsets.
#define ECNE_RPL 4
#ifndef ECNE_WGROW
#define ECNE_WGROW 2
#endif
#ifndef ECNE_WMIN
#define ECNE_WMIN 64
#endif
#ifndef ECNE_MULTI_MIN
#define ECNE_MULTI_MIN 128    // queued rows from which a round runs on all workgroups of the job (measured optimum)
#endif
__device__ __noinline__ void queue_phase_chunked(const Job& J, QState& q, ChunkShared& S, unsigned long long* hits,
                                    unsigned long long& steps, unsigned long long& nuniq,
                                    unsigned long long& pops, unsigned long long& pop_nnz, int* s_err) {
    const int tid = threadIdx.x, lane = lane_id(), w = wave_id();
    const unsigned long long pop_cap = 4096ull + 64ull * (J.rpA[J.nC] + J.rpB[J.nC] + J.rpC[J.nC]);
    if (tid < 12) S.acc[tid] = 0;
    for (uint32_t i = tid; i < ECNE_HSLOTS; i += ECNE_WG) { S.hkey[i] = 0; S.hrank[i] = 0xFFFFFFFFu; }
    if (tid == 0) { S.small_ovf = 0; big_reset(S); }
    unsigned long long qt_last = wall_clock64();
#ifdef ECNE_FINE_TICKS
#define QTICK(slot) do { if (tid == 0) { unsigned long long t_ = wall_clock64(); S.qt[slot] += t_ - qt_last; qt_last = t_; } } while (0)
#else
#define QTICK(slot) do { } while (0)
#endif
    LaneCtr C;
    C.steps = C.nuniq = 0;
    for (int i = 0; i < 8; ++i) C.hits[i] = 0;
    uint32_t my_pops = 0, my_nnz = 0;
    unsigned long long pops_total = pops;
    uint32_t round = 0, burst = 0, next_burst = 16, window = ECNE_RPL * ECNE_WG;
    uint32_t mwindow = 16384;        // window of multi-workgroup rounds (adaptive like `window`)
    bool helpers_released = false;   // an error seen at a job barrier has already sent the helpers home
    __syncthreads();
    while (q.head != q.tail) {
        // the error word is polled every 8th round (a raised error only has to stop the solve soon)
        if ((round++ & 7u) == 0 && wg_error(J, s_err)) break;
        if (pops_total > pop_cap) { raise(J, K_ECAPACITY); break; }
        const uint32_t avail = q.tail - q.head;
        if (burst) {
            // The last chunk round committed only a handful of rows (a dependency chain): pop the next
            // `burst` rows strictly sequentially on wave 0 (cheaper per pop than a round), then look again.
            if (w == 0) {
                QState qq = q;
                qq.evout = nullptr; qq.nev = 0; qq.emit = 0;
                unsigned long long st = 0, nu = 0, ht[16], pn = 0;
                for (int i = 0; i < 16; ++i) ht[i] = 0;
                uint32_t done = 0;
                while (done < burst && qq.head != qq.tail && !J.ctr->error) {
                    const uint32_t rr = J.queue[qq.head & J.qmask];
                    qq.head++;
                    if (lane == 0) J.inq[rr] = 0;
                    wg_fence();
                    ++done;
                    pn += (J.rpA[rr + 1] - J.rpA[rr]) + (J.rpB[rr + 1] - J.rpB[rr]) + (J.rpC[rr + 1] - J.rpC[rr]);
                    if (!J.solved[rr]) exec_row(J, qq, rr, ht, st, nu);
                }
                if (lane == 0) {
                    S.acc[0] += st; S.acc[1] += nu;
                    for (int i = 0; i < 8; ++i) S.acc[2 + i] += ht[i];
                    S.acc[10] += done; S.acc[11] += pn;
                    S.head = qq.head; S.tail = qq.tail; S.nbig = done;
                }
            }
            __syncthreads();
            q.head = S.head; q.tail = S.tail;
            pops_total += S.nbig;
            burst = 0;
            __syncthreads();
            QTICK(6);
            continue;
        }
        // adaptive window: examining rows that end up behind the cut is wasted work, so the window
        // follows the prefix lengths actually achieved (shrinks on short prefixes, doubles on full ones)
        const uint32_t n = avail < window ? avail : window;
        if (n <= 64) {
            // a narrow level: the whole round on wavefront 0, no workgroup barrier inside (queue_round_wave)
            if (w == 0) {
                uint32_t nt = q.tail;
                const uint32_t cw = queue_round_wave(J, S, q.head, q.tail, n, C, my_pops, my_nnz, &nt, &hits[15]);
                if (lane == 0) { S.nbig = cw; S.tail = nt; }
            }
            __syncthreads();
            const uint32_t cw = S.nbig, ntw = S.tail;
            __syncthreads();
            if (cw != 0xFFFFFFFFu) {
                q.head += cw;
                q.tail = ntw;
                pops_total += cw;
                hits[13]++;
                if (cw < 8 && avail < 64) { burst = next_burst; if (next_burst < 512) next_burst *= 2; }
                if (cw == n) window = (window * ECNE_WGROW < ECNE_RPL * ECNE_WG) ? window * ECNE_WGROW : ECNE_RPL * ECNE_WG;
                else if (cw < n / 4) { uint32_t wn = 4 * cw; window = wn < ECNE_WMIN ? ECNE_WMIN : wn; }
                else next_burst = 16;
                QTICK(6);
                continue;
            }
            // (the window starts with a live long row: the general path below takes this round)
        }
        const uint32_t rpl = (n + ECNE_WG - 1) / ECNE_WG;          // rows per lane this round
        const uint32_t r0 = (uint32_t)tid * rpl;                    // my first rank
        uint32_t row[ECNE_RPL], shape[ECNE_RPL], xv[ECNE_RPL];
        uint32_t live = 0, noop = 0, noop_b = 0;                    // bit s = slot s
#pragma unroll
        for (uint32_t sl = 0; sl < ECNE_RPL; ++sl) {
            row[sl] = 0; shape[sl] = 0; xv[sl] = 0;
            if (sl < rpl && r0 + sl < n) {
                row[sl] = J.queue[(q.head + r0 + sl) & J.qmask];
                const RowInfo ri = J.rinfo[row[sl]];
                shape[sl] = ri.shape;
                xv[sl] = ri.x;
                if (!J.solved[row[sl]]) live |= 1u << sl;
            }
        }
        if (tid == 0) { S.cut = n; S.fallback = ((shape[0] & SH_BIG) && (live & 1u) && !big_plain(shape[0])) ? 1u : 0u; }
#pragma unroll
        for (uint32_t sl = 0; sl < ECNE_RPL; ++sl)   // a long row that can ride along sends the round down the general path
            if (sl < rpl && r0 + sl < n && (shape[sl] & SH_BIG) && (live & (1u << sl)) && big_plain(shape[sl])) S.hasbig = 1;
        __syncthreads();
        QTICK(0);
        if (!S.fallback && J.nwg > 1 && avail >= ECNE_MULTI_MIN && window >= ECNE_MULTI_MIN) {
            // a wide frontier: one round on all workgroups of the job (see queue_round_multi)
            const uint32_t cap_n = J.nwg * ECNE_WG * 2;
            uint32_t nm = avail < cap_n ? avail : cap_n;
            if (nm > mwindow) nm = mwindow;
            if (tid == 0) {
                J.ctr->q_cmd[1] = q.head; J.ctr->q_cmd[2] = q.tail; J.ctr->q_cmd[3] = nm;
                __hip_atomic_store(&J.ctr->q_cmd[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (job_barrier(J, s_err)) { helpers_released = true; break; }
            uint32_t cm = 0, ntm = q.tail;
            if (queue_round_multi(J, S, 0, q.head, q.tail, nm, C, my_pops, my_nnz, s_err, &cm, &ntm)) { helpers_released = true; break; }
            q.head += cm;
            q.tail = ntm;
            pops_total += cm;
            hits[13]++;
            hits[14] += 1u << 16;                 // diagnostics: multi rounds in the high half
            hits[15] += (unsigned long long)cm << 8;   // and the rows they committed
            if (cm == nm) mwindow = (mwindow * 2 < cap_n) ? mwindow * 2 : cap_n;
            else if (cm < nm / 4) {
                const uint32_t wn = 4 * cm;
                if (wn >= 4096) mwindow = wn;
                else { mwindow = 4096; window = wn < ECNE_WMIN ? ECNE_WMIN : (wn < ECNE_RPL * ECNE_WG ? wn : ECNE_RPL * ECNE_WG); }
            }
            QTICK(7);
            continue;
        }
        if (S.fallback) {
            // a big row at the queue head: popped alone. Wave 0 runs the wave-cooperative rules in emit
            // mode; the whole workgroup then resolves its REQUEUE events in order.
            const uint32_t brow = J.queue[q.head & J.qmask];
            if (tid == 0) {
                J.inq[brow] = 2;                       // being popped at rank 0
                S.acc[10] += 1;
                S.acc[11] += (J.rpA[brow + 1] - J.rpA[brow]) + (J.rpB[brow + 1] - J.rpB[brow]) + (J.rpC[brow + 1] - J.rpC[brow]);
                S.nbig = 0;
            }
            __syncthreads();
            const bool wgdone = J.solved[brow] || exec_big_row_wg(J, S, brow, J.bigev, &S.nbig);
            if (wgdone) {
                // done by the whole workgroup (or an already solved row: the pop is all that happens)
            } else if (w == 0) {
                const uint32_t rr = brow;
                QState qq;
                qq.head = q.head + 1; qq.tail = q.tail; qq.evout = J.bigev; qq.nev = 0; qq.emit = 1;
                unsigned long long st = 0, nu = 0, ht[16];
                for (int i = 0; i < 16; ++i) ht[i] = 0;
                exec_row(J, qq, rr, ht, st, nu);
                if (lane == 0) {
                    S.acc[0] += st; S.acc[1] += nu;
                    for (int i = 0; i < 8; ++i) S.acc[2 + i] += ht[i];
                    S.nbig = qq.nev;
                }
            }
            __syncthreads();
            {
                const uint32_t nt = resolve_pushes(J, S, J.bigev, false, S.nbig, (long long)q.head, 1, q.tail, &hits[15]);
                if (tid == 0 && J.inq[brow] >= 2) J.inq[brow] = 0;
                q.head += 1;
                q.tail = nt;
            }
            pops_total++;
            hits[14]++;
            if (tid == 0) S.hasbig = 0;
            __syncthreads();
            QTICK(6);
            continue;
        }
        uint32_t c;
        // ---- small round (at most one row per lane): write-marks in the LDS hash table, every lane's
        // access set cached in LDS between the two passes -- no device-memory atomics, one walk per row
        bool small = n <= ECNE_WG && !S.hasbig;
        uint32_t acnt = 0;
        if (small) {
            if ((uint32_t)tid < n) {
                const uint32_t rank = (uint32_t)tid;
                if (!(live & 1u)) { }                                     // solved row: the pop is all that happens
                else if (shape[0] & SH_BIG) atomicMin(&S.cut, rank);   // (a long row of the R2..R6 shapes, rank > 0)
                else {
                    const RowInfo ri = J.rinfo[row[0]];
                    bool nb = false;
                    if (row_is_noop(J, row[0], ri, nb)) {
                        noop |= 1u;
                        if (nb) {
                            noop_b |= 1u;
                            for (uint32_t k = J.rpC[row[0]]; k < J.rpC[row[0] + 1]; ++k) {
                                if (acnt < ECNE_ASET) S.aset[tid][acnt] = J.colC[k] | (2u << 28);
                                ++acnt;
                            }
                        }
                    } else {
                        for_row_sets(J, row[0], shape[0], xv[0], [&](uint32_t v, uint32_t rd, uint32_t wr) {
                            if (acnt < ECNE_ASET) S.aset[tid][acnt] = v | (rd << 28) | (wr << 30);
                            ++acnt;
                            if (wr & 1) hmark(S, v, 0, rank);
                            if (wr & 2) hmark(S, v, 1, rank);
                        });
                    }
                }
            }
            __syncthreads();
            QTICK(1);
            if (S.small_ovf) {   // (uniform) the table overflowed: wipe it and take the general path
                __syncthreads();
                for (uint32_t i = tid; i < ECNE_HSLOTS; i += ECNE_WG) { S.hkey[i] = 0; S.hrank[i] = 0xFFFFFFFFu; }
                if (tid == 0) { S.small_ovf = 0; S.cut = n; }
                noop = noop_b = 0;
                small = false;
                __syncthreads();
            }
        }
        if (small) {
            // ---- check against the table. Lower rank than mine: I would read (or overwrite) what an earlier
            // row writes -> blocked. Higher: that row would overwrite what I read -> the prefix is cut there.
            if ((uint32_t)tid < n && (live & 1u) && !(shape[0] & SH_BIG)) {
                const uint32_t rank = (uint32_t)tid;
                bool blocked = false;
                auto test = [&](uint32_t v, uint32_t rd, uint32_t wr) {
                    if ((rd | wr) & 1) {
                        const uint32_t m = hlook(S, v, 0);
                        if (m < rank) blocked = true; else if (m > rank && m != 0xFFFFFFFFu) atomicMin(&S.cut, m);
                    }
                    if ((rd | wr) & 2) {
                        const uint32_t m = hlook(S, v, 1);
                        if (m < rank) blocked = true; else if (m > rank && m != 0xFFFFFFFFu) atomicMin(&S.cut, m);
                    }
                };
                if (noop & 1u) {
                    if (noop_b & 1u) {
                        if (acnt <= ECNE_ASET) { for (uint32_t i = 0; i < acnt; ++i) if (hlook(S, S.aset[tid][i] & 0x0FFFFFFFu, 1) < rank) blocked = true; }
                        else for (uint32_t k = J.rpC[row[0]]; k < J.rpC[row[0] + 1]; ++k) if (hlook(S, J.colC[k], 1) < rank) blocked = true;
                    }
                } else if (acnt <= ECNE_ASET) {
                    for (uint32_t i = 0; i < acnt; ++i) { const uint32_t e = S.aset[tid][i]; test(e & 0x0FFFFFFFu, (e >> 28) & 3u, e >> 30); }
                } else for_row_sets(J, row[0], shape[0], xv[0], test);
                if (blocked) atomicMin(&S.cut, rank);
            }
            __syncthreads();
            c = S.cut;   // >= 1: rank 0 is never blocked and not big
            // ---- wipe the table; tag the rows being popped with their rank (see resolve_pushes)
            for (uint32_t i = tid; i < ECNE_HSLOTS; i += ECNE_WG) { S.hkey[i] = 0; S.hrank[i] = 0xFFFFFFFFu; }
            if ((uint32_t)tid < c) J.inq[row[0]] = (uint16_t)(tid + 2);
            __syncthreads();
            QTICK(2);
        } else {
            // ---- mark
#pragma unroll
            for (uint32_t sl = 0; sl < ECNE_RPL; ++sl) {
                if (sl >= rpl || r0 + sl >= n) continue;
                const uint32_t rank = r0 + sl;
                if (!(live & (1u << sl))) continue;
                if (shape[sl] & SH_BIG) {
                    // a plain long row rides along (marked / checked / executed by the whole workgroup, below);
                    // any other long row ends the prefix and is popped alone
                    if (!big_plain(shape[sl]) || !big_register(S, row[sl], rank)) atomicMin(&S.cut, rank);
                    continue;
                }
                const RowInfo ri = J.rinfo[row[sl]];
                bool nb = false;
                if (row_is_noop(J, row[sl], ri, nb)) { noop |= 1u << sl; if (nb) noop_b |= 1u << sl; continue; }
                // only WRITE sets are marked: the readers find write-after-read hazards themselves (below)
                for_row_sets(J, row[sl], shape[sl], xv[sl], [&](uint32_t v, uint32_t rd, uint32_t wr) {
                    if (wr & 1) atomicMin(&J.wmarkU[v], rank);
                    if (wr & 2) atomicMin(&J.wmarkB[v], rank);
                });
            }
            __syncthreads();
            if (S.bl_any) { big_rows_mark(J, S); __syncthreads(); }
            QTICK(1);
            // ---- check: blocked if an earlier rank may write state I read, or reads/writes state I may write.
            // Marks are updated with device-scope atomics (performed at L2): read them past the L1.
#pragma unroll
            for (uint32_t sl = 0; sl < ECNE_RPL; ++sl) {
                if (sl >= rpl || r0 + sl >= n || !(live & (1u << sl)) || (shape[sl] & SH_BIG)) continue;
                const uint32_t rank = r0 + sl;
                bool blocked = false;
                if (noop & (1u << sl)) {
                    if (noop_b & (1u << sl))
                        for (uint32_t k = J.rpC[row[sl]]; k < J.rpC[row[sl] + 1]; ++k)
                            if (ld_agent(&J.wmarkB[J.colC[k]]) < rank) blocked = true;
                } else {
                    // wmark holds the LOWEST rank that may write that state. Lower than mine: I would read
                    // (or overwrite) what an earlier row writes -> I am blocked. Higher than mine: that row
                    // would overwrite what I read -> it (and everything after it) is cut off.
                    for_row_sets(J, row[sl], shape[sl], xv[sl], [&](uint32_t v, uint32_t rd, uint32_t wr) {
                        if ((rd | wr) & 1) {
                            const uint32_t m = ld_agent(&J.wmarkU[v]);
                            if (m < rank) blocked = true; else if (m > rank && m != 0xFFFFFFFFu) atomicMin(&S.cut, m);
                        }
                        if ((rd | wr) & 2) {
                            const uint32_t m = ld_agent(&J.wmarkB[v]);
                            if (m < rank) blocked = true; else if (m > rank && m != 0xFFFFFFFFu) atomicMin(&S.cut, m);
                        }
                    });
                }
                if (blocked) atomicMin(&S.cut, rank);
            }
            if (S.bl_any) big_rows_check(J, S);
            __syncthreads();
            c = S.cut;   // >= 1: rank 0 is never blocked and not big
            if (S.bl_any) big_rows_unmark(J, S);
            // ---- unmark; tag the rows being popped with their rank (in_queue bookkeeping, see resolve_pushes)
#pragma unroll
            for (uint32_t sl = 0; sl < ECNE_RPL; ++sl) {
                if (sl >= rpl || r0 + sl >= n) continue;
                if ((live & (1u << sl)) && !(shape[sl] & SH_BIG) && !(noop & (1u << sl)))
                    for_row_sets(J, row[sl], shape[sl], xv[sl], [&](uint32_t v, uint32_t rd, uint32_t wr) {
                        if (wr & 1) J.wmarkU[v] = 0xFFFFFFFFu;
                        if (wr & 2) J.wmarkB[v] = 0xFFFFFFFFu;
                    });
                if (r0 + sl < c) J.inq[row[sl]] = (uint16_t)(r0 + sl + 2);
            }
            __syncthreads();
            QTICK(2);
        }
        // ---- execute the independent prefix, one lane per row (rpl rows per lane, in rank order)
        uint32_t nev[ECNE_RPL], nev_tot = 0;
#pragma unroll
        for (uint32_t sl = 0; sl < ECNE_RPL; ++sl) {
            nev[sl] = 0;
            if (sl >= rpl || r0 + sl >= c) continue;
            my_pops++;
            my_nnz += (J.rpA[row[sl] + 1] - J.rpA[row[sl]]) + (J.rpB[row[sl] + 1] - J.rpB[row[sl]]) + (J.rpC[row[sl] + 1] - J.rpC[row[sl]]);
            if (live & (1u << sl)) {
                if (noop & (1u << sl)) { if ((shape[sl] & SH_R4_T) && (shape[sl] & SH_R4_T2)) J.flip3[row[sl]] ^= 1; }   // the pop's only effect
                else if (!(shape[sl] & SH_BIG)) exec_row_lane(J, row[sl], J.evbuf + (size_t)(r0 + sl) * ECNE_EVCAP, nev[sl], C);
            }
            nev_tot += nev[sl];
        }
        uint32_t bigsl = 0;          // slots of mine that hold a long row executed in this round
        if (S.bl_any) {                // (uniform) the long rows of the prefix, by the whole workgroup
            big_rows_exec(J, S, c, 0);
#pragma unroll
            for (uint32_t sl = 0; sl < ECNE_RPL; ++sl)
                if (sl < rpl && r0 + sl < c && (shape[sl] & SH_BIG) && (live & (1u << sl))) {
                    const int k = big_slot_of(S, r0 + sl);
                    if (k >= 0) { nev[sl] = S.bl_nev[k]; nev_tot += nev[sl]; bigsl |= 1u << sl; }
                }
        }
        QTICK(3);
        // ---- REQUEUE resolution in sequential order: flatten the per-rank event lists, then resolve
        uint32_t Nev;
        {
            const uint32_t eoff = wg_exclusive_scan(nev_tot, S.scan, &Nev);
            uint32_t o = eoff;
#pragma unroll
            for (uint32_t sl = 0; sl < ECNE_RPL; ++sl) {
                if (sl >= rpl) continue;
                if (bigsl & (1u << sl)) {   // a long row's events are copied by the whole workgroup, below
                    S.bl_base[big_slot_of(S, r0 + sl)] = o;
                    o += nev[sl];
                    continue;
                }
                const uint32_t* ev = J.evbuf + (size_t)(r0 + sl) * ECNE_EVCAP;
                for (uint32_t e = 0; e < nev[sl]; ++e) { J.fvar[o] = ev[e]; J.frank[o] = r0 + sl; ++o; }
            }
            if (S.bl_any) {
                __syncthreads();
                for (uint32_t k = 0; k < ECNE_BIGK; ++k) {
                    if (S.bl_rank[k] >= c) continue;
                    const uint32_t* ev = big_ev(J, 0, k);
                    for (uint32_t e = tid; e < S.bl_nev[k]; e += ECNE_WG) { J.fvar[S.bl_base[k] + e] = ev[e]; J.frank[S.bl_base[k] + e] = S.bl_rank[k]; }
                }
            }
            __syncthreads();   // the flat list is read across lanes
        }
        QTICK(4);
        const uint32_t new_tail = resolve_pushes(J, S, J.fvar, true, Nev, (long long)q.head, c, q.tail, &hits[15]);
        QTICK(5);
        // rows of the prefix that nobody re-queued are out of the queue now
#pragma unroll
        for (uint32_t sl = 0; sl < ECNE_RPL; ++sl)
            if (sl < rpl && r0 + sl < c && J.inq[row[sl]] >= 2) J.inq[row[sl]] = 0;
        if (tid == 0) big_reset(S);
        __syncthreads();
        q.head += c;
        q.tail = new_tail;
        pops_total += c;
        hits[13]++;
        // adaptive: a short queue with a short independent prefix is a dependency chain -> sequential
        // burst, doubling while it stays that way
        if (c < 8 && avail < 64) { burst = next_burst; if (next_burst < 512) next_burst *= 2; }
        if (c == n) window = (window * ECNE_WGROW < ECNE_RPL * ECNE_WG) ? window * ECNE_WGROW : ECNE_RPL * ECNE_WG;
        else if (c < n / 4) { uint32_t wn = 4 * c; window = wn < ECNE_WMIN ? ECNE_WMIN : wn; }
        else next_burst = 16;
    }
    // ---- tell the helper workgroups (waiting at the command barrier) that the queue phase is over
    if (J.nwg > 1 && !helpers_released) {
        if (tid == 0) __hip_atomic_store(&J.ctr->q_cmd[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        job_barrier(J, s_err);
    }
    // ---- reduce the per-lane counters
    __syncthreads();
    if (C.steps) atomicAdd(&S.acc[0], (unsigned long long)C.steps);
    if (C.nuniq) atomicAdd(&S.acc[1], (unsigned long long)C.nuniq);
    for (int i = 0; i < 8; ++i)
        if (C.hits[i]) atomicAdd(&S.acc[2 + i], (unsigned long long)C.hits[i]);
    if (my_pops) atomicAdd(&S.acc[10], (unsigned long long)my_pops);
    if (my_nnz) atomicAdd(&S.acc[11], (unsigned long long)my_nnz);
    __syncthreads();
    steps += S.acc[0];
    nuniq += S.acc[1];
    for (int i = 0; i < 8; ++i) hits[i] += S.acc[2 + i];
    pops += S.acc[10];
    pop_nnz += S.acc[11];
    __syncthreads();
}

// Queue phase as seen by a helper workgroup: wait for the master's commands, join multi rounds.
__device__ __noinline__ void queue_phase_helper(const Job& J, ChunkShared& S, uint32_t wgrank, int* s_err) {
    LaneCtr C;
    C.steps = C.nuniq = 0;
    for (int i = 0; i < 8; ++i) C.hits[i] = 0;
    uint32_t my_pops = 0, my_nnz = 0;
    if (threadIdx.x < 12) S.acc[threadIdx.x] = 0;   // long rows executed by this workgroup count here
    if (threadIdx.x == 0) big_reset(S);
    __syncthreads();
    for (;;) {
        if (job_barrier(J, s_err)) break;
        if (ld_agent(&J.ctr->q_cmd[0]) == 0) break;
        const uint32_t head = ld_agent(&J.ctr->q_cmd[1]), tail = ld_agent(&J.ctr->q_cmd[2]), n = ld_agent(&J.ctr->q_cmd[3]);
        uint32_t c, nt;
        if (queue_round_multi(J, S, wgrank, head, tail, n, C, my_pops, my_nnz, s_err, &c, &nt)) break;
    }
    Counters* ctr = J.ctr;
    if (C.steps) atomicAdd(&ctr->q_acc[0], (unsigned long long)C.steps);
    if (C.nuniq) atomicAdd(&ctr->q_acc[1], (unsigned long long)C.nuniq);
    for (int i = 0; i < 8; ++i)
        if (C.hits[i]) atomicAdd(&ctr->q_acc[2 + i], (unsigned long long)C.hits[i]);
    if (my_pops) atomicAdd(&ctr->q_acc[10], (unsigned long long)my_pops);
    if (my_nnz) atomicAdd(&ctr->q_acc[11], (unsigned long long)my_nnz);
    __syncthreads();
    if (threadIdx.x < 10 && S.acc[threadIdx.x]) atomicAdd(&ctr->q_acc[threadIdx.x], S.acc[threadIdx.x]);
}

// ---------------------------------------------------------------------------------------- k_solve
struct WgDesc { uint32_t job, rank; };

__global__ __launch_bounds__(ECNE_WG) void k_solve(const Job* jobs, const WgDesc* wgs) {
    __shared__ Job J;
    __shared__ uint32_t s_scan[ECNE_NWAVES + 2];
    __shared__ uint32_t s_u32[8];
    __shared__ uint32_t s_htn;          // P3 group-table slots this workgroup created in the current sweep
    __shared__ unsigned long long s_steps;
    __shared__ uint32_t m_rows[10], m_vars[10];
    __shared__ int s_err;
    __shared__ QState s_q;
    __shared__ ChunkShared s_chunk;
    const int tid = threadIdx.x, lane = lane_id(), w = wave_id();
    const WgDesc me = wgs[blockIdx.x];
    if (tid < (int)(sizeof(Job) / 4)) ((uint32_t*)&J)[tid] = ((const uint32_t*)&jobs[me.job])[tid];
    __syncthreads();
    const uint32_t nC = J.nC, nV = J.nV;
    const bool master = me.rank == 0;
    if (tid < 8) { s_chunk.qt[tid] = 0; s_chunk.mt[tid] = 0; }
    const uint32_t gtid = me.rank * ECNE_WG + tid, gstride = J.nwg * ECNE_WG;   // job-wide thread index
    Counters* const ctr = J.ctr;
    const uint32_t ht_cap = (nC + J.nwg - 1) / J.nwg + 2048;   // this workgroup's share of ht_list (its rows + slack)
    if (tid == 0) { s_htn = 0; job_barrier_init(); }
    unsigned long long tk[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long t_last = wall_clock64();
#define ECNE_TICK(slot) do { unsigned long long t_now = wall_clock64(); tk[slot] += t_now - t_last; t_last = t_now; } while (0)

    // ---------------- setup (:593-704), all workgroups
    for (uint32_t i = tid; i < ECNE_BIGTAB; i += ECNE_WG) s_chunk.bt[i] = 0xFFFFFFFFu;
    for (uint32_t v = gtid; v <= nV; v += gstride) {
        J.flags[v] = 0;
        J.abz[v] = -1;
        J.nvalues[v] = 0;
        st256(J.lb + 4ull * v, fp::make(0));
        st256(J.ub + 4ull * v, fp::pminus1());
        J.varmin[v] = 0xFFFFFFFFu;
        J.wmarkU[v] = 0xFFFFFFFFu;
        J.wmarkB[v] = 0xFFFFFFFFu;
    }
    for (uint32_t r = gtid; r < nC; r += gstride) { J.inq[r] = 0; J.solved[r] = 0; J.flip3[r] = 0; J.best[r] = 0xFFFFFFFFu; J.rdead[r] = 0; J.p3k[r] = 0; }
    for (uint32_t r = gtid; r < nC + J.nSp; r += gstride) J.fired[r] = 0;   // [nC..) = special_solved
    for (uint32_t s = gtid; s <= J.htmask; s += gstride) { J.ht_key[s] = 0; J.ht_key2[s] = 0; J.ht_new[s] = 0; J.ht_frozen[s] = 0; }
    if (master && tid == 0) { ctr->p3_cand1 = 0xFFFFFFFFu; ctr->p3_nhot = 0; ctr->p3_any = 0; ctr->p3_hot = 0; ctr->p3_fire = 0xFFFFFFFFu; ctr->q_cut = 0xFFFFFFFFu; }
    job_barrier(J, &s_err);
    for (uint32_t i = gtid; i < J.nKnown; i += gstride) {
        uint32_t v = J.knowns[i];
        J.flags[v] = 3;
        if (v == 1) { J.nvalues[1] = 1; st256(J.values + 8ull, fp::make(1)); }
    }
    job_barrier(J, &s_err);
    // initial queue: rows with at most one variable outside known_variables, ascending (:621-627).
    // Every workgroup owns a contiguous block of rows: count, job-wide scan of the block totals, write.
    QState q;
    q.head = 0; q.tail = 0; q.evout = nullptr; q.nev = 0; q.emit = 0;
    {
        const uint32_t per = (nC + J.nwg - 1) / J.nwg;
        const uint32_t blk0 = me.rank * per < nC ? me.rank * per : nC;
        const uint32_t blk1 = (me.rank + 1) * per < nC ? (me.rank + 1) * per : nC;
        auto wants = [&](uint32_t r) -> uint32_t {
            uint32_t first = 0, cnt = 0;
            const uint32_t* rp[3] = {J.rpA, J.rpB, J.rpC};
            const uint32_t* cl[3] = {J.colA, J.colB, J.colC};
            for (int p = 0; p < 3 && cnt < 2; ++p)
                for (uint32_t e = rp[p][r]; e < rp[p][r + 1]; ++e) {
                    uint32_t v = cl[p][e];
                    if (!(J.flags[v] & 1)) {
                        if (cnt == 0) { first = v; cnt = 1; }
                        else if (v != first) { cnt = 2; break; }
                    }
                }
            return cnt <= 1;
        };
        uint32_t mine = 0;
        for (uint32_t r = blk0 + tid; r < blk1; r += ECNE_WG) mine += wants(r);
        uint32_t total_pushes = 0;
        int scan_err = 0;
        uint32_t base = team_exclusive_scan_any(J, s_chunk, me.rank, mine, &total_pushes, &s_err, &scan_err);
        // base = pushes of all lower workgroups + of lower threads of mine; but rows are interleaved
        // across my threads, so redo my block in row order with workgroup scans from my block's base
        uint32_t wg_base = base;
        {   // subtract my own lower threads' share: block base = value at thread 0
            if (tid == 0) s_u32[0] = base;
            __syncthreads();
            wg_base = s_u32[0];
            __syncthreads();
        }
        uint32_t off_run = wg_base;
        for (uint32_t b = blk0; b < blk1; b += ECNE_WG) {
            const uint32_t r = b + tid;
            const uint32_t push = (r < blk1) ? wants(r) : 0u;
            uint32_t tot;
            const uint32_t off = wg_exclusive_scan(push, s_scan, &tot);
            if (push) { J.queue[(off_run + off) & J.qmask] = r; J.inq[r] = 1; }
            off_run += tot;
        }
        q.tail = total_pushes;
        (void)scan_err;
    }
    ECNE_TICK(0);
    unsigned long long steps = 0, prev_steps = ~0ull, nuniq = 0, pops = 0, outer = 0, pop_nnz = 0;
    unsigned long long hits[16];
    for (int i = 0; i < 16; ++i) hits[i] = 0;
    // `steps` is the loop-control value: the master publishes it in ctr->sync_steps before each barrier

    for (;;) {
        if (master && tid == 0) ctr->sync_steps = steps;
        if (job_barrier(J, &s_err)) break;
        steps = __hip_atomic_load(&ctr->sync_steps, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (prev_steps == steps) break;   // (:708-711)
        prev_steps = steps;
        outer++;
        // ================= P1, P2 and the queue: master only, in the reference's order
        if (master) {
            if (w == 0) {
                // P1 (:718-747). 64 specials are tested at a time, one per lane; the ones whose inputs are all
                // unique fire in index order, and after every firing the later lanes look again (its outputs
                // may complete their inputs), which is what the one-by-one sweep would have seen.
                for (uint32_t base = 0; base < J.nSp; base += 64) {
                    const uint32_t i = base + lane;
                    int from = 0;
                    for (;;) {
                        bool can = i < J.nSp && lane >= from && !J.fired[nC + i];   // [nC..) = special_solved
                        if (can)
                            for (uint32_t e = J.sp_in_ptr[i]; e < J.sp_in_ptr[i + 1] && can; ++e) can = (J.flags[J.sp_in[e]] & 1) != 0;
                        const uint64_t m = __ballot(can);
                        if (!m) break;
                        const int src = __ffsll((long long)m) - 1;
                        const uint32_t is = base + (uint32_t)src;
                        if (lane == 0) J.fired[nC + is] = 1;
                        steps++; hits[8]++;
                        for (uint32_t e = J.sp_out_ptr[is]; e < J.sp_out_ptr[is + 1]; ++e) {
                            uint32_t v = J.sp_out[e];
                            if (J.flags[v] & 1) continue;
                            mark_unique(J, v);
                            requeue(J, q, v);
                        }
                        from = src + 1;
                    }
                }
                // P2 (:750-800): every (BigMultModP i, BigLessThan j) pair, from the two index lists
                for (uint32_t a = 0; a < J.nK1; ++a) {
                    const uint32_t i = J.k1_list[a];
                    for (uint32_t bj = 0; bj < J.nK2; ++bj) {
                        const uint32_t j = J.k2_list[bj];
                        if (!J.secp_solve) { raise(J, K_EUNDEF_DSU); break; }                 // `dsu` undefined (:762)
                        uint32_t ni = J.sp_in_ptr[i + 1] - J.sp_in_ptr[i], nj = J.sp_in_ptr[j + 1] - J.sp_in_ptr[j];
                        if (ni < 9 || nj < 6) { raise(J, K_EBOUNDS); break; }                // [k+3], [k] for k = 1..6
                        hits[9]++;
                        for (uint32_t t = 0; t < 3; ++t) {                                   // constraint_j[2][1:3]
                            uint32_t v = J.sp_in[J.sp_in_ptr[j] + t];
                            if (J.flags[v] & 1) continue;
                            mark_unique(J, v);
                            requeue(J, q, v);
                        }
                    }
                    if (J.ctr->error) break;
                }
                if (J.queue_mode == 1) {
                    // QUEUE (:805-1349), strictly sequential pops (debug / parity reference schedule)
                    const unsigned long long pop_cap = 4096ull + 64ull * (J.rpA[nC] + J.rpB[nC] + J.rpC[nC]);
                    while (q.head != q.tail && !J.ctr->error) {
                        if (pops > pop_cap) { raise(J, K_ECAPACITY); break; }
                        uint32_t row = J.queue[q.head & J.qmask];
                        q.head++;
                        if (lane == 0) J.inq[row] = 0;
                        wg_fence();
                        pops++;
                        pop_nnz += (J.rpA[row + 1] - J.rpA[row]) + (J.rpB[row + 1] - J.rpB[row]) + (J.rpC[row + 1] - J.rpC[row]);
                        if (J.solved[row]) continue;
                        exec_row(J, q, row, hits, steps, nuniq);
                    }
                }
                if (lane == 0) { s_q = q; s_steps = steps; }
            }
            __syncthreads();
            steps = s_steps;
            q = s_q;
            if (J.queue_mode != 1 && wg_error(J, &s_err)) {
                // P1/P2 raised: the queue phase is skipped, but the helpers are waiting at its command
                // barrier — meet them there (they leave on the error snapshot)
                if (J.nwg > 1) job_barrier(J, &s_err);
            } else if (J.queue_mode != 1) {
                // QUEUE (:805-1349), chunk-parallel schedule; counters other than `steps` live in wave 0
                unsigned long long st2 = steps, nu2 = 0, pp2 = 0, pn2 = 0, ht2[16];
                for (int i = 0; i < 16; ++i) ht2[i] = 0;
                queue_phase_chunked(J, q, s_chunk, ht2, st2, nu2, pp2, pn2, &s_err);
                steps = st2;
                if (w == 0) { nuniq += nu2; pops += pp2; pop_nnz += pn2; for (int i = 0; i < 8; ++i) hits[i] += ht2[i]; for (int i = 13; i < 16; ++i) hits[i] += ht2[i]; }
            }
        }
        else if (J.queue_mode != 1) queue_phase_helper(J, s_chunk, me.rank, &s_err);
        if (job_barrier(J, &s_err)) break;      // publishes the queue phase's state changes to the helpers
        if (master && J.nwg > 1 && J.queue_mode != 1) {
            // fold in what the helpers did during multi-workgroup rounds
            steps += ctr->q_acc[0];
            if (w == 0) {
                nuniq += ctr->q_acc[1]; pops += ctr->q_acc[10]; pop_nnz += ctr->q_acc[11];
                for (int i = 0; i < 8; ++i) hits[i] += ctr->q_acc[2 + i];
            }
            __syncthreads();
            if (tid < 16) ctr->q_acc[tid] = 0;
        }
        ECNE_TICK(1);

        // ================= P3 linear systems (:1357-1417): evaluation passes on all workgroups
        {
            uint32_t f = 0;   // rows < f are frozen (already swept in this pass)
            bool p3_err = false;
            for (;;) {
                tk[6]++;
                // phase 1: evaluate rows >= f against the current state
                // (the dead-row bytes are read four rows at a time: most of a large system is dead or idle)
                for (uint32_t r4 = (f & ~3u) + 4u * gtid; r4 < nC; r4 += 4u * gstride) {
                    const uint32_t dead4 = *reinterpret_cast<const uint32_t*>(J.rdead + r4);   // padded to a multiple of 4
                    if (dead4 == 0x01010101u) continue;
                    for (uint32_t r = r4 < f ? f : r4; r < r4 + 4 && r < nC; ++r) {
                        if ((dead4 >> (8 * (r - r4))) & 1) continue;   // every variable unique already (p3k[r] stays 0)
                        uint32_t k; uint64_t h, h2;
                        p3_eval(J, r, k, h, h2);
                        if (k == 0) J.rdead[r] = 1;        // eligible with no unknown left: nothing can change for this row
                        if (k == 0xFFFFFFFFu) k = 0;
                        J.p3k[r] = (uint8_t)(k > 255 ? 255 : k);
                        if (k == 1) atomicMin(&ctr->p3_cand1, r);
                        else if (k >= 2) {
                            J.p3h[r] = h; J.p3h2[r] = h2;   // only read for k >= 2
                            bool created = false;
                            uint32_t s = ht_slot(J, h, h2, true, &created);
                            if (s != 0xFFFFFFFFu) {
                                // p3_hot is raised only when this group could be complete with this member (k rows
                                // counting the frozen ones): otherwise nobody has to look for trigger rows this pass
                                const uint32_t before = atomicAdd(&J.ht_new[s], 1u);
                                if (before + 1 + ld_agent(&J.ht_frozen[s]) >= k)
                                    __hip_atomic_store(&ctr->p3_hot, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            }
                            __hip_atomic_store(&ctr->p3_any, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if (created) {   // remembered, so that only the slots in use are wiped afterwards
                                const uint32_t pos = atomicAdd(&s_htn, 1u);
                                if (pos < ht_cap) J.ht_list[(size_t)me.rank * ht_cap + pos] = s; else raise(J, K_ECAPACITY);
                            }
                        }
                    }
                }
                if (job_barrier(J, &s_err)) { p3_err = true; break; }
                const bool any = ld_agent(&ctr->p3_any) != 0;
                const bool hot = ld_agent(&ctr->p3_hot) != 0;
                // phase 2: rows whose group could reach its size in this pass
                if (hot) {
                    for (uint32_t r = f + gtid; r < nC; r += gstride) {
                        uint32_t k = J.p3k[r];
                        if (k < 2) continue;
                        uint32_t s = ht_slot(J, J.p3h[r], J.p3h2[r], false);
                        if (s == 0xFFFFFFFFu) continue;
                        uint32_t fr = ld_agent(&J.ht_frozen[s]);
                        if (fr < k && fr + ld_agent(&J.ht_new[s]) >= k) {
                            uint32_t pos = atomicAdd(&ctr->p3_nhot, 1u);
                            if (pos < J.hotcap) J.hot[pos] = r;
                        }
                    }
                    if (job_barrier(J, &s_err)) { p3_err = true; break; }
                }
                // phase 3 (master, wave 0): find the earliest trigger row that passes the test
                if (master) {
                    if (w == 0) {
                        uint32_t nhot = hot ? ld_agent(&ctr->p3_nhot) : 0;
                        if (nhot > J.hotcap) { raise(J, K_ECAPACITY); nhot = 0; }
                        uint32_t best = ld_agent(&ctr->p3_cand1);   // k == 1: first arrival of a one-variable group always fires
                        for (uint32_t a = 0; a < nhot; ++a) {
                            uint32_t t = J.hot[a];
                            if (t >= best) continue;
                            uint32_t k = J.p3k[t];
                            uint64_t h = J.p3h[t], h2 = J.p3h2[t];
                            uint32_t s = ht_slot(J, h, h2, false);
                            uint32_t fr = (s == 0xFFFFFFFFu) ? 0 : ld_agent(&J.ht_frozen[s]);
                            // arrival number of t = frozen + fresh members with index <= t
                            uint32_t part = 0;
                            for (uint32_t b = lane; b < nhot; b += 64) {
                                uint32_t o = J.hot[b];
                                if (o <= t && J.p3h[o] == h && J.p3h2[o] == h2 && J.p3k[o] == k) part++;
                            }
                            for (int d = 32; d >= 1; d >>= 1) part += __shfl_xor(part, d, 64);
                            if (fr + part != k) continue;
                            if (k > 10) { raise(J, K_EDETSIZE); break; }
                            // collect the k member rows in arrival (index) order and the k variables ascending
                            if (lane == 0) {
                                uint32_t n = 0;
                                if (fr) {   // frozen members: rows < f with the same key at their time
                                    for (uint32_t r = 0; r < f && n < k; ++r)
                                        if (J.p3k[r] == k && J.p3h[r] == h && J.p3h2[r] == h2) m_rows[n++] = r;
                                }
                                uint32_t last = 0; bool have = false;
                                while (n < k) {
                                    uint32_t mn = 0xFFFFFFFFu;
                                    for (uint32_t b = 0; b < nhot; ++b) {
                                        uint32_t o = J.hot[b];
                                        if (J.p3h[o] == h && J.p3h2[o] == h2 && J.p3k[o] == k && (!have || o > last) && o < mn) mn = o;
                                    }
                                    if (mn == 0xFFFFFFFFu) break;
                                    m_rows[n++] = mn; last = mn; have = true;
                                }
                                uint32_t nv = 0;
                                for (uint32_t e = J.rpC[t]; e < J.rpC[t + 1]; ++e) {
                                    uint32_t v = J.colC[e];
                                    if (!(J.flags[v] & 1)) {
                                        uint32_t pos = nv++;
                                        while (pos > 0 && m_vars[pos - 1] > v) { m_vars[pos] = m_vars[pos - 1]; --pos; }
                                        m_vars[pos] = v;
                                    }
                                }
                            }
                            wg_fence();
                            if (p3_odd_perm_sum_nonzero(J, m_rows, m_vars, k)) best = t;
                        }
                        if (lane == 0) {
                            ctr->p3_fire = best;
                            ctr->p3_cand1 = 0xFFFFFFFFu; ctr->p3_nhot = 0; ctr->p3_any = 0; ctr->p3_hot = 0;   // ready for the next round
                        }
                    }
                }
                if (job_barrier(J, &s_err)) { p3_err = true; break; }
                const uint32_t fire = ld_agent(&ctr->p3_fire);
                if (fire == 0xFFFFFFFFu) break;
                const uint32_t upto = fire + 1;
                // phase 4: freeze rows [f, upto): their arrivals are now history; forget fresh counts
                if (any) {
                    for (uint32_t r = f + gtid; r < nC; r += gstride) {
                        uint32_t k = J.p3k[r];
                        if (k < 2) continue;
                        uint32_t s = ht_slot(J, J.p3h[r], J.p3h2[r], false);
                        if (s == 0xFFFFFFFFu) continue;
                        if (r < upto) atomicAdd(&J.ht_frozen[s], 1u);
                        J.ht_new[s] = 0;
                    }
                }
                // apply the firing (master): the group's variables, ascending, become unique (:1403-1414)
                if (master) {
                    if (w == 0) {
                        uint32_t k = J.p3k[fire];
                        steps += k; hits[10]++;
                        uint32_t lastv = 0;
                        for (uint32_t n = 0; n < k; ++n) {
                            uint32_t mn = 0xFFFFFFFFu;
                            for (uint32_t e = J.rpC[fire] + lane; e < J.rpC[fire + 1]; e += 64) {
                                uint32_t v = J.colC[e];
                                if (!(J.flags[v] & 1) && v > lastv && v < mn) mn = v;
                            }
                            for (int d = 32; d >= 1; d >>= 1) { uint32_t o = __shfl_xor(mn, d, 64); mn = o < mn ? o : mn; }
                            if (mn == 0xFFFFFFFFu) break;
                            lastv = mn;
                            J.events[n] = mn;
                        }
                        wg_fence();
                        for (uint32_t n = 0; n < k; ++n) {
                            uint32_t v = J.events[n];
                            mark_unique(J, v);
                            requeue(J, q, v);
                        }
                        if (lane == 0) s_steps = steps;
                    }
                    __syncthreads();
                    steps = s_steps;
                }
                if (job_barrier(J, &s_err)) { p3_err = true; break; }   // the firing's writes reach the helpers
                f = fire + 1;
            }
            if (p3_err) break;
            // leave the table clean for the next outer iteration: every workgroup wipes the slots it created
            __syncthreads();
            {
                const uint32_t nmine = s_htn < ht_cap ? s_htn : ht_cap;
                for (uint32_t i = tid; i < nmine; i += ECNE_WG) {
                    const uint32_t s = J.ht_list[(size_t)me.rank * ht_cap + i];
                    J.ht_key[s] = 0; J.ht_key2[s] = 0; J.ht_new[s] = 0; J.ht_frozen[s] = 0;
                }
                __syncthreads();
                if (tid == 0) s_htn = 0;
            }
        }
        ECNE_TICK(2);

        // ================= P4 ABZ tagging (:1425-1483): marking and tagging on all workgroups.
        // p4_b[i] / p4_s[i]: B variable and slope variable (bit 31: no slope -> DivideError) of the i-th
        // statically eligible row. A row tags b iff b is not unique, still untagged, and the row is the
        // FIRST such row of b in index order (varmin[b]).
        {
            for (uint32_t i = gtid; i < J.nP4; i += gstride) {
                const uint32_t b = J.p4_b[i];
                if (J.flags[b] & 1) continue;
                if (J.p4_s[i] & 0x80000000u) { raise(J, K_EDIVZERO); continue; }
                if (ld_agent(&J.varmin[b]) > i) atomicMin(&J.varmin[b], i);
            }
            if (job_barrier(J, &s_err)) break;
            for (uint32_t i = gtid; i < J.nP4; i += gstride) {
                const uint32_t b = J.p4_b[i];
                if ((J.flags[b] & 1) || ld_agent(&J.varmin[b]) != i || J.abz[b] != -1) continue;
                J.abz[b] = (int32_t)(J.p4_s[i] & 0x7FFFFFFFu);
                J.flags[b] |= 2;
                J.fired[i] = 1;          // by list position; cleared again when the events are collected
                atomicAdd(&ctr->p4_nfired, 1u);
            }
            if (master && tid == 0) ctr->q_cmd[2] = q.tail;   // (thread 0 holds the queue cursor) for p4's job-wide REQUEUE
            if (job_barrier(J, &s_err)) break;
            const uint32_t p4_fired = ld_agent(&ctr->p4_nfired);   // stable until the master clears it at the end of P4
            // forget the per-variable minima (all workgroups; the next use is a whole queue phase away)
            for (uint32_t i = gtid; i < J.nP4; i += gstride) {
                const uint32_t b = J.p4_b[i];
                if (!(J.flags[b] & 1)) J.varmin[b] = 0xFFFFFFFFu;
            }
            bool p4_done = false, p4_err = false;
            if (J.nwg > 1 && p4_fired >= 2048) {
                // Many rows tagged (the first sweep of a large circuit tags every decoder output): the ordered
                // REQUEUE of their B variables runs on ALL workgroups -- same steps as resolve_pushes, with
                // contiguous blocks per thread and job-wide scans (nothing is being popped: a candidate may
                // push iff its row is not queued; the lowest candidate index per row wins).
                const uint32_t tail0 = ld_agent(&ctr->q_cmd[2]);
                const uint32_t T = gstride;
                int err = 0;
                // 1. the event list: B variables of the fired rows, ascending
                const uint32_t iper = (J.nP4 + T - 1) / T;
                const uint32_t i0 = gtid * iper < J.nP4 ? gtid * iper : J.nP4, i1 = (gtid + 1) * iper < J.nP4 ? (gtid + 1) * iper : J.nP4;
                uint32_t cnt = 0;
                for (uint32_t i = i0; i < i1; ++i) cnt += J.fired[i];
                uint32_t nev = 0;
                uint32_t o = team_exclusive_scan(J, s_chunk, me.rank, cnt, 0, &nev, &s_err, &err);
                if (!err) {
                    for (uint32_t i = i0; i < i1; ++i)
                        if (J.fired[i]) { J.events[o++] = J.p4_b[i]; J.fired[i] = 0; }
                    err = job_barrier(J, &s_err);
                }
                // 2. candidates
                uint32_t M = 0, e0 = 0, e1 = 0, cbase = 0;
                if (!err) {
                    const uint32_t eper = (nev + T - 1) / T;
                    e0 = gtid * eper < nev ? gtid * eper : nev;
                    e1 = (gtid + 1) * eper < nev ? (gtid + 1) * eper : nev;
                    uint32_t deg = 0;
                    for (uint32_t e = e0; e < e1; ++e) { const uint32_t v = J.events[e]; deg += J.fo_ptr[v + 1] - J.fo_ptr[v]; }
                    cbase = team_exclusive_scan(J, s_chunk, me.rank, deg, 1, &M, &s_err, &err);
                }
                if (!err && M <= J.candcap) {
                    if (tid == 0) s_chunk.nbigev = 0;
                    __syncthreads();
                    uint32_t j = cbase;
                    for (uint32_t e = e0; e < e1; ++e) {
                        const uint32_t v = J.events[e];
                        expand_event(J, s_chunk, v, 0, j, false);
                        j += J.fo_ptr[v + 1] - J.fo_ptr[v];
                    }
                    expand_big_events(J, s_chunk, false);
                    err = job_barrier(J, &s_err);
                    // 3. winners, in candidate order
                    uint32_t W = 0;
                    if (!err) {
                        const uint32_t cper = (M + T - 1) / T;
                        const uint32_t j0 = gtid * cper < M ? gtid * cper : M, j1 = (gtid + 1) * cper < M ? (gtid + 1) * cper : M;
                        uint32_t nwin = 0;
                        for (uint32_t jj = j0; jj < j1; ++jj) {
                            const uint32_t cw = J.cand[jj];
                            const uint32_t t = cw & 0x7FFFFFFFu;
                            const bool win = (cw & 0x80000000u) && ld_agent(&J.best[t]) == jj;
                            J.cand[jj] = t | (win ? 0x80000000u : 0u);
                            nwin += win;
                        }
                        const uint32_t wbase = team_exclusive_scan(J, s_chunk, me.rank, nwin, 0, &W, &s_err, &err);
                        if (!err) {
                            uint32_t oq = tail0 + wbase;
                            for (uint32_t jj = j0; jj < j1; ++jj) {
                                const uint32_t cw = J.cand[jj];
                                const uint32_t t = cw & 0x7FFFFFFFu;
                                if (cw & 0x80000000u) { J.queue[oq & J.qmask] = t; J.inq[t] = 1; ++oq; }
                                J.best[t] = 0xFFFFFFFFu;
                            }
                            err = job_barrier(J, &s_err);
                        }
                    }
                    if (!err) {
                        p4_done = true;
                        if (master) {
                            if (w == 0 && lane == 0) { q.tail = tail0 + W; s_q = q; ctr->p4_nfired = 0; }
                            __syncthreads();
                            q = s_q;
                            steps += nev;
                            if (w == 0) hits[11] += nev;
                        }
                    }
                } else if (!err) {
                    // (a B variable with a huge fan-out) the master replays the events one by one
                    if (master) {
                        if (w == 0) {
                            for (uint32_t e = 0; e < nev; ++e) requeue(J, q, J.events[e]);
                            if (lane == 0) { s_q = q; ctr->p4_nfired = 0; }
                        }
                        __syncthreads();
                        q = s_q;
                        steps += nev;
                        if (w == 0) { hits[11] += nev; hits[15]++; }
                    }
                    err = job_barrier(J, &s_err);
                    if (!err) p4_done = true;
                }
                if (err) p4_err = true;
            }
            if (p4_err) break;
            if (master && p4_fired != 0 && !p4_done) {
                __syncthreads();
                if (tid == 0) ctr->p4_nfired = 0;
                // wave 0 owns the queue cursor during P1-P3; every master thread needs it now
                if (w == 0 && lane == 0) s_q = q;
                __syncthreads();
                q = s_q;
                // ordered event list = fired rows ascending -> their b variable
                uint32_t nev = 0;
                for (uint32_t base = 0; base < J.nP4; base += ECNE_WG) {
                    uint32_t i = base + tid;
                    uint32_t fl = (i < J.nP4) ? J.fired[i] : 0;
                    uint32_t total, off = wg_exclusive_scan(fl, s_scan, &total);
                    if (fl) { J.events[nev + off] = J.p4_b[i]; J.fired[i] = 0; }
                    nev += total;
                }
                __syncthreads();
                // REQUEUE(b) for every fired row, in row order, resolved by the whole workgroup
                {
                    uint32_t tl = q.tail;
                    for (uint32_t eb = 0; eb < nev; eb += 4096) {
                        const uint32_t cnt = (nev - eb) < 4096u ? (nev - eb) : 4096u;
                        tl = resolve_pushes(J, s_chunk, J.events + eb, false, cnt, -1, 0, tl, &hits[15]);
                    }
                    q.tail = tl;
                    steps += nev;
                    if (w == 0) hits[11] += nev;
                }
            }
        }
        ECNE_TICK(3);

        // ================= P5 isZero pairs (:1492-1550), ascending over the static candidates (master)
        if (master) {
            if (w == 0) {
                // 64 candidates are tested at a time, one per lane; the ones that pass fire in index order,
                // and after every firing the later lanes look again (its newly unique y may complete their A)
                for (uint32_t base = 0; base < J.nP5; base += 64) {
                    const uint32_t i = base + lane;
                    const uint32_t r = i < J.nP5 ? J.p5_rows[i] : 0, y = i < J.nP5 ? J.p5_y[i] : 0;
                    int from = 0;            // lanes below `from` are done
                    for (;;) {
                        bool can = i < J.nP5 && lane >= from && !(J.flags[y] & 1);
                        if (can)
                            for (uint32_t e = J.rpA[r]; e < J.rpA[r + 1] && can; ++e) can = (J.flags[J.colA[e]] & 1) != 0;
                        const uint64_t m = __ballot(can);
                        if (!m) break;
                        const int src = __ffsll((long long)m) - 1;
                        const uint32_t rs = __shfl(r, src, 64), ys = __shfl(y, src, 64);
                        mark_unique(J, ys);
                        if (lane == 0) { J.solved[rs] = 1; J.solved[rs + 1] = 1; }
                        wg_fence();
                        steps++; hits[12]++;
                        requeue(J, q, ys);
                        from = src + 1;
                    }
                }
                if (lane == 0) s_steps = steps;
            }
            __syncthreads();
            steps = s_steps;
        }
        ECNE_TICK(4);
    }

    // ---------------- verdict counts (:1558-1597), all workgroups
    job_barrier(J, &s_err);
    uint32_t un = 0, nn = 0, ut = 0;
    for (uint32_t v = 1 + gtid; v <= nV; v += gstride) {
        if (J.nontrivial[v]) { nn++; if (J.flags[v] & 1) un++; }
    }
    for (uint32_t i = gtid; i < J.nTarget; i += gstride)
        if (J.flags[J.targets[i]] & 1) ut++;
    {
        uint32_t t0, t1, t2;
        wg_exclusive_scan(un, s_scan, &t0);
        wg_exclusive_scan(nn, s_scan, &t1);
        wg_exclusive_scan(ut, s_scan, &t2);
        if (tid == 0) {
            atomicAdd(&ctr->unique_nontrivial, (unsigned long long)t0);
            atomicAdd(&ctr->n_nontrivial, (unsigned long long)t1);
            atomicAdd(&ctr->unique_targets, (unsigned long long)t2);
            if (master) {
                ctr->successful_steps = steps;
                ctr->num_unique = nuniq;
                ctr->pops = pops;
                ctr->pop_nnz = pop_nnz;
                ctr->outer_iterations = outer;
                for (int i = 0; i < 16; ++i) ctr->rule_hits[i] = hits[i];
                ctr->q_head = q.head;
                ctr->q_tail = q.tail;
                ECNE_TICK(5);
                for (int i = 0; i < 8; ++i) ctr->phase_ticks[i] = tk[i];
                for (int i = 0; i < 8; ++i) ctr->qticks[i] = s_chunk.qt[i];
                for (int i = 0; i < 8; ++i) ctr->mticks[i] = s_chunk.mt[i];
            }
        }
    }
}

// ---------------------------------------------------------------------------------- fp self-test
__global__ void k_fp_selftest(int op, size_t n, const uint64_t* a, const uint64_t* b, uint64_t* out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    fp::u256 x = ld256(a + 4 * i), y = ld256(b + 4 * i), r = fp::make(0);
    switch (op) {
        case 0: r = fp::add(x, y); break;
        case 1: r = fp::sub(x, y); break;
        case 2: r = fp::mul(x, y); break;
        case 3: r = fp::is_zero(x) ? fp::make(0) : fp::inv(x); break;
        case 4: r = fp::neg(x); break;
        case 5: r = fp::is_zero(y) ? fp::make(0) : fp::mul(x, fp::inv(y)); break;
        case 6: { fp::u256 q, rem; if (!fp::is_zero(y)) { fp::divmod(x, y, q, rem); r = q; } } break;
        case 7: r = fp::make(fp::mul_gt_p(x, y) ? 1 : 0); break;
    }
    st256(out + 4 * i, r);
}

}  // namespace ecne
