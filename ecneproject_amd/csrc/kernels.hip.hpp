// kernels.hip.hpp — gfx950 device code of the Ecne propagation engine (included by ecne_engine.hip).
//
// Kernels
//   k_classify_rows   one launch, two kinds of workgroups (one lane per short row, one wavefront per longer row): streams the
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
//                     DESIGN.md "Schedule"). Files, bottom up: rules_wave / rules_lane (one pop by a wavefront / a lane),
//                     schedule (access sets, long rows, ordered REQUEUE), job_barrier, fastrow (the pop of the common row
//                     shapes decided in registers, the walk of a long row), chain (sequential pops of a single-workgroup
//                     job out of LDS), wave2 (the fast wavefront round), level / crew (the narrow dependency levels of a deep circuit: one lane / one
//                     wavefront per queued row), drain (a window of the queue in dataflow order on a team of workgroups), rounds (multi-workgroup round, the queue
//                     phase's policy), k_solve (setup, outer loop, P1-P5, verdict).
//   (k_abs_*, k_fe_*, k_lay_*: the device front-end -- parse, abstraction, layout -- lives in the second translation unit,
//                     ecne_frontend.hip / frontend.hip.hpp / abstract.hip.hpp)
//   k_fp_selftest     field-arithmetic known-answer vectors on the device.
#pragma once
#include "k_solve.hip.hpp"
#include "classify.hip.hpp"

namespace ecne {

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
