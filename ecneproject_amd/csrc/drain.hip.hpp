// drain.hip.hpp — the drain round: a window of the queue executed in DATAFLOW order by all workgroups of the job.
// Part of the gfx950 device code of libecne_hip (see kernels.hip.hpp for the overview).
//
// A round of queue_round_multi commits the longest prefix of pairwise independent rows and hands the rest back: the reference
// pops one row at a time (/root/reference/src/R1CSConstraintSolver.jl:805-1349), so a row that depends on an earlier one of the
// window ends the round -- and a circuit whose independent blocks sit one behind the other in the FIFO (52 multiplexers of an
// ECDSA circuit: decoder sum, its 4 096 dependents, next decoder sum, ...) is then worked off one block per three rounds although
// the blocks have nothing to do with each other (tests/tools/dataflow_depth.py: ecdsa_like(26) needs 165 dataflow levels, the
// prefix schedule took 692 rounds).
//
// The drain round executes the WHOLE window, level by level: at every level each pending row looks at the pending rows of LOWER
// rank only and runs as soon as none of them can still touch what it touches. That is the sequential result: a row then reads
// exactly the state all earlier pops leave (every earlier row that could write it has run, none that runs later writes it), no
// earlier pending row reads or writes what it writes, and the pops of a window all precede the pops of anything the window pushes
// (FIFO). The pushes are resolved ONCE, after the window has drained, in (rank, emission, fan-out) order with the reference's
// in_queue rule -- multi_finish, the same code as after a prefix round.
//
// Hazard test of one level (three planes of per-variable marks, U and B class each, lowest rank wins; epoch-keyed so that
// nothing is ever reset):
//   X   exact writes: what the row would write if it ran on the state as it is now (every pending row marks, P1)
//   A   accesses that matter: a row that finds a HIGHER X mark on something it accesses (a later row wants to write it) says so
//       here, and an unstable row marks everything it can access in any state it may still see -- so a writer finds every
//       earlier pending row that still has to read or write its target, and nobody pays an atomic for an access nobody contests
//   C   conservative writes of UNSTABLE rows -- rows that found a LOWER X mark on something they access: their inputs may still
//       change, so what they will write is not known; they mark everything they could
//   P1 mark X | P2 look at X: unstable? -> mark C and A; contested access -> mark A | P3 (only if somebody marked): a lower C mark
//   on what a row accesses demotes it (the lowest demoted rank becomes the level's cut: rows above it cannot tell which lower rows
//   are trustworthy and wait), a lower A mark on what it writes makes it wait a level | run the rest.
// A level in which nobody marked A or C is two job barriers, and the window has drained after it.
// The lowest pending row always runs, so the window drains; the window size follows the levels a drain needed (rounds.hip.hpp).
// Long rows (plain ones, as in every round) are taken by their workgroup as a whole; a long row of another shape, or one the
// workgroup has no slot for, ends the window BEFORE anything has run (level 1 only).
#pragma once

namespace ecne {

enum : int { DX_U = 0, DX_B = 1, DA_U = 2, DA_B = 3, DC_U = 4, DC_B = 5 };
#define ECNE_DRAIN_EPOCH_MAX 32766u

__device__ __forceinline__ uint32_t dr_key(uint32_t epoch, uint32_t rank) { return (epoch << 17) | (0x1FFFFu - rank); }
__device__ __forceinline__ void dr_mark(uint32_t* plane, uint32_t v, uint32_t key) {
    if (ld_agent(&plane[v]) < key) atomicMax(&plane[v], key);
}
// (P1: the exact write marks are hardly ever contested -- no look before the atomic, one round trip less)
__device__ __forceinline__ void dr_mark_now(uint32_t* plane, uint32_t v, uint32_t key) { atomicMax(&plane[v], key); }
// a mark of the current epoch with a LOWER rank than key's (older epochs are smaller than every key of this one)
__device__ __forceinline__ bool dr_lower(const uint32_t* plane, uint32_t v, uint32_t key) { return ld_agent(&plane[v]) > key; }
// 1: a lower rank marked, 2: a higher rank did (and no lower one), 0: nobody else
__device__ __forceinline__ uint32_t dr_see(const uint32_t* plane, uint32_t v, uint32_t key) {
    const uint32_t k = ld_agent(&plane[v]);
    return k > key ? 1u : (k < key && (k >> 17) == (key >> 17)) ? 2u : 0u;
}

// ---- a small row on the general path (no record decision): the access-set walk of schedule.hip.hpp, up to four times per level
__device__ __noinline__ void dr_gen_mark(const Job& J, uint32_t row, uint32_t shape, uint32_t x, uint32_t key) {
    for_row_sets4(J, row, shape, x, [&](uint32_t v, uint32_t rd, uint32_t wr, uint32_t wrc) {
        if (wr & 1) dr_mark_now(J.dmk[DX_U], v, key);
        if (wr & 2) dr_mark_now(J.dmk[DX_B], v, key);
    });
}
// bit 0: unstable (a lower row writes, as things stand, something this one accesses), bit 1: it marked a contested access in A
__device__ __noinline__ uint32_t dr_gen_p2(const Job& J, uint32_t row, uint32_t shape, uint32_t x, uint32_t key) {
    uint32_t r = 0;
    for_row_sets4(J, row, shape, x, [&](uint32_t v, uint32_t rd, uint32_t wr, uint32_t wrc) {
        const uint32_t acc = rd | wrc;
        if (acc & 1) { const uint32_t s = dr_see(J.dmk[DX_U], v, key); if (s == 1) r |= 1u; else if (s == 2) { dr_mark(J.dmk[DA_U], v, key); r |= 2u; } }
        if (acc & 2) { const uint32_t s = dr_see(J.dmk[DX_B], v, key); if (s == 1) r |= 1u; else if (s == 2) { dr_mark(J.dmk[DA_B], v, key); r |= 2u; } }
    });
    return r;
}
// an unstable row: everything it could write goes to C, everything it can access to A
__device__ __noinline__ void dr_gen_mark_unstable(const Job& J, uint32_t row, uint32_t shape, uint32_t x, uint32_t key) {
    for_row_sets4(J, row, shape, x, [&](uint32_t v, uint32_t rd, uint32_t wr, uint32_t wrc) {
        const uint32_t acc = rd | wrc;
        if (wrc & 1) dr_mark(J.dmk[DC_U], v, key);
        if (wrc & 2) dr_mark(J.dmk[DC_B], v, key);
        if (acc & 1) dr_mark(J.dmk[DA_U], v, key);
        if (acc & 2) dr_mark(J.dmk[DA_B], v, key);
    });
}
// bit 0: demoted (a lower UNSTABLE row could still write something this one accesses; looked at only when somebody is unstable),
// bit 1: has to wait a level (a lower pending row still accesses what this one writes)
__device__ __noinline__ uint32_t dr_gen_p3(const Job& J, uint32_t row, uint32_t shape, uint32_t x, uint32_t key, bool noop, bool look_c) {
    uint32_t r = 0;
    for_row_sets4(J, row, shape, x, [&](uint32_t v, uint32_t rd, uint32_t wr, uint32_t wrc) {
        const uint32_t acc = rd | wrc;
        if (look_c) {
            if ((acc & 1) && dr_lower(J.dmk[DC_U], v, key)) r |= 1u;
            if ((acc & 2) && dr_lower(J.dmk[DC_B], v, key)) r |= 1u;
        }
        if (!noop) {
            if ((wr & 1) && dr_lower(J.dmk[DA_U], v, key)) r |= 2u;
            if ((wr & 2) && dr_lower(J.dmk[DA_B], v, key)) r |= 2u;
        }
    });
    return r;
}

// ---- the registered long rows of this workgroup (S.bl_*, S.dr_st[k]: bit 0 pending, 1 unstable, 2 demoted, 3 waiting, 4 ran
// in this level). All threads of the workgroup, uniform control flow. Sets: reads U of every non-final variable and B of C's
// non-unique ones, may write U of C's non-final ones (exact = conservative: R1 / R7 / R8 decide on the whole row).
// (round 5) The walks over a long row's entries take FOUR strides per trip -- the ids of all four first, then their flag bytes, then the
// mark words, then the atomics -- so that a 1 025-term row is three or four dependent trips to memory per phase instead of three per
// stride (nine and more): with two or three such rows registered in one workgroup (the sums of a multiplexer sit next to each other
// in the queue) every level of the round waited for them.
#define ECNE_BW 4u
template <class F>
__device__ __forceinline__ void big_walk(const uint32_t* __restrict__ col, const uint8_t* flags, uint32_t e0, uint32_t e1, F f) {
    for (uint32_t base = e0 + threadIdx.x; base < e1; base += ECNE_BW * ECNE_WG) {
        uint32_t v[ECNE_BW];
        uint8_t fl[ECNE_BW];
#pragma unroll
        for (uint32_t j = 0; j < ECNE_BW; ++j) v[j] = base + j * ECNE_WG < e1 ? col[base + j * ECNE_WG] : 0u;
#pragma unroll
        for (uint32_t j = 0; j < ECNE_BW; ++j) fl[j] = base + j * ECNE_WG < e1 ? flags[v[j]] : (uint8_t)3;
        f(v, fl, base, e1);
    }
}
// marks of up to four variables on one plane: the words first, the atomics where they are needed (dr_mark)
__device__ __forceinline__ void dr_mark4(uint32_t* plane, const uint32_t* v, const bool* want, uint32_t key) {
    uint32_t m[ECNE_BW];
#pragma unroll
    for (uint32_t j = 0; j < ECNE_BW; ++j) m[j] = want[j] ? ld_agent(&plane[v[j]]) : 0xFFFFFFFFu;
#pragma unroll
    for (uint32_t j = 0; j < ECNE_BW; ++j) if (want[j] && m[j] < key) atomicMax(&plane[v[j]], key);
}
__device__ __noinline__ void dr_big_p1(const Job& J, ChunkShared& S, uint32_t epoch) {
    for (uint32_t k = 0; k < ECNE_BIGK; ++k) {
        if (!(S.dr_st[k] & 1u)) continue;
        const uint32_t row = S.bl_row[k], key = dr_key(epoch, S.bl_rank[k]);
        big_walk(J.colC, J.flags, J.rpC[row], J.rpC[row + 1], [&](const uint32_t* v, const uint8_t* fl, uint32_t base, uint32_t e1) {
            bool w[ECNE_BW];
#pragma unroll
            for (uint32_t j = 0; j < ECNE_BW; ++j) w[j] = base + j * ECNE_WG < e1 && (fl[j] & 3) != 3;
            dr_mark4(J.dmk[DX_U], v, w, key);
        });
    }
}
// what a reader finds on plane px for up to four variables (dr_see): 1 a lower rank marked (unstable), 2 a higher one (the access is contested:
// marked on plane pa); `u` / `am` collect the two outcomes
__device__ __forceinline__ void dr_look4(const Job& J, int px, int pa, const uint32_t* v, const bool* want, uint32_t key, bool& u, bool& am) {
    uint32_t x[ECNE_BW];
    bool contested[ECNE_BW];
#pragma unroll
    for (uint32_t j = 0; j < ECNE_BW; ++j) x[j] = want[j] ? ld_agent(&J.dmk[px][v[j]]) : 0u;
#pragma unroll
    for (uint32_t j = 0; j < ECNE_BW; ++j) {
        const uint32_t s = !want[j] ? 0u : x[j] > key ? 1u : (x[j] < key && (x[j] >> 17) == (key >> 17)) ? 2u : 0u;
        if (s == 1) u = true;
        contested[j] = s == 2;
        if (s == 2) am = true;
    }
    dr_mark4(J.dmk[pa], v, contested, key);
}
// returns (to every thread) bit 1 if some long row of this workgroup marked a contested access
__device__ __noinline__ uint32_t dr_big_p2(const Job& J, ChunkShared& S, uint32_t epoch) {
    bool am = false;
    for (uint32_t k = 0; k < ECNE_BIGK; ++k) {
        if (!(S.dr_st[k] & 1u)) continue;
        const uint32_t row = S.bl_row[k], key = dr_key(epoch, S.bl_rank[k]);
        bool u = false;
        auto ab = [&](const uint32_t* v, const uint8_t* fl, uint32_t base, uint32_t e1) {
            bool w[ECNE_BW];
#pragma unroll
            for (uint32_t j = 0; j < ECNE_BW; ++j) w[j] = base + j * ECNE_WG < e1 && (fl[j] & 3) != 3;
            dr_look4(J, DX_U, DA_U, v, w, key, u, am);
        };
        big_walk(J.colA, J.flags, J.rpA[row], J.rpA[row + 1], ab);
        big_walk(J.colB, J.flags, J.rpB[row], J.rpB[row + 1], ab);
        big_walk(J.colC, J.flags, J.rpC[row], J.rpC[row + 1], [&](const uint32_t* v, const uint8_t* fl, uint32_t base, uint32_t e1) {
            bool wu[ECNE_BW], wb[ECNE_BW];
#pragma unroll
            for (uint32_t j = 0; j < ECNE_BW; ++j) { const bool on = base + j * ECNE_WG < e1; wu[j] = on && (fl[j] & 3) != 3; wb[j] = on && !(fl[j] & 1); }
            dr_look4(J, DX_U, DA_U, v, wu, key, u, am);
            dr_look4(J, DX_B, DA_B, v, wb, key, u, am);
        });
        if (u) atomicOr(&S.dr_st[k], 2u);
    }
    const int any_am = __syncthreads_or(am ? 1 : 0);
    for (uint32_t k = 0; k < ECNE_BIGK; ++k) {
        if ((S.dr_st[k] & 3u) != 3u) continue;       // unstable: C (what it could write) and A (what it can access)
        const uint32_t row = S.bl_row[k], key = dr_key(epoch, S.bl_rank[k]);
        auto ab = [&](const uint32_t* v, const uint8_t* fl, uint32_t base, uint32_t e1) {
            bool w[ECNE_BW];
#pragma unroll
            for (uint32_t j = 0; j < ECNE_BW; ++j) w[j] = base + j * ECNE_WG < e1 && (fl[j] & 3) != 3;
            dr_mark4(J.dmk[DA_U], v, w, key);
        };
        big_walk(J.colA, J.flags, J.rpA[row], J.rpA[row + 1], ab);
        big_walk(J.colB, J.flags, J.rpB[row], J.rpB[row + 1], ab);
        big_walk(J.colC, J.flags, J.rpC[row], J.rpC[row + 1], [&](const uint32_t* v, const uint8_t* fl, uint32_t base, uint32_t e1) {
            bool wu[ECNE_BW], wb[ECNE_BW];
#pragma unroll
            for (uint32_t j = 0; j < ECNE_BW; ++j) { const bool on = base + j * ECNE_WG < e1; wu[j] = on && (fl[j] & 3) != 3; wb[j] = on && !(fl[j] & 1); }
            dr_mark4(J.dmk[DC_U], v, wu, key);
            dr_mark4(J.dmk[DA_U], v, wu, key);
            dr_mark4(J.dmk[DA_B], v, wb, key);
        });
    }
    return any_am ? 2u : 0u;
}
// a mark of the current epoch with a lower rank on plane p for any of up to four variables (dr_lower)
__device__ __forceinline__ bool dr_lower4(const Job& J, int p, const uint32_t* v, const bool* want, uint32_t key) {
    uint32_t x[ECNE_BW];
#pragma unroll
    for (uint32_t j = 0; j < ECNE_BW; ++j) x[j] = want[j] ? ld_agent(&J.dmk[p][v[j]]) : 0u;
    bool any = false;
#pragma unroll
    for (uint32_t j = 0; j < ECNE_BW; ++j) any |= want[j] && x[j] > key;
    return any;
}
__device__ __noinline__ void dr_big_p3(const Job& J, ChunkShared& S, uint32_t epoch, bool look_c) {
    for (uint32_t k = 0; k < ECNE_BIGK; ++k) {
        if ((S.dr_st[k] & 3u) != 1u) continue;       // pending and not unstable
        const uint32_t row = S.bl_row[k], key = dr_key(epoch, S.bl_rank[k]);
        uint32_t r = 0;
        if (look_c) {
            auto ab = [&](const uint32_t* v, const uint8_t* fl, uint32_t base, uint32_t e1) {
                bool w[ECNE_BW];
#pragma unroll
                for (uint32_t j = 0; j < ECNE_BW; ++j) w[j] = base + j * ECNE_WG < e1 && (fl[j] & 3) != 3;
                if (dr_lower4(J, DC_U, v, w, key)) r |= 4u;
            };
            big_walk(J.colA, J.flags, J.rpA[row], J.rpA[row + 1], ab);
            big_walk(J.colB, J.flags, J.rpB[row], J.rpB[row + 1], ab);
        }
        big_walk(J.colC, J.flags, J.rpC[row], J.rpC[row + 1], [&](const uint32_t* v, const uint8_t* fl, uint32_t base, uint32_t e1) {
            bool wu[ECNE_BW], wb[ECNE_BW], none[ECNE_BW];
#pragma unroll
            for (uint32_t j = 0; j < ECNE_BW; ++j) { const bool on = base + j * ECNE_WG < e1; wu[j] = on && (fl[j] & 3) != 3; wb[j] = on && look_c && !(fl[j] & 1); none[j] = false; }
            if (look_c && dr_lower4(J, DC_U, v, wu, key)) r |= 4u;
            if (dr_lower4(J, DA_U, v, wu, key)) r |= 8u;
            if (look_c && dr_lower4(J, DC_B, v, wb, key)) r |= 4u;
            (void)none;
        });
        if (r) atomicOr(&S.dr_st[k], r);
        if (r & 4u) atomicMin(&S.dcut, S.bl_rank[k]);
    }
}
__device__ __noinline__ void dr_big_exec(const Job& J, ChunkShared& S, uint32_t wgrank, uint32_t dcut) {
    for (uint32_t k = 0; k < ECNE_BIGK; ++k) {
        const uint32_t st = S.dr_st[k];
        if (!(st & 1u)) continue;
        const bool ready = !(st & (2u | 4u | 8u)) && S.bl_rank[k] < dcut;      // uniform
        if (ready) exec_big_row_wg(J, S, S.bl_row[k], big_ev(J, wgrank, k), &S.bl_nev[k]);
        __syncthreads();
        if (threadIdx.x == 0) S.dr_st[k] = ready ? 16u : 1u;
    }
    __syncthreads();
}

// All workgroups of the job. n <= J.nwg * ECNE_WG rows (one per lane). Returns nonzero on error; *out_c = rows that left the
// queue (the window, or what is in front of a long row the round does not take), *out_levels = levels the drain needed.
__device__ __noinline__ int queue_round_drain(const Job& J, ChunkShared& S, uint32_t wgrank, uint32_t head, uint32_t tail,
                                             uint32_t n, LaneCtr& C, uint32_t& my_pops, uint32_t& my_nnz, int* s_err,
                                             uint32_t* out_c, uint32_t* out_tail, uint32_t* out_levels) {
    const int tid = threadIdx.x, lane = lane_id(), w = wave_id();
    Counters* const ctr = J.ctr;
    const uint32_t g = wgrank * ECNE_WG + tid;
    const uint32_t r0 = ((uint32_t)w * J.nwg + wgrank) * 64u + (uint32_t)lane;     // ranks dealt out wavefront by wavefront, as in queue_round_multi
    unsigned long long mt_last = wall_clock64();
    int err;
    uint32_t row[2] = {0, 0}, nev[2] = {0, 0};
    uint32_t shape = 0, xv = 0;
    bool live = false, pending = r0 < n;
    if (pending) {
        row[0] = J.queue[(head + r0) & J.qmask];
        const RowInfo ri = J.rinfo[row[0]];
        shape = ri.shape;
        xv = ri.x;
        live = !J.solved[row[0]];
    }
    // the row's record (if it has one): loaded once, the flag bytes again at every level
    FastIn fin;
    bool rec_ok = false;
    if (pending && live && !(shape & SH_BIG)) {
        const ECNE_GLOBAL u32x4* const rec = as_global(reinterpret_cast<const u32x4*>(J.rec));
        const RowInfo ri = J.rinfo[row[0]];
        u32x4 w4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) w4[i] = rec[4u * row[0] + (uint32_t)i];
#pragma unroll
        for (int i = 0; i < 4; ++i) { fin.w[4 * i] = w4[i].x; fin.w[4 * i + 1] = w4[i].y; fin.w[4 * i + 2] = w4[i].z; fin.w[4 * i + 3] = w4[i].w; }
        fin.shape = ri.shape; fin.rx = ri.x; fin.kpos = ri.kpos; fin.kneg = ri.kneg; fin.k1 = ri.k1; fin.k2 = ri.k2;
        fin.nA = fin.w[0] & 0xFFu; fin.nB = (fin.w[0] >> 8) & 0xFFu; fin.nE = fin.nA + fin.nB + ((fin.w[0] >> 16) & 0xFFu);
        fin.xy = (ri.shape & (SH_R5 | SH_R4_T | SH_R4_T2 | SH_R3)) == (SH_R5 | SH_R4_T | SH_R4_T2);
        const bool f1 = (ri.shape & SH_HAS_AB) && !(ri.shape & SH_C_EMPTY);
        fin.f2 = (ri.shape & SH_C_EMPTY) != 0;
        fin.f4 = !(ri.shape & (SH_HAS_AB | SH_C_EMPTY | SH_R3 | SH_R4_T | SH_R4_T2 | SH_R5 | SH_R6));
        fin.live = true; fin.bigsum = false;
        rec_ok = (fin.w[0] >> 24) != 0 && (fin.xy || f1 || fin.f2 || fin.f4);
    }
    const ECNE_GLOBAL uint8_t* const Fg = as_global(J.flags);
    if (tid == 0) { S.cut = 0xFFFFFFFFu; S.dcut = 0xFFFFFFFFu; }
    if (tid < ECNE_BIGK) S.dr_st[tid] = 0;
    __syncthreads();
    uint32_t epoch = S.depoch;
    uint32_t n_eff = n, mycand = 0, bigsl = 0, level = 0;
    int myslot = -1;               // my row is a long row registered in this slot
    for (;;) {
        ++level;
#ifdef ECNE_ROUNDLOG
        unsigned long long rd_t[6]; rd_t[0] = wall_clock64(); rd_t[1] = rd_t[2] = rd_t[3] = rd_t[4] = rd_t[5] = rd_t[0];
#endif
        if (wgrank == 0 && (level & 15u) == 0) job_heartbeat(J);      // (a window of dependent rows drains one row per level: thousands of levels in one round)
        if (epoch >= ECNE_DRAIN_EPOCH_MAX) {
            // the epoch field is used up (once per 32 766 levels): wipe the planes; the marks of this round start over
            const uint32_t T = J.nwg * ECNE_WG;
            for (int p = 0; p < 6; ++p)
                for (uint32_t v = g; v <= J.nV; v += T) J.dmk[p][v] = 0;
            epoch = 0;
            if ((err = job_barrier(J, s_err))) return err;
        }
        ++epoch;
        const uint32_t key = dr_key(epoch, r0);
        const uint32_t par = level & 1u;
        // ------------------------------------------------------------------ P1: decide on the state as it is, mark X and A
        uint32_t kind = 0;          // 0 nothing to look at (dead / final no-op), 1 record decision, 2 general walk, 3 general walk of a no-op, 4 long row
        uint32_t fz_wva = 0, fz_cls = 0;          // (x == y rows write k1 / k2, bit checks x, products and sums one variable: fz_wva)
        if (pending) {
            if (shape & SH_BIG) {
                kind = live ? 4u : 0u;
                if (live && level == 1) {
                    if (!big_plain(shape) || !big_register(S, row[0], r0)) atomicMin(&S.cut, r0);
                }
            } else if (live) {
                bool fz = false;
                if (rec_ok) {
                    fin.flip_in = J.flip3[row[0]];
                    const bool walk = !fin.xy && !fin.f2;
#pragma unroll
                    for (uint32_t e = 0; e < 15; ++e) fin.fl[e] = (walk && e < fin.nE) ? Fg[fin.w[1 + e]] : (uint8_t)3;
                    fin.fa = fin.fb = fin.fx = 3;
                    if (fin.xy) { fin.fa = Fg[fin.k1]; fin.fb = Fg[fin.k2]; }
                    if (fin.f2 && (fin.shape & SH_R2)) fin.fx = Fg[fin.rx];
                    FastOut D;
                    fast_decide(J, fin, D);
                    if (!D.slow) {
                        fz = true;
                        fz_wva = D.wva;
                        const uint8_t ia = fin.xy ? fin.fa : fin.f2 ? fin.fx : (uint8_t)(D.wfa & ~3u), ib = fin.fb;     // flag bytes before (products / sums only set bits 0, 1)
                        if (D.wa) fz_cls |= (((D.wfa ^ ia) & 3u) ? 1u : 0u) | ((((D.wfa ^ ia) & ~3u) || D.a01 || D.xa_w || D.r2) ? 2u : 0u);
                        if (D.wb) fz_cls |= (((D.wfb ^ ib) & 3u) ? 4u : 0u) | ((((D.wfb ^ ib) & ~3u) || D.b01 || D.xb_w) ? 8u : 0u);
                    }
                }
                if (fz) kind = 1;
                else {
                    const RowInfo ri = J.rinfo[row[0]];
                    bool nb = false;
                    if (row_is_noop(J, row[0], ri, nb)) kind = nb ? 3u : 0u;
                    else kind = 2;
                }
            }
        }
        // the sets of a record row: f(variable, classes it can access, classes it writes now, classes it could ever write)
        auto fz_each = [&](auto f) {
            if (fin.xy) { f(fin.k1, 3u, fz_cls & 3u, 3u); f(fin.k2, 3u, (fz_cls >> 2) & 3u, 3u); }
            else if (fin.f2) { if (fin.shape & SH_R2) f(fin.rx, 3u, fz_cls & 3u, 3u); }
            else {
                const uint32_t acc = fin.f4 ? 3u : 1u;
#pragma unroll
                for (uint32_t e = 0; e < 15; ++e)
                    if (e < fin.nE && (fin.fl[e] & 3) != 3) {
                        const uint32_t v = fin.w[1 + e];
                        f(v, acc, ((fz_cls & 3u) && v == fz_wva) ? (fz_cls & 3u) : 0u, e >= fin.nA + fin.nB ? 1u : 0u);
                    }
            }
        };
        if (kind == 1)
            fz_each([&](uint32_t v, uint32_t, uint32_t wx, uint32_t) {
                if (wx & 1) dr_mark_now(J.dmk[DX_U], v, key);
                if (wx & 2) dr_mark_now(J.dmk[DX_B], v, key);
            });
        else if (kind == 2) dr_gen_mark(J, row[0], shape, xv, key);
        __syncthreads();
        if (level == 1) {
            if (tid == 0 && S.cut != 0xFFFFFFFFu) atomicMin(&ctr->q_cut, S.cut);
            if (tid < ECNE_BIGK && S.bl_rank[tid] != 0xFFFFFFFFu) S.dr_st[tid] = 1u;
            __syncthreads();
            if (kind == 4) myslot = big_slot_of(S, r0);
        }
        if (S.bl_any) dr_big_p1(J, S, epoch);
#ifdef ECNE_ROUNDLOG
        rd_t[1] = wall_clock64();
#endif
        if ((err = job_barrier(J, s_err))) return err;
        MTICK(0);
#ifdef ECNE_ROUNDLOG
        rd_t[2] = wall_clock64();
#endif
        if (level == 1) {
            const uint32_t qc = ld_agent(&ctr->q_cut);
            if (qc < n_eff) n_eff = qc;           // the window ends in front of a long row the round does not take (nothing has run yet)
            if (r0 >= n_eff) { pending = false; kind = 0; }
            if (tid < ECNE_BIGK && S.bl_rank[tid] != 0xFFFFFFFFu && S.bl_rank[tid] >= n_eff) { S.dr_st[tid] = 0; S.bl_rank[tid] = 0xFFFFFFFFu; }
            __syncthreads();
        }
        // ------------------------------------------------------------------ P2: look at X. A lower mark on something the row accesses: unstable
        // (marks what it could ever write in C, what it can access in A); a higher one: the access is contested, the row says so in A
        bool unstable = false, amark = false;
        if (kind == 1) {
            fz_each([&](uint32_t v, uint32_t acc, uint32_t, uint32_t) {
                if (acc & 1) { const uint32_t s = dr_see(J.dmk[DX_U], v, key); if (s == 1) unstable = true; else if (s == 2) { dr_mark(J.dmk[DA_U], v, key); amark = true; } }
                if (acc & 2) { const uint32_t s = dr_see(J.dmk[DX_B], v, key); if (s == 1) unstable = true; else if (s == 2) { dr_mark(J.dmk[DA_B], v, key); amark = true; } }
            });
            if (unstable)
                fz_each([&](uint32_t v, uint32_t acc, uint32_t, uint32_t wc) {
                    if (wc & 1) dr_mark(J.dmk[DC_U], v, key);
                    if (wc & 2) dr_mark(J.dmk[DC_B], v, key);
                    if (acc & 1) dr_mark(J.dmk[DA_U], v, key);
                    if (acc & 2) dr_mark(J.dmk[DA_B], v, key);
                });
        } else if (kind == 2 || kind == 3) {
            const uint32_t r = dr_gen_p2(J, row[0], shape, xv, key);
            unstable = (r & 1u) != 0; amark = (r & 2u) != 0;
            if (unstable) dr_gen_mark_unstable(J, row[0], shape, xv, key);
        }
        if (tid == 0) S.dcut = 0xFFFFFFFFu;
        uint32_t big_am = 0;
        if (S.bl_any) {
            big_am = dr_big_p2(J, S, epoch);
            __syncthreads();
            if (pending && kind == 4 && myslot >= 0) unstable = (S.dr_st[myslot] & 2u) != 0;
        }
        {
            // two bits per workgroup: is anybody unstable (then C has to be looked at, P3)? did anybody mark A (then writers look at it)?
            const int any_unst = __syncthreads_or(pending && unstable ? 1 : 0);
            const int any_am = __syncthreads_or(pending && (amark || unstable) ? 1 : 0) | (int)big_am;
            if (tid == 0 && (any_unst || any_am)) atomicOr(&ctr->d_flag[par], (any_unst ? 1u : 0u) | (any_am ? 2u : 0u));
            if (g == 0) { ctr->d_flag[par ^ 1u] = 0; ctr->d_pend2[par ^ 1u] = 0; ctr->d_cut[par ^ 1u] = 0xFFFFFFFFu; }
        }
        if ((err = job_barrier(J, s_err))) return err;
        MTICK(1);
#ifdef ECNE_ROUNDLOG
        rd_t[3] = wall_clock64();
#endif
        const uint32_t lflag = ld_agent(&ctr->d_flag[par]);
        const bool slow_level = (lflag & 1u) != 0;      // somebody is unstable: its conservative marks have to be looked at
        uint32_t dcut = 0xFFFFFFFFu;
        bool demoted = false, waiting = false;
        if (lflag) {
            // -------------------------------------------------------------- P3: demoted by a lower unstable row / waiting for a lower accessor
            if (pending && !unstable) {
                uint32_t r = 0;
                if (kind == 1)
                    fz_each([&](uint32_t v, uint32_t acc, uint32_t wx, uint32_t) {
                        if (slow_level) {
                            if ((acc & 1) && dr_lower(J.dmk[DC_U], v, key)) r |= 1u;
                            if ((acc & 2) && dr_lower(J.dmk[DC_B], v, key)) r |= 1u;
                        }
                        if ((wx & 1) && dr_lower(J.dmk[DA_U], v, key)) r |= 2u;
                        if ((wx & 2) && dr_lower(J.dmk[DA_B], v, key)) r |= 2u;
                    });
                else if (kind == 2 || kind == 3) r = dr_gen_p3(J, row[0], shape, xv, key, kind == 3, slow_level);
                demoted = (r & 1u) != 0; waiting = (r & 2u) != 0;
            }
            if (S.bl_any) {
                dr_big_p3(J, S, epoch, slow_level);
                __syncthreads();
                if (pending && kind == 4 && myslot >= 0) waiting = (S.dr_st[myslot] & 8u) != 0;
            }
            if (slow_level) {
                { const uint32_t wm = wave_min(demoted ? r0 : 0xFFFFFFFFu); if (lane == 0 && wm != 0xFFFFFFFFu) atomicMin(&S.dcut, wm); }
                __syncthreads();
                if (tid == 0 && S.dcut != 0xFFFFFFFFu) atomicMin(&ctr->d_cut[par], S.dcut);
                if ((err = job_barrier(J, s_err))) return err;
                dcut = ld_agent(&ctr->d_cut[par]);
            }
        }
#ifdef ECNE_ROUNDLOG
        rd_t[4] = wall_clock64();
#endif
        // ------------------------------------------------------------------ run what is ready
        if (pending && kind != 4 && !unstable && !waiting && !demoted && r0 < dcut) {
            pending = false;
            J.inq[row[0]] = (uint16_t)2;
            J.prank[row[0]] = r0;
            my_pops++;
            uint32_t* ev = J.evbuf + (size_t)r0 * ECNE_EVCAP;
            if (kind == 1) {
                my_nnz += fin.nE;
                FastOut D;
                fast_decide(J, fin, D);      // the decision again: its inputs are what they were (the row is stable)
                if (D.wa) J.flags[D.wva] = D.wfa;
                if (D.wb) J.flags[D.wvb] = D.wfb;
                if (D.a01) { st256(J.lb + 4ull * D.wva, fp::make(0)); st256(J.ub + 4ull * D.wva, fp::make(1)); }
                if (D.b01) { st256(J.lb + 4ull * D.wvb, fp::make(0)); st256(J.ub + 4ull * D.wvb, fp::make(1)); }
                if (D.xa_w) { st256(J.lb + 4ull * D.wva, D.xlb0); st256(J.ub + 4ull * D.wva, D.xub0); }
                if (D.xb_w) { st256(J.lb + 4ull * D.wvb, D.xlb1); st256(J.ub + 4ull * D.wvb, D.xub1); }
                if (D.r2) {        // make_values (:921-927)
                    const uint32_t validx = J.rinfo[row[0]].validx;
                    st256(J.values + 8ull * fin.rx, ld256(J.vals + 4ull * validx));
                    st256(J.values + 8ull * fin.rx + 4, ld256(J.vals + 4ull * (validx + 1)));
                    J.nvalues[fin.rx] = 2;
                    J.abz[fin.rx] = -1;
                    J.solved[row[0]] = 1;
                }
                if (D.flip_w) J.flip3[row[0]] = D.flip_new;
                C.steps += D.d_steps; C.nuniq += D.d_nuniq;
                C.hits[0] += D.d_h0; C.hits[1] += D.d_h1; C.hits[3] += D.d_h3; C.hits[4] += D.d_h4;
                nev[0] = D.nev;
#pragma unroll
                for (uint32_t e = 0; e < 5; ++e) if (e < D.nev) { ev[e] = D.ev[e]; mycand += J.fo_ptr[D.ev[e] + 1] - J.fo_ptr[D.ev[e]]; }
            } else {
                my_nnz += (J.rpA[row[0] + 1] - J.rpA[row[0]]) + (J.rpB[row[0] + 1] - J.rpB[row[0]]) + (J.rpC[row[0] + 1] - J.rpC[row[0]]);
                if (live) {
                    if (kind == 2) { C.rank = head + r0; exec_row_lane(J, row[0], ev, nev[0], C); }
                    else if ((shape & SH_R4_T) && (shape & SH_R4_T2)) J.flip3[row[0]] ^= 1;      // a no-op pop of an x == y row: its R4 orientation flips
                }
                for (uint32_t e = 0; e < nev[0]; ++e) mycand += J.fo_ptr[ev[e] + 1] - J.fo_ptr[ev[e]];
            }
            J.evcnt[r0] = nev[0];           // for the sequential replay fallback (a dense array: the last word of the rank's 800-byte event slot was a dirty line per pop)
        }
        if (S.bl_any) {
            dr_big_exec(J, S, wgrank, dcut);
            if (pending && kind == 4 && myslot >= 0 && (S.dr_st[myslot] & 16u)) {
                pending = false;
                bigsl = 1u;
                J.inq[row[0]] = (uint16_t)2;
                J.prank[row[0]] = r0;
                my_pops++;
                my_nnz += (J.rpA[row[0] + 1] - J.rpA[row[0]]) + (J.rpB[row[0] + 1] - J.rpB[row[0]]) + (J.rpC[row[0] + 1] - J.rpC[row[0]]);
            }
            __syncthreads();
            if (tid < ECNE_BIGK && (S.dr_st[tid] & 16u)) S.dr_st[tid] = 0;
        }
        // nobody marked A or C: nobody was unstable, demoted or made to wait -- everything has run, the window has drained
        // (multi_finish's first barrier orders these stores before anybody reads them)
#ifdef ECNE_ROUNDLOG
        rd_t[5] = wall_clock64();
        if (g == 0) printf("RDT level %u p1 %llu bar %llu p2+bar %llu p3 %llu run %llu bl_any %u\n", level, rd_t[1] - rd_t[0], rd_t[2] - rd_t[1], rd_t[3] - rd_t[2], rd_t[4] - rd_t[3], rd_t[5] - rd_t[4], (unsigned)S.bl_any);
#endif
        if (!lflag) break;
        {
            const int left = __syncthreads_count(pending ? 1 : 0);
            if (tid == 0 && left) atomicAdd(&ctr->d_pend2[par], (unsigned int)left);
        }
        if ((err = job_barrier(J, s_err))) return err;
#ifdef ECNE_ROUNDLOG
        if (g == 0) printf("RD level %u n %u pending %u flag %u\n", level, n_eff, ld_agent(&ctr->d_pend2[par]), lflag);
#endif
        if (ld_agent(&ctr->d_pend2[par]) == 0) {
            if (g == 0) { ctr->d_flag[par] = 0; ctr->d_cut[par] = 0xFFFFFFFFu; }     // (all read before this level's last barrier)
            break;
        }
    }
    if (tid == 0) S.depoch = epoch;
    if (g == 0) { S.mt[6] += level; S.mt[7] += 1; }
    __syncthreads();
    *out_levels = level;
    return multi_finish(J, S, wgrank, head, tail, n_eff, 1u, r0, row, nev, bigsl, mycand, false, mt_last, s_err, out_c, out_tail);
}

}  // namespace ecne
