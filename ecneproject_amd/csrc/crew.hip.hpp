// crew.hip.hpp — crew rounds: the narrow dependency levels of a deep circuit (Poseidon / MiMC / EdDSA: 2-8 queued rows, hundreds to
// thousands of levels) with ONE WAVEFRONT PER QUEUED ROW -- all eight wavefronts of a single-workgroup job, lanes across the row's
// entries as the chain executor has them, the window committed in queue order.
// Part of the gfx950 device code of libecne_hip (see kernels.hip.hpp for the overview).
//
// Why: a level round (level.hip.hpp) puts one LANE on each queued row, so the instruction stream of a level is as long as the
// longest row's walk whatever the width of the level -- ~1 000 instructions at ~7 cycles each on the one busy wavefront, 2.9-3.7 us
// per level with 2-8 of 64 lanes at work (profiles/r04_level_round_stages.txt). The chain executor (chain.hip.hpp) pops ONE row in
// ~300 instructions with its lanes across the row's entries. A crew round runs that short stream on up to eight rows at once, one
// wavefront each, and pays for it with workgroup barriers (~0.05 us each) and the LDS round trips of what the wavefronts tell each other:
//   1. wavefront r takes the row at queue position head + r from the LDS mirror of the ring (level.hip.hpp's mirror);
//   2. record + descriptor + solved / orientation bytes in one trip to the L2 (lanes 0..15 / 16..23), then lane 1 + e the flag
//      byte of entry e (LDS) and, speculatively, the variable's inline fan-out foi[v] -- chain_pops() statement for statement;
//   3. the pop is DECIDED with ballots, nothing is written: up to two flag bytes (+ bounds [0,1], the two roots of a bit check, the
//      orientation byte), the REQUEUE candidates of its events in emission order (lane i holds candidate i), counters;
//   4. published before the one barrier of the dependency test: write marks in one of two hashed LDS tables (minimum rank per slot; a
//      collision ends the prefix early, never wrongly late; the tables alternate with every pass of the loop, so that a wavefront takes its
//      marks back while the others already set theirs in the other table), the read set (the variables whose flag bytes the pop has
//      read), the row, status and candidate count | barrier | EVERY wavefront works out the whole prefix by itself: lane 8 r + j looks
//      entries j and j + 8 of rank r's read set up in the marks -- blocked iff an EARLIER rank writes what the row reads --, the prefix
//      ends in front of the first blocked row (or the first row this loop does not take);
//   5. the wavefronts of the prefix commit their pops and write their candidates (rank << 24 | target, in (rank, emission, position)
//      order) to the LDS list | barrier | wavefront 0 resolves the pushes exactly as a level round does (the rows of the prefix carry
//      2 + rank in their in_queue tag meanwhile, the lowest eligible candidate per target wins, winners go to the mirror) | barrier. A
//      round without candidates ends at the commit barrier: two to three barriers per round, 2.0-2.2 us against 2.9-3.7 for a level round.
// What the loop does not take -- a row without a record that its watched pair does not settle, another shape, a bound of the third
// kind, R7 / R8 in reach, errors -- ends the window in front of it; at rank 0 wavefront 0 pops that one row with the general
// executor right here (the others wait at the barrier), a live long row goes back to the caller (popped by the whole workgroup).
// More than ECNE_CREW_MAX rows queued: LV_FAT -- the caller runs level rounds (64 rows per round) until the frontier is narrow again.
// The schedule is a prefix of the FIFO order in every round and only true dependencies cut it, so the result is the sequential one
// (DESIGN.md "Schedule"); reference: the queue loop /root/reference/src/R1CSConstraintSolver.jl:805-1349.
#pragma once
#include "level.hip.hpp"

namespace ecne {

static_assert(ECNE_CREW_MAX <= ECNE_NWAVES, "one wavefront per row");
#define ECNE_CREW_X (ECNE_LV_MARKS + ECNE_LV_QM + ECNE_LV_CAND)      // exchange words behind the level rounds' tables
static_assert(4u * (ECNE_CREW_X + 32u + 128u) <= ECNE_W2_BYTES, "the crew's exchange words live in the fast wavefront round's LDS block");
static_assert(ECNE_CREW_MAX * 64u <= ECNE_LV_CAND, "every row of a crew round can have 64 candidates");
__device__ __forceinline__ bool crew_on(const Job& J) { return (J.lv_off & 3u) == 0; }

// ALL threads of the workgroup (uniform control flow: every decision is taken from words every wavefront reads from LDS).
// head / tail: the queue cursors, in and out (the same values on every thread). Returns why it stopped (LV_*); *n_rounds = rounds
// run, *n_general = rows popped by the general executor in between, *big_out = 1: the row at the head is a live long row.
__device__ __noinline__ uint32_t crew_rounds(const Job& J, ChunkShared& S, uint32_t& head_io, uint32_t& tail_io, uint32_t max_rounds, bool wide_ok,
                                             LaneCtr& C, uint32_t& my_pops, uint32_t& my_nnz, uint32_t* n_rounds, uint32_t* n_general, uint32_t* big_out, bool warm, uint32_t& mtop_io) {
    // (the wavefront's number as a scalar: everything that depends on it -- is there a row for me? am I in the prefix? -- is a scalar branch then)
    const uint32_t lane = (uint32_t)lane_id(), rank = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), tid = threadIdx.x;
    const uint32_t NONE = 0xFFFFFFFFu;
    auto uni = [](const void* p) -> uint64_t {
        const uint64_t x = (uint64_t)p;
        return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(x >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)x);
    };
    const ECNE_GLOBAL uint32_t* const rec = (const ECNE_GLOBAL uint32_t*)uni(J.rec);
    const ECNE_GLOBAL u32x4* const foi = (const ECNE_GLOBAL u32x4*)uni(J.foi);
    const ECNE_GLOBAL uint32_t* const fo_rows = (const ECNE_GLOBAL uint32_t*)uni(J.fo_rows);
    const ECNE_GLOBAL uint32_t* const rinfo = (const ECNE_GLOBAL uint32_t*)uni(J.rinfo);
    ECNE_GLOBAL uint32_t* const queue = (ECNE_GLOBAL uint32_t*)uni(J.queue);
    const uint32_t qmask = (uint32_t)__builtin_amdgcn_readfirstlane((int)J.qmask);
    ECNE_GLOBAL uint8_t* const solved = (ECNE_GLOBAL uint8_t*)uni(J.solved);
    uint8_t* const F = (uint8_t*)(ecne_dyn_lds + J.lds_flags_off);
    uint16_t* const Q = (uint16_t*)(ecne_dyn_lds + J.lds_inq_off);
    const bool flip_lds = J.lds_flip_off != 0xFFFFFFFFu;
    uint8_t* const flipL = (uint8_t*)(ecne_dyn_lds + (flip_lds ? J.lds_flip_off : 0u));
    ECNE_GLOBAL uint8_t* const flipG = flip_lds ? (ECNE_GLOBAL uint8_t*)nullptr : (ECNE_GLOBAL uint8_t*)uni(J.flip3);
    uint32_t* const tb = (uint32_t*)(ecne_dyn_lds + J.lds_w2_off);
    uint32_t* const wm = tb;                                   // write marks
    uint32_t* const qm = tb + ECNE_LV_MARKS;                   // queue mirror
    uint32_t* const cl = tb + ECNE_LV_MARKS + ECNE_LV_QM;      // candidate list
    uint32_t* const X_ROW = tb + ECNE_CREW_X;                  // the window's rows
    uint32_t* const X_ST = X_ROW + 8;                          // per rank: 0 taken, 1 the general executor's, 2 a live long row, 3 more than 64 candidates; candidates << 8
    uint32_t* const X_CTL = X_ROW + 24;                        // wavefront 0's results: tail, mirror top, error
    uint32_t* const X_RV = X_ROW + 32;                         // per rank 16 words: the variables whose flag bytes the pop has read (lane = word), 0xFFFFFFFF = none
    // write marks: two tables of 128 slots, by round parity -- a round's marks are taken back while the next round already sets its own
    auto wslot = [](uint32_t v, uint32_t par) -> uint32_t { return ((v * 2654435761u) >> 25) | (par << 7); };

    auto sc_ = [](uint32_t x) -> uint32_t { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); };      // a value every lane holds, as a scalar
    uint32_t head = sc_(head_io), tail = sc_(tail_io), rounds = 0, gdone = 0, why = LV_EMPTY, big = 0;
    max_rounds = sc_(max_rounds);
    const bool wide = sc_(wide_ok ? 1u : 0u) != 0;
    // (counters are wave-uniform here: lane 0 of every wavefront folds its own into the caller's per-thread counters at the exit)
    // steps, nuniq, hits[0], hits[1], hits[3], hits[4] travel as six 10-bit fields of one 64-bit word (dd per pop, acc per 128 rounds)
    uint32_t c_steps = 0, c_nuniq = 0, c_h0 = 0, c_h1 = 0, c_h3 = 0, c_h4 = 0, c_pops = 0, c_nnz = 0;
    unsigned long long acc = 0;
    const unsigned long long D_ST = 1ull, D_NU = 1ull << 10, D_H0 = 1ull << 20, D_H1 = 1ull << 30, D_H3 = 1ull << 40, D_H4 = 1ull << 50;
    auto fold = [&]() {
        c_steps += (uint32_t)(acc & 1023u); c_nuniq += (uint32_t)((acc >> 10) & 1023u); c_h0 += (uint32_t)((acc >> 20) & 1023u);
        c_h1 += (uint32_t)((acc >> 30) & 1023u); c_h3 += (uint32_t)((acc >> 40) & 1023u); c_h4 += (uint32_t)((acc >> 50) & 1023u);
        acc = 0;
    };
    uint32_t lvl = 0;
    uint32_t pm0 = NONE, pm1 = NONE;             // my write marks of the last round (slots), taken back at the top of the next one
    // the general executor's counters (wavefront 0)
    unsigned long long g_st = 0, g_nu = 0, g_ht[16];
    for (int i = 0; i < 16; ++i) g_ht[i] = 0;
    uint32_t g_nnz = 0;
#ifdef ECNE_LVPROF
    unsigned long long* const prof = &S.sd[0];
    unsigned long long cwt_last = wall_clock64();
#define CWT(k) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); if (tid == 0) { const unsigned long long t_ = wall_clock64(); prof[k] += t_ - cwt_last; cwt_last = t_; } } while (0)
#else
#define CWT(k) do { } while (0)
#endif
    // ---- entry: the tables of the fast wavefront round become ours (clean on entry, restored on exit); the mirror is filled from the ring
    // (warm: the level rounds left marks and mirror as this loop keeps them)
    uint32_t mtop;
    if (sc_(warm ? 1u : 0u)) mtop = sc_(mtop_io);
    else {
        if (tid < ECNE_LV_MARKS) wm[tid] = NONE;
        mtop = head + ((tail - head) < ECNE_LV_QM ? (tail - head) : ECNE_LV_QM);
        if (tid < mtop - head) qm[(head + tid) & (ECNE_LV_QM - 1)] = queue[(head + tid) & qmask];
        __syncthreads();
    }
    CWT(0);        // entry
    for (;;) {
        // (my marks of the last round: everybody has looked at them -- a barrier lies behind us -- and this round's go to the other table)
        if (lane == 0) { if (pm0 != NONE) { wm[pm0] = NONE; pm0 = NONE; } if (pm1 != NONE) { wm[pm1] = NONE; pm1 = NONE; } }
        if (head == tail) { why = LV_EMPTY; break; }
        const uint32_t avail = tail - head;
        if (!wide && avail > ECNE_LV_WIDE_AVAIL) { why = LV_WIDE; break; }
        if (rounds + gdone >= max_rounds) { why = LV_ROUNDS; break; }
        if (avail > ECNE_CREW_MAX) { why = LV_FAT; break; }
        if (head + avail > mtop) {
            // the window reaches beyond the mirror (the mirror was full at some point and pushes went to the ring): fill it again
            __syncthreads();
            if (tid < mtop - head) queue[(head + tid) & qmask] = qm[(head + tid) & (ECNE_LV_QM - 1)];
            __syncthreads();
            mtop = head + avail;
            if (tid < avail) qm[(head + tid) & (ECNE_LV_QM - 1)] = queue[(head + tid) & qmask];
            __syncthreads();
        }
        const uint32_t n = avail;
        const bool mine = rank < n;
        // ---- 1-3: my row, decided. Outputs (wave-uniform): ob -- bit 0 the pop does something, 1 / 2 flag byte of wv0 / wv1 written, 3 / 4 their
        // bounds become [0,1], 5 bit check solved (values = the two roots), 6 orientation byte written, 7 its new value; dd -- counters, 4 bits each
        uint32_t row = 0, slowk = 0, nnz_d = 0, ob = 0;
        unsigned long long dd = 0;
        uint32_t rv = NONE;                          // per lane: a variable whose flag byte this pop has read
        uint32_t cv = 0, ncand = 0;                  // lane i: target row of candidate i
        uint32_t wv0 = 0, wv1 = 0, wf = 0, validx = 0;
        if (mine) {
            row = (uint32_t)__builtin_amdgcn_readfirstlane((int)qm[(head + rank) & (ECNE_LV_QM - 1)]);
            // everything the pop needs to know about the row, in flight together: record (lanes 0..15), descriptor (16..23), solved and orientation bytes
            uint32_t w = 0;
            const ECNE_GLOBAL uint32_t* const wp = lane < 16u ? rec + (16u * row + lane) : rinfo + (8u * row + (lane - 16u));
            if (lane < 24u) w = *wp;
            const uint32_t sv_ = solved[row];
            const uint32_t fl_ = flip_lds ? flipL[row] : flipG[row];
            const uint32_t w0 = rdlane(w, 0), shape = rdlane(w, 16);
            const uint32_t is_solved = rdlane(sv_, 0), flip_in = rdlane(fl_, 0);
            CWT(1);        // record + descriptor
            const uint32_t nA = w0 & 0xFFu, nB = (w0 >> 8) & 0xFFu, nCc = (w0 >> 16) & 0xFFu, nn = nA + nB + nCc;
            const bool recok = (w0 >> 24) != 0;
            const uint32_t e = lane - 1u;                    // my entry of the row, if any
            const bool valid = recok && e < nn;
            const uint32_t v = valid ? w : 1u;
            uint32_t f = 3;
            u32x4 fo = {0, 0, 0, 0};
            if (valid) { f = F[v]; fo = foi[v]; }
            CWT(2);        // flag bytes + fan-out lists
            const bool xy = (shape & (SH_R5 | SH_R4_T | SH_R4_T2 | SH_R3)) == (SH_R5 | SH_R4_T | SH_R4_T2);
            const bool f1 = (shape & SH_HAS_AB) && !(shape & SH_C_EMPTY);
            const bool f2 = (shape & SH_C_EMPTY) != 0;
            const bool f4 = !(shape & (SH_HAS_AB | SH_C_EMPTY | SH_R3 | SH_R4_T | SH_R4_T2 | SH_R5 | SH_R6));
            const bool live = is_solved == 0;
            if (!recok) {
                // a long linear row whose watched pair still says why its pop does nothing (fastrow.hip.hpp, long_row_walk); anything else
                // without a record is the general executor's (a live long row: the whole workgroup's)
                const bool lr4 = long_r4(shape);
                const uint32_t lenC = rdlane(w, 23);
                bool bl_nop = false;
                if (live && (f4 || lr4) && lenC > 15u) {
                    const uint32_t h0 = rdlane(w, 1), h1 = rdlane(w, 2);
                    const bool watched = h0 < 0xFFFFFFFEu;
                    const uint32_t g0_ = F[watched ? h0 : 0u], g1_ = F[watched ? h1 : 0u];
                    const uint32_t g0 = rdlane(g0_, 0), g1 = rdlane(g1_, 0);
                    bl_nop = (h0 == 0xFFFFFFFEu && f4) || (watched && long_watch_holds((uint8_t)g0, (uint8_t)g1, lr4));
                    if (bl_nop && watched) rv = lane == 1u ? h0 : lane == 2u ? h1 : NONE;
                    if (!bl_nop && lr4 && h0 == 0xFFFFFFFEu) {
                        // a decomposition all of whose terms are unique: nothing happens unless R4 still has the pivot's bounds to cut (long_r4_done)
                        const uint32_t kpos = rdlane(w, 18), kneg = rdlane(w, 19);
                        bl_nop = long_r4_done(J, shape, kpos, kneg, lenC, h0);
                        if (bl_nop) rv = lane == 1u ? long_r4_pivot(shape, kpos, kneg) : NONE;
                    }
                }
                if (bl_nop) nnz_d = lenC;
                else slowk = ((shape & SH_BIG) && live) ? 2u : 1u;
            } else if (!(xy || f1 || f2 || f4)) slowk = 1u;
            else if (!live) nnz_d = nn;
            else {
                nnz_d = nn;
                ob = 1u;
                rv = valid ? v : NONE;
                // REQUEUE of the variable lane `src` owns: its row list becomes candidates ncand .. ncand + n - 1 (lane i = candidate i)
                auto emit = [&](uint32_t src) {
                    const uint32_t en = rdlane(fo.x, src), e0 = rdlane(fo.y, src);
                    if (en <= 3u) {
                        const uint32_t e1 = rdlane(fo.z, src), e2 = rdlane(fo.w, src);
                        const uint32_t i = lane - ncand;
                        if (i < en) cv = i == 0u ? e0 : i == 1u ? e1 : e2;
                    } else {
                        const uint32_t i = lane - ncand;             // (lanes below ncand wrap around: not < en)
                        if (i < en) cv = fo_rows[e0 + i];
                    }
                    ncand += en;
                };
                const bool inC = valid && e >= nA + nB;
                const uint64_t m_nuab = __ballot(valid && !inC && !(f & 1u));
                const uint64_t m_nuc = __ballot(inC && !(f & 1u));
                // ---- R1 check_unique (:827-873), every fast shape
                bool r1_fired = false;
                uint32_t r1_src = 0;
                if (!m_nuab && __popcll(m_nuc) == 1) {
                    r1_src = (uint32_t)(__ffsll((long long)m_nuc) - 1);
                    wv0 = rdlane(v, r1_src); wf = rdlane(f, r1_src) | 3u; ob |= 2u;
                    dd += D_ST + D_NU + D_H0;
                    emit(r1_src);
                    r1_fired = true;
                }
                if (f2) {
                    // ---- R2 check_quadratic (:875-942); C is empty, so nothing else can apply to this row. Errors: the general executor raises them
                    if (shape & SH_R2_BOUNDSERR) slowk = 1u;
                    else if (shape & SH_R2) {
                        const uint32_t x = rdlane(w, 17);
                        const uint64_t mx = __ballot(valid && v == x);
                        const uint32_t src = (uint32_t)(__ffsll((long long)mx) - 1);     // a lane that owns x (x is in A or B)
                        const uint32_t fx = rdlane(f, src);
                        if (!(fx & 2u)) {
                            if (shape & SH_R2_DIV0) slowk = 1u;
                            else {
                                validx = rdlane(w, 22);
                                uint32_t nf = (fx | 2u) & ~16u;
                                ob |= 2u | 32u;
                                if (shape & SH_R2_IS01) { nf = (nf & ~12u) | 4u; ob |= 8u; }      // make_bounds (:923-927)
                                wv0 = x; wf = nf;
                                emit(src);
                                dd += D_ST + D_H1;
                            }
                        }
                    }
                } else if (f1) {
                    // (:944-946) a non-zero A or B: R3..R8 never run
                } else if (xy) {
                    // x == y (:991-1146 with l == 2); lanes 1 and 2 own the two variables in C order
                    const uint32_t k1 = rdlane(w, 20), k2 = rdlane(w, 21);
                    const uint32_t fl1 = rdlane(f, 1), fl2 = rdlane(f, 2);
                    if (((fl1 | fl2) & 8u) || k1 == k2 || nn != 2u) slowk = 1u;      // a bound of the third kind: on the limbs, the general executor
                    else {
                        const bool sw = (shape & SH_R56_SWAP) != 0;        // C order starts with k2
                        const uint32_t l1 = sw ? 2u : 1u, l2 = sw ? 1u : 2u;        // lanes owning k1 / k2
                        const uint32_t fa_in = sw ? fl2 : fl1, fb_in = sw ? fl1 : fl2;
                        uint32_t fa = fa_in, fb = fb_in;
                        if (r1_fired) { if (r1_src == l1) fa |= 3u; else fb |= 3u; }      // (R1's update included)
                        const uint32_t kpos = rdlane(w, 18), kneg = rdlane(w, 19);
                        // R4 (:991-1076), l == 2: the row is negated on every visit, the pivot alternates
                        {
                            const uint32_t flip_new = flip_in ^ 1u;
                            ob |= 64u | (flip_new << 7);
                            const uint32_t new_key = flip_new ? kneg : kpos;
                            const bool n_is_a = new_key == k1;
                            uint32_t fn = n_is_a ? fa : fb, fo_ = n_is_a ? fb : fa;
                            if (fo_ & 4u) {                                 // the other variable has bounds exactly [0,1]
                                if (!(fn & 4u)) {                           // pivot still [0,p-1]: ub > 1 -> [0,1]  (:1035-1046)
                                    ob |= n_is_a ? 8u : 16u;
                                    fn = (fn & ~12u) | 4u | 2u;
                                    dd += D_ST + D_H3;
                                    emit(n_is_a ? l1 : l2);
                                }
                                if ((fn & 1u) && !(fo_ & 1u)) {             // pivot unique: the other one becomes unique (:1049-1067)
                                    fo_ |= 3u;
                                    dd += D_ST + D_NU + D_H3;
                                    emit(n_is_a ? l2 : l1);
                                }
                            }
                            if (n_is_a) { fa = fn; fb = fo_; } else { fb = fn; fa = fo_; }
                        }
                        // R5 (:1078-1146): bounds are [0,1] or [0,p-1] here, equal iff the class bits agree
                        if (((fa ^ fb) & 4u) || ((fa ^ fb) & 1u)) {
                            bool cha = false, chb = false;
                            if ((fa ^ fb) & 1u) { fa |= 3u; dd += 2 * D_NU; cha = chb = true; }        // key_1 written twice (sic, :1107-1108)
                            const bool wa = ((fa ^ fb) & 4u) && !(fa & 4u), wb = ((fa ^ fb) & 4u) && !(fb & 4u);
                            if (wa) { fa = (fa & ~12u) | 4u | 2u; ob |= 8u; }
                            if (wb) { fb = (fb & ~12u) | 4u | 2u; ob |= 16u; }
                            cha |= wa; chb |= wb;
                            const uint32_t nset = (cha ? 1u : 0u) + (chb ? 1u : 0u);
                            dd += nset * D_ST;
                            if (nset) dd += D_H4;
                            if (sw) { if (chb) emit(l2); if (cha) emit(l1); }
                            else { if (cha) emit(l1); if (chb) emit(l2); }
                        }
                        wv0 = k1; wv1 = k2; wf = fa | (fb << 8);
                        ob &= ~6u;
                        if (fa != fa_in || (ob & 8u)) ob |= 2u;
                        if (fb != fb_in || (ob & 16u)) ob |= 4u;
                        // R7 / R8 (:1235-1348) in reach (see chain.hip.hpp): the general executor does the whole pop
                        const bool nua = !(fa & 1u), nub = !(fb & 1u);
                        if ((nua || nub) && !((nua && (fa & 18u) != 18u) || (nub && (fb & 18u) != 18u))) slowk = 1u;
                    }
                } else {
                    // ---- plain sum (no R3..R6 shape): after R1 only R7 / R8 are left
                    const uint64_t m_nk = __ballot(inC && !(f & 1u) && !(f & 2u));
                    if (!r1_fired && m_nuc && !m_nk) slowk = 1u;
                }
                if (ncand > 64u && !slowk) slowk = 3u;
            }
            if (slowk) { ob = 0; ncand = 0; rv = NONE; }
        }
        CWT(3);        // decisions
        // ---- 4: write marks, read sets and what else the others have to know | barrier | every wavefront works out the whole prefix: lane 8 r + j
        // looks up entries j and j + 8 of rank r's read set -- blocked iff an EARLIER rank writes what the row reads
        const uint32_t par = (lvl++) & 1u;           // (every pass of the loop, also the ones that end in the general executor: the marks taken back at the top are the other table's)
        if (lane < 16u) X_RV[16u * rank + lane] = rv;
        if (lane == 0) {
            X_ROW[rank] = row;
            X_ST[rank] = slowk | (ncand << 8);
            if ((ob & 3u) == 3u && n > 1u) { pm0 = wslot(wv0, par); atomicMin(&wm[pm0], rank); }
            if ((ob & 5u) == 5u && n > 1u) { pm1 = wslot(wv1, par); atomicMin(&wm[pm1], rank); }
        }
        __syncthreads();
        uint32_t c, base, M;
        {
            const uint32_t r_ = lane >> 3, j_ = lane & 7u;
            const uint32_t v0 = X_RV[16u * r_ + j_], v1 = X_RV[16u * r_ + j_ + 8u];
            const uint32_t xs = lane < 8u ? X_ST[lane] : 0u;
            const uint32_t m0 = v0 != NONE ? wm[wslot(v0, par)] : NONE, m1 = v1 != NONE ? wm[wslot(v1, par)] : NONE;
            const uint64_t bm = __ballot(r_ < n && (m0 < r_ || m1 < r_));                 // byte r: rank r is blocked
            const uint64_t sm = __ballot(lane < n && (xs & 0xFFu) != 0u);                  // bit r: rank r is not taken by this loop
            const uint64_t nz = (((bm & 0x7F7F7F7F7F7F7F7Full) + 0x7F7F7F7F7F7F7F7Full) | bm) & 0x8080808080808080ull;
            const uint32_t cb = nz ? (uint32_t)(__ffsll((long long)nz) - 1) >> 3 : n, cs = sm ? (uint32_t)(__ffsll((long long)sm) - 1) : n;
            c = cb < cs ? cb : cs;
            // candidates of the ranks below mine / of the whole prefix: an inclusive scan over lanes 0..7 (DPP row shifts, zero fill)
            int sc = lane < c ? (int)(xs >> 8) : 0;
            sc += __builtin_amdgcn_update_dpp(0, sc, 0x111, 0xF, 0xF, true);
            sc += __builtin_amdgcn_update_dpp(0, sc, 0x112, 0xF, 0xF, true);
            sc += __builtin_amdgcn_update_dpp(0, sc, 0x114, 0xF, 0xF, true);
            M = rdlane((uint32_t)sc, 7);
            base = rank ? rdlane((uint32_t)sc, rank - 1u) : 0u;
        }
        CWT(4);        // marks + check (one barrier)
        if (c == 0u) {
            // ---- the row at the head is not ours
            const uint32_t k0 = sc_(X_ST[0]) & 0xFFu;
            if (k0 == 3u) { why = LV_FAT; break; }      // more candidates than a wavefront has lanes: a level round resolves up to 960
            if (k0 != 1u || gdone >= 256u) { why = LV_DECLINED; big = k0 == 2u ? 1u : 0u; break; }
            // the general executor pops it right here (wavefront 0; the ring has to hold the whole queue for it)
            __syncthreads();
            if (tid < mtop - head) queue[(head + tid) & qmask] = qm[(head + tid) & (ECNE_LV_QM - 1)];
            __syncthreads();
            if (rank == 0u) {
                const uint32_t rr = row;
                const bool sv = J.solved[rr] != 0;
                if (lane == 0) J.inq[rr] = 0;
                wg_fence();
                g_nnz += (J.rpA[rr + 1] - J.rpA[rr]) + (J.rpB[rr + 1] - J.rpB[rr]) + (J.rpC[rr + 1] - J.rpC[rr]);
                uint32_t tl = tail;
                if (!sv) {
                    QState qq;
                    qq.head = head + 1u; qq.tail = tail; qq.evout = nullptr; qq.nev = 0; qq.emit = 0;
                    exec_row(J, qq, rr, g_ht, g_st, g_nu);
                    wg_fence();
                    tl = qq.tail;
                }
                if (lane == 0) { X_CTL[0] = tl; X_CTL[2] = J.ctr->error ? 1u : 0u; }
            }
            __syncthreads();
            head += 1u; tail = sc_(X_CTL[0]); ++gdone;
            const uint32_t err = sc_(X_CTL[2]);
            mtop = head + ((tail - head) < ECNE_LV_QM ? (tail - head) : ECNE_LV_QM);
            if (tid < mtop - head) qm[(head + tid) & (ECNE_LV_QM - 1)] = queue[(head + tid) & qmask];
            __syncthreads();
            if (err) { why = LV_ROUNDS; break; }
            continue;
        }
        // ---- 5: commit the prefix: every wavefront its own pop; the rows of the prefix carry 2 + rank while the pushes are resolved
        if (rank < c) {
            c_pops++; c_nnz += nnz_d;
            if (lane == 0) Q[row] = (uint16_t)(M ? 2u + rank : 0u);      // (no candidates in this round: nobody looks at the tags)
            if (ob & 1u) {
                if (lane == 0) {
                    if (ob & 2u) F[wv0] = (uint8_t)wf;
                    if (ob & 4u) F[wv1] = (uint8_t)(wf >> 8);
                    if (ob & 64u) { if (flip_lds) flipL[row] = (uint8_t)((ob >> 7) & 1u); else flipG[row] = (uint8_t)((ob >> 7) & 1u); }
                }
                if (ob & 32u) {        // make_values (:921-927)
                    if (lane == 0) {
                        st256(J.values + 8ull * wv0, ld256(J.vals + 4ull * validx));
                        st256(J.values + 8ull * wv0 + 4, ld256(J.vals + 4ull * (validx + 1)));
                        J.nvalues[wv0] = 2;
                        J.abz[wv0] = -1;
                        solved[row] = 1;
                    }
                }
                if (ob & 24u) {
                    if ((ob & 8u) && lane < 8u) { uint64_t* const p = (lane < 4u ? J.lb : J.ub) + 4ull * wv0; p[lane & 3u] = lane == 4u ? 1ull : 0ull; }
                    if ((ob & 16u) && lane >= 8u && lane < 16u) { uint64_t* const p = (lane < 12u ? J.lb : J.ub) + 4ull * wv1; p[lane & 3u] = lane == 12u ? 1ull : 0ull; }
                }
                acc += dd;
                if (lane < ncand) cl[base + lane] = (rank << 24) | cv;
            }
        }
        __syncthreads();
        CWT(6);        // prefix + commit (one barrier)
        if (M) {
            // ---- REQUEUE resolution in sequential order (rank, emission index, position in the variable's row list): wavefront 0, as in a level round
            if (rank == 0u) {
                uint32_t new_tail = tail, mt = mtop;
                for (uint32_t b0 = 0; b0 < M; b0 += 64) {
                    const uint32_t jj = b0 + lane;
                    const uint32_t pk = cl[jj < M ? jj : 0u];
                    const uint32_t t = pk & 0xFFFFFFu, rk = pk >> 24;
                    const uint32_t st = Q[t];
                    const bool el = jj < M && (st == 0u || (st >= 2u && st - 2u <= rk));
                    const uint64_t em = __ballot(el);
                    bool dup = false;
                    for (uint64_t mm = em & (em - 1) ? em : 0ull; mm; mm &= mm - 1) {      // (one eligible candidate: nothing to compare)
                        const uint32_t src = (uint32_t)(__ffsll((long long)mm) - 1);
                        const uint32_t ts = rdlane(t, src);
                        if (lane > src && ts == t) dup = true;
                    }
                    const bool win = el && !dup;
                    const uint64_t wmask = __ballot(win);
                    if (win) {
                        // into the mirror while it holds everything queued and has room (256 positions from the new head on), else to the ring
                        const uint32_t pos = new_tail + (uint32_t)__popcll(wmask & lanes_below());
                        if (mt == new_tail && pos - (head + c) < ECNE_LV_QM) qm[pos & (ECNE_LV_QM - 1)] = t; else queue[pos & qmask] = t;
                        Q[t] = 1;
                    }
                    const uint32_t nw = (uint32_t)__popcll(wmask);
                    if (mt == new_tail) { const uint32_t room = ECNE_LV_QM - (new_tail - (head + c)); mt += nw < room ? nw : room; }
                    new_tail += nw;
                }
                if (lane == 0) { X_CTL[0] = new_tail; X_CTL[1] = mt; }
            }
            __syncthreads();
            const uint32_t t_ = X_CTL[0], m_ = X_CTL[1];
            // rows of the prefix that nobody re-queued are out of the queue now (every wavefront its own; the tags are next looked at behind two barriers)
            if (rank < c && lane == 0 && Q[row] >= 2u) Q[row] = 0;
            tail = sc_(t_); mtop = sc_(m_);
        }
        head += c;
        if ((++rounds & 127u) == 0u) fold();
        CWT(7);        // push resolution (one barrier)
    }
    // ---- exit: what is queued goes to the ring; the tables are left as the fast wavefront round expects them (LV_FAT: the level rounds go on with them)
    __syncthreads();
    if (lane == 0) { if (pm0 != NONE) wm[pm0] = NONE; if (pm1 != NONE) wm[pm1] = NONE; }
    if (why != LV_FAT) {
        if (tid < mtop - head) queue[(head + tid) & qmask] = qm[(head + tid) & (ECNE_LV_QM - 1)];
        __syncthreads();
        const uint32_t NS = ECNE_W2_SLOTS(0);
        for (uint32_t i = tid; i < NS; i += ECNE_WG) { tb[i] = 0u; tb[NS + i] = 0xFFFFFFFFu; }
    }
    mtop_io = mtop;
    if (lane == 0) {
        fold();
        C.steps += c_steps; C.nuniq += c_nuniq; C.hits[0] += c_h0; C.hits[1] += c_h1; C.hits[3] += c_h3; C.hits[4] += c_h4;
        my_pops += c_pops; my_nnz += c_nnz;
    }
    if (tid == 0) {
        S.acc[0] += g_st; S.acc[1] += g_nu;
        for (int i = 0; i < 8; ++i) S.acc[2 + i] += g_ht[i];
        S.acc[10] += gdone; S.acc[11] += g_nnz;
    }
    __syncthreads();
    head_io = head; tail_io = tail;
    *n_rounds = rounds; *n_general = gdone; *big_out = big;
    return why;
}

}  // namespace ecne
