// wave2.hip.hpp — the fast wavefront round: a window of up to 64 queue entries, one lane per row, decided from the
// one-line row records and the flag bytes alone, committed in queue order.
// Part of the gfx950 device code of libecne_hip (see kernels.hip.hpp for the overview).
//
// queue_round_wave() (rounds.hip.hpp) walks CSR rows three times per round (no-op test, access sets, execution),
// keeps its REQUEUE candidates and their per-target minima in device memory and costs ~23 us per round on the master
// of a large job -- about a hundred dependent memory round trips. The narrow dependency levels it serves are half of
// the solve time of ecdsa_like(26) for 0.4 % of its pops. This round needs five:
//   1. the window's rows (one coalesced load of the ring);
//   2. per lane: the 64-byte record rec[row], its RowInfo, solved and orientation bytes;
//   3. the flag bytes of the row's variables (all loads in flight together);
//      -> every lane now DECIDES its pop in registers (what the four common shapes do is a function of the flag bytes:
//         products R1, bit checks R2, x == y rows R1/R4/R5 through the bounds-class bits, plain sums R1) without
//         writing anything; written variables are marked in an LDS table, every lane looks its read set up: a lane is
//         blocked iff an EARLIER rank writes something it reads. The prefix below the first blocked rank is exactly
//         what sequential pops would do (decide-then-commit: only true dependencies matter, later writers of what an
//         earlier row reads commit after its decision was taken);
//   4. inline fan-out lists foi[v] of the emitted REQUEUE events and the in_queue tags of their targets;
//      -> candidates (rank, event, position) are numbered by a wave scan; a target is eligible iff it is not queued as of
//         the candidate's rank (not queued at all, or itself popped at a rank <= the candidate's: LDS table of the
//         prefix rows); the lowest eligible candidate per target wins (LDS table), winners are appended in candidate order;
//   5. the stores' completion.
// Anything else -- another shape, more than 15 entries, a bound that is neither [0,1] nor [0,p-1], R7 / R8 in reach,
// an event with more than 3 target rows, errors -- ends the window in front of that row; at rank 0 the round declines
// (nothing touched): 0xFFFFFFFE = pop that one row with the general executor, 0xFFFFFFFF = a long row (> 64 entries) for the
// general round, which runs it on the whole workgroup.
#pragma once
#include "chain.hip.hpp"

namespace ecne {

// LDS tables of the round, at the top of the dynamic LDS (k_solve keeps the state of single-workgroup jobs below them):
// key / value pairs, linear probing, never more than half full, wiped after every round.
// Two sizes: the wavefront round (64 rows) and the workgroup round (512 rows, all eight wavefronts of the master); the
// second one is used when the dynamic LDS has room for its tables (always on the master of a multi-workgroup job, on a
// single-workgroup job when its LDS-resident state leaves 56 KB).
#define ECNE_W2_LOG_MARKS 8    // write marks: variable -> lowest writer rank (<= 64 rows x 2 written variables); x8 in the workgroup round
#define ECNE_W2_LOG_ROWS 7     // rows of the prefix -> rank
#define ECNE_W2_LOG_TGT 9      // push targets -> lowest eligible candidate index
#define ECNE_W2_MAXCAND 256    // candidates one wavefront round resolves (x8 in the workgroup round)
#define ECNE_W2_SLOTS(big) (((1u << ECNE_W2_LOG_MARKS) + (1u << ECNE_W2_LOG_ROWS) + (1u << ECNE_W2_LOG_TGT)) << ((big) ? 3 : 0))
#define ECNE_W2_BYTES (8u * ECNE_W2_SLOTS(0))
#define ECNE_W2_BYTES_BIG (8u * ECNE_W2_SLOTS(1))

struct W2Tab { uint32_t* key; uint32_t* val; uint32_t mask, shift; };
__device__ __forceinline__ void w2_min(const W2Tab& t, uint32_t key, uint32_t v) {          // key != 0
    uint32_t s = (key * 2654435761u) >> t.shift;
    for (;;) {
        const uint32_t k = atomicCAS(&t.key[s], 0u, key);
        if (k == 0u || k == key) { atomicMin(&t.val[s], v); return; }
        s = (s + 1) & t.mask;
    }
}
__device__ __forceinline__ uint32_t w2_get(const W2Tab& t, uint32_t key) {
    uint32_t s = (key * 2654435761u) >> t.shift;
    for (;;) {
        const uint32_t k = t.key[s];
        if (k == key) return t.val[s];
        if (k == 0u) return 0xFFFFFFFFu;
        s = (s + 1) & t.mask;
    }
}
// all threads of the workgroup, once per launch
__device__ __forceinline__ void w2_tables_init(uint32_t w2_off, bool big) {
    uint32_t* const base = (uint32_t*)(ecne_dyn_lds + w2_off);
    const uint32_t nslots = ECNE_W2_SLOTS(big);
    for (uint32_t i = threadIdx.x; i < nslots; i += ECNE_WG) { base[i] = 0u; base[nslots + i] = 0xFFFFFFFFu; }
}

// WG = false: wavefront 0 alone, up to 64 rows. WG = true: ALL threads of the workgroup, up to ECNE_WG rows (rank = thread),
// the wave-level votes and scans become workgroup-level ones through LDS; the return values are uniform across the workgroup.
template <bool LDS, bool WG>
__device__ __noinline__ uint32_t queue_round_fast(const Job& J, ChunkShared& S, uint32_t head, uint32_t tail, uint32_t n, LaneCtr& C,
                                                  uint32_t& my_pops, uint32_t& my_nnz, uint32_t* out_tail, uint32_t* out_examined,
                                                  unsigned long long* why) {
    const int lane = lane_id();
    const uint32_t rank = WG ? (uint32_t)threadIdx.x : (uint32_t)lane;
    const uint32_t NT = WG ? (uint32_t)ECNE_WG : 64u;          // threads taking part
    __shared__ uint32_t s_red[8];                              // workgroup votes: lowest rank with a property (slots 0..3), rank 0's shape (4)
    if (WG) { if (threadIdx.x < 8) s_red[threadIdx.x] = 0xFFFFFFFFu; __syncthreads(); }
    auto sync = [&]() { if constexpr (WG) __syncthreads(); else lds_fence(); };
    // lowest rank for which p holds (0xFFFFFFFF: none); slot: a vote of its own per call site
    auto first_rank = [&](bool p, int slot) -> uint32_t {
        const uint64_t m = __ballot(p);
        if constexpr (!WG) { (void)slot; return m ? (uint32_t)(__ffsll((long long)m) - 1) : 0xFFFFFFFFu; }
        else {
            if (m && lane == 0) atomicMin(&s_red[slot], (uint32_t)(threadIdx.x & ~63u) + (uint32_t)(__ffsll((long long)m) - 1));
            __syncthreads();
            return s_red[slot];
        }
    };
    auto excl_scan = [&](uint32_t x, uint32_t* tot) -> uint32_t {
        if constexpr (WG) return wg_exclusive_scan(x, S.scan, tot); else return wave_excl_scan(x, tot);
    };
    const bool big = WG;
    const uint32_t MAXC = ECNE_W2_MAXCAND << (big ? 3 : 0);
    auto uni = [](const void* p) -> uint64_t {
        const uint64_t x = (uint64_t)p;
        return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(x >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)x);
    };
    const ECNE_GLOBAL u32x4* const rec = (const ECNE_GLOBAL u32x4*)uni(J.rec);
    const ECNE_GLOBAL u32x4* const foi = (const ECNE_GLOBAL u32x4*)uni(J.foi);
    const ECNE_GLOBAL u32x4* const rinfo = (const ECNE_GLOBAL u32x4*)uni(J.rinfo);
    ECNE_GLOBAL uint32_t* const queue = (ECNE_GLOBAL uint32_t*)uni(J.queue);
    const uint32_t qmask = (uint32_t)__builtin_amdgcn_readfirstlane((int)J.qmask);
    ECNE_GLOBAL uint8_t* const solved = (ECNE_GLOBAL uint8_t*)uni(J.solved);
    // flags / in_queue tags / orientation bytes: LDS (single-workgroup job, resident) or device memory
    uint8_t* const Fl = (uint8_t*)(ecne_dyn_lds + (LDS ? J.lds_flags_off : 0u));
    uint16_t* const Ql = (uint16_t*)(ecne_dyn_lds + (LDS ? J.lds_inq_off : 0u));
    ECNE_GLOBAL uint8_t* const Fg = LDS ? (ECNE_GLOBAL uint8_t*)nullptr : (ECNE_GLOBAL uint8_t*)uni(J.flags);
    ECNE_GLOBAL uint16_t* const Qg = LDS ? (ECNE_GLOBAL uint16_t*)nullptr : (ECNE_GLOBAL uint16_t*)uni(J.inq);
    const bool flip_lds = LDS && J.lds_flip_off != 0xFFFFFFFFu;
    uint8_t* const flipL = (uint8_t*)(ecne_dyn_lds + (flip_lds ? J.lds_flip_off : 0u));
    ECNE_GLOBAL uint8_t* const flipG = flip_lds ? (ECNE_GLOBAL uint8_t*)nullptr : (ECNE_GLOBAL uint8_t*)uni(J.flip3);
    auto ldF = [&](uint32_t v) -> uint8_t { if constexpr (LDS) return Fl[v]; else return Fg[v]; };
    auto stF = [&](uint32_t v, uint8_t f) { if constexpr (LDS) Fl[v] = f; else Fg[v] = f; };
    auto ldQ = [&](uint32_t r) -> uint16_t { if constexpr (LDS) return Ql[r]; else return Qg[r]; };
    auto stQ = [&](uint32_t r, uint16_t x) { if constexpr (LDS) Ql[r] = x; else Qg[r] = x; };
    uint32_t* const tb = (uint32_t*)(ecne_dyn_lds + (WG ? J.lds_w2b_off : J.lds_w2_off));      // the two rounds have tables of their own
    const uint32_t LM = ECNE_W2_LOG_MARKS + (big ? 3 : 0), LR = ECNE_W2_LOG_ROWS + (big ? 3 : 0), LT = ECNE_W2_LOG_TGT + (big ? 3 : 0);
    const uint32_t NS = ECNE_W2_SLOTS(big), NMARK = 1u << LM, NROW = 1u << LR;
    const W2Tab Tm = {tb, tb + NS, NMARK - 1, 32 - LM};
    const W2Tab Tr = {tb + NMARK, tb + NS + NMARK, NROW - 1, 32 - LR};
    const W2Tab Tt = {tb + NMARK + NROW, tb + NS + NMARK + NROW, (1u << LT) - 1, 32 - LT};

#ifdef ECNE_W2PROF
    unsigned long long w2t_last = wall_clock64();
#define W2T(k) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); if (rank == 0) { const unsigned long long t_ = wall_clock64(); why[2 + (k)] += t_ - w2t_last; w2t_last = t_; } } while (0)
#else
#define W2T(k) do { } while (0)
#endif
    // ---- 1, 2: my row
    const bool mine = rank < n;
    uint32_t row = 0;
    if (mine) row = queue[(head + rank) & qmask];
    u32x4 w4[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}}, ri4[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    uint8_t is_solved = 0, flip_in = 0;
    if (mine) {
#pragma unroll
        for (int i = 0; i < 4; ++i) w4[i] = rec[4u * row + (uint32_t)i];
        ri4[0] = rinfo[2u * row]; ri4[1] = rinfo[2u * row + 1u];
        is_solved = solved[row];
        flip_in = flip_lds ? flipL[row] : flipG[row];
    }
    W2T(0);        // queue + record + descriptor
    uint32_t w[16];
#pragma unroll
    for (int i = 0; i < 4; ++i) { w[4 * i] = w4[i].x; w[4 * i + 1] = w4[i].y; w[4 * i + 2] = w4[i].z; w[4 * i + 3] = w4[i].w; }
    const uint32_t shape = ri4[0].x, rx = ri4[0].y, kpos = ri4[0].z, kneg = ri4[0].w, k1 = ri4[1].x, k2 = ri4[1].y, validx = ri4[1].z;
    uint32_t nA = w[0] & 0xFFu, nB = (w[0] >> 8) & 0xFFu, nCc = (w[0] >> 16) & 0xFFu, nE = nA + nB + nCc;
    const uint32_t lenC = ri4[1].w;
    const bool xy = (shape & (SH_R5 | SH_R4_T | SH_R4_T2 | SH_R3)) == (SH_R5 | SH_R4_T | SH_R4_T2);
    const bool f1 = (shape & SH_HAS_AB) && !(shape & SH_C_EMPTY);
    const bool f2 = (shape & SH_C_EMPTY) != 0;
    const bool f4 = !(shape & (SH_HAS_AB | SH_C_EMPTY | SH_R3 | SH_R4_T | SH_R4_T2 | SH_R5 | SH_R6));
    const bool r4s = !(shape & (SH_HAS_AB | SH_C_EMPTY | SH_R3 | SH_R5 | SH_R6)) && (((shape & SH_R4_T) != 0) != ((shape & SH_R4_T2) != 0));
    const bool r3x = (shape & (SH_R3 | SH_R4_T | SH_R4_T2 | SH_R5 | SH_R6 | SH_HAS_AB | SH_C_EMPTY)) == (SH_R3 | SH_R4_T | SH_R4_T2 | SH_R5);
    const bool r3f = ((shape & SH_R3) && !(shape & (SH_HAS_AB | SH_C_EMPTY | SH_R4_T | SH_R4_T2 | SH_R5 | SH_R6))) || r3x;
    const bool r6f = (shape & SH_R6) && !(shape & (SH_HAS_AB | SH_C_EMPTY | SH_R3 | SH_R4_T | SH_R4_T2 | SH_R5));
    const bool live = mine && !is_solved;
    // (a row without a record is declined even when it is solved: its pop still counts the row's non-zeros)
    // A long linear row (no record: more than 15 terms -- the 1 025-term sum of a decoder, the 254 bits of a Num2Bits) is
    // re-queued by each of its terms and nearly all of those pops do nothing. Words 1 and 2 of its record line hold two
    // watched variables that say why (long_row_walk, fastrow.hip.hpp): while they still do, the pop is settled here from their
    // two flag bytes, at any rank. Without a pair yet, a plain sum's lane tries the first 8 terms (two of them non-unique, one
    // of those not is_known); anything else waits for rank 0, where the whole wavefront walks the row.
    const bool norec = (w[0] >> 24) == 0;
    const bool lr4 = r4s;                                      // (a binary decomposition; r4s proper needs the record)
    const bool biglin = live && norec && (f4 || lr4) && lenC > 15;
    bool alldone = false, watched = false;
    if (biglin) {
        const uint32_t h0 = w[1], h1 = w[2];
        if (h0 == 0xFFFFFFFEu && f4) { alldone = true; nA = 0; nB = 0; nCc = 0; nE = 0; }        // every term unique (for good): nothing can happen
        else if (h0 < 0xFFFFFFFEu) { watched = true; w[1] = h0; w[2] = h1; nA = 0; nB = 0; nCc = 2; nE = 2; }
        else if (f4) {
            const ECNE_GLOBAL uint32_t* const colC = as_global(J.colC);
            const uint32_t c0 = as_global(J.rpC)[row];
#pragma unroll
            for (uint32_t e = 0; e < 8; ++e) w[1 + e] = colC[c0 + e];
            nA = 0; nB = 0; nCc = 8; nE = 8;
        }
    }
    const bool bigsum = biglin && f4 && !alldone && !watched;      // the first-8-terms test (fast_decide)
    const bool bl_local = biglin && (alldone || watched || f4);    // settled (or declined) by this lane
    bool slow = mine && !bl_local && (norec || (!is_solved && ((shape & SH_BIG) || !(xy || f1 || f2 || f4 || (r4s && !norec) || r3f || r6f))));
    uint32_t reason = slow ? (((w[0] >> 24) == 0 || (shape & SH_BIG)) ? 0u : 1u) : 7u;
    // A row of another shape (a constant x = c, 1 = x + y, ...) all of whose variables are unique and known, with values and
    // bounds as its rules would leave them, is popped without effect (row_is_noop, schedule.hip.hpp: the test the
    // multi-workgroup round uses) -- e.g. the second pop of every constant row, which its own REQUEUE causes.
    bool nop_row = false;
    if (slow && reason == 1u) {
        const RowInfo ri_ = J.rinfo[row];
        bool nb_ = false;
        if (row_is_noop(J, row, ri_, nb_)) { slow = false; reason = 7u; nop_row = true; }
    }
#ifdef ECNE_W2SHAPES
    if (reason == 1) reason = (shape & SH_R3) ? 1u : (shape & SH_R6) ? 2u : (shape & (SH_R4_T | SH_R4_T2)) ? 5u : 3u;
    if (reason == 0) reason = (shape & (SH_R4_T | SH_R4_T2)) ? 4u : 0u;
#endif
    // ---- 3: flag bytes
    const bool walk = live && !slow && !xy && !f2;        // products and plain sums look at every entry
    uint8_t fl[15];
#pragma unroll
    for (uint32_t e = 0; e < 15; ++e) fl[e] = (walk && e < nE) ? ldF(w[1 + e]) : (uint8_t)3;
    uint8_t fa = 3, fb = 3, fx = 3;
    if (live && !slow && (xy || r6f)) { fa = ldF(k1); fb = ldF(k2); }
    if (live && !slow && ((f2 && (shape & SH_R2)) || r3f)) fx = ldF(rx);
    W2T(1);        // flag bytes
    // ---- the decision, in registers (fast_decide)
    FastIn fin;
    fin.shape = shape; fin.rx = rx; fin.kpos = kpos; fin.kneg = kneg; fin.k1 = k1; fin.k2 = k2; fin.nA = nA; fin.nB = nB; fin.nE = nE;
#pragma unroll
    for (int i = 0; i < 16; ++i) fin.w[i] = w[i];
#pragma unroll
    for (int i = 0; i < 15; ++i) fin.fl[i] = fl[i];
    fin.fa = fa; fin.fb = fb; fin.fx = fx; fin.flip_in = flip_in;
    bool wnop = false;                                            // the watched pair still holds: an empty pop
    if (watched) { if (long_watch_holds(fl[0], fl[1], lr4)) wnop = true; else { slow = true; reason = 0; } }
    fin.live = live && !wnop; fin.xy = xy; fin.f2 = f2; fin.f4 = f4; fin.bigsum = bigsum; fin.r4s = r4s && !norec && !nop_row; fin.r3f = r3f && !nop_row; fin.r3x = r3x && !nop_row; fin.r6f = r6f && !nop_row; fin.validx = validx;
    FastOut fo_;
    fo_.slow = slow; fo_.reason = reason;
    fast_decide(J, fin, fo_);
    slow = fo_.slow; reason = fo_.reason;
    uint32_t &wva = fo_.wva, &wvb = fo_.wvb, &nev = fo_.nev;
    uint8_t &wfa = fo_.wfa, &wfb = fo_.wfb, &flip_new = fo_.flip_new;
    bool &wa = fo_.wa, &wb = fo_.wb, &a01 = fo_.a01, &b01 = fo_.b01, &r2 = fo_.r2, &flip_w = fo_.flip_w, &xa_w = fo_.xa_w, &xb_w = fo_.xb_w;
    fp::u256 &xlb0 = fo_.xlb0, &xub0 = fo_.xub0, &xlb1 = fo_.xlb1, &xub1 = fo_.xub1;
    uint32_t* const ev = fo_.ev;
    uint32_t &d_steps = fo_.d_steps, &d_nuniq = fo_.d_nuniq, &d_h0 = fo_.d_h0, &d_h1 = fo_.d_h1, &d_h3 = fo_.d_h3, &d_h4 = fo_.d_h4;
    auto emit = [&](uint32_t v) {
        if (nev == 0) ev[0] = v; else if (nev == 1) ev[1] = v; else if (nev == 2) ev[2] = v; else if (nev == 3) ev[3] = v; else ev[4] = v;
        ++nev;
    };
    if (nop_row && (shape & SH_R4_T) && (shape & SH_R4_T2)) { flip_w = true; flip_new = (uint8_t)(flip_in ^ 1); }     // the empty pop's only effect (:1001-1011)
    // ---- a long row at the head of the window (a decoder's 1 025-term sum, a long product) whose pop the lane could not
    // settle from its first terms: the WHOLE wavefront walks it, lanes across its entries -- 16 strides for 1 025 terms
    // instead of a round of its own on the workgroup. Only what R1 asks is gathered (plus whether R7 / R8 are in reach of a
    // sum): exactly one non-unique variable in C with A and B unique -> it becomes unique; two or more, one of them not
    // is_known -> nothing happens. Rank 0 is never blocked, so no read set is needed; what it writes is marked as usual.
    uint32_t nnz_long = 0;
    if constexpr (!WG) {
        const bool cand0 = mine && !is_solved && (w[0] >> 24) == 0 && slow && (f4 || f1 || long_r4(shape));
        if (rdlane(cand0 ? 1u : 0u, 0)) {
            const uint32_t row0 = rdlane(row, 0), shape0 = rdlane(shape, 0);
            const bool lin0 = !(shape0 & SH_HAS_AB), r40 = long_r4(shape0);
            const uint32_t pivot0 = r40 ? ((shape0 & SH_R4_T) ? rdlane(kpos, 0) : rdlane(kneg, 0)) : 0xFFFFFFFFu;
            LongWalk R;
            long_row_walk<LDS>(J, row0, pivot0, rdlane(biglin ? (f4 ? 1u : 2u) : 0u, 0), R);
            const bool reach78 = lin0 && R.cnt > 0 && !(R.cnt == 1 && !R.nuab) && !R.notknown;      // R7 / R8 could fire: the general executor decides
            const bool r4go = r40 && !R.bad;                                                          // R4's precondition holds: likewise
            if (!reach78 && !r4go) {
                if (lane == 0) {
                    slow = false;
                    nnz_long = R.nnz;
                    if (!R.nuab && R.cnt == 1) {
                        wva = R.u; wfa = (uint8_t)(R.uf | 3u); wa = true;
                        emit(R.u);
                        d_nuniq = 1; d_steps = 1; d_h0 = 1;
                    }
                }
            }
        }
    }
    // ---- the window ends in front of the first row this round does not take
    uint32_t cmax = n;
    {
        if (WG && threadIdx.x == 0) s_red[4] = shape;
        const uint32_t fs_ = first_rank(slow, 0);
        if (fs_ == 0) {                                         // nothing has been touched
            if (rank == 0) why[reason] += 1;
#ifdef ECNE_DECLINELOG
            if (rank == 0) printf("DECL reason %u shape %x nE %u norec %d row %u fa %x fb %x fx %x\n", reason, shape, nE, (int)norec, row, fa, fb, fx);
#endif
            const uint32_t shape0 = WG ? s_red[4] : rdlane(shape, 0);
            return (shape0 & SH_BIG) ? 0xFFFFFFFFu : 0xFFFFFFFEu;
        }
        if (fs_ < cmax) cmax = fs_;
    }
    W2T(2);        // decisions (+ long row scan)
    const bool cand = mine && rank < cmax;
    // ---- write marks, then every lane looks its read set up: blocked iff an earlier rank writes what it reads
    if (cand && live) {
        if (wa) w2_min(Tm, wva + 1u, rank);
        if (wb) w2_min(Tm, wvb + 1u, rank);
    }
    sync();
    bool blocked = false;
    if (cand && live) {
        if (f2) { if (shape & SH_R2) blocked = w2_get(Tm, rx + 1u) < rank; }
        else if (xy) blocked = w2_get(Tm, k1 + 1u) < rank || w2_get(Tm, k2 + 1u) < rank;
        else {
#pragma unroll
            for (uint32_t e = 0; e < 15; ++e)       // (an empty pop and a decomposition row also depend on the bounds of their unique variables)
                if (e < nE && ((fl[e] & 3) != 3 || nop_row || (r4s && !norec) || r3f || r6f) && w2_get(Tm, w[1 + e] + 1u) < rank) blocked = true;
        }
    }
    uint32_t c = cmax;
    const uint32_t fb_ = first_rank(blocked, 1);
    if (fb_ < c) c = fb_;                                                        // >= 1: rank 0 is never blocked
    W2T(3);        // marks + check
    // ---- 4: fan-out of the events of the prefix; an event with more than three target rows ends the prefix in front of it
    u32x4 fo[5] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    uint32_t ncand = 0;
    bool bigev = false;
    if (mine && rank < c && live) {
#pragma unroll
        for (uint32_t k = 0; k < 5; ++k) if (k < nev) fo[k] = foi[ev[k]];
#pragma unroll
        for (uint32_t k = 0; k < 5; ++k) if (k < nev) { if (fo[k].x > 3u) bigev = true; else ncand += fo[k].x; }
    }
    {
        const uint32_t f0 = first_rank(bigev, 2);
        if (f0 == 0) {
            if constexpr (WG) {                                 // rank 0: the general executor takes it (only marks were written)
                for (uint32_t i = rank; i < NMARK; i += NT) { Tm.key[i] = 0u; Tm.val[i] = 0xFFFFFFFFu; }
                sync();
                if (rank == 0) why[6] += 1;
                return 0xFFFFFFFEu;
            } else {
                // rank 0 makes a variable with a long row list unique (a bit feeding dozens of rows): the round is that one
                // pop. Lane 0 commits it, then the wavefront walks the lists of its events in order (REQUEUE as the sequential
                // executor does it, rules_wave.hip.hpp) -- a tenth of what declining the row and popping it through the
                // general executor costs.
                if (lane == 0) {
                    my_pops++;
                    my_nnz += nnz_long ? nnz_long : (biglin ? lenC : nE);
                    if (wa) stF(wva, wfa);
                    if (wb) stF(wvb, wfb);
                    if (a01) { st256(J.lb + 4ull * wva, fp::make(0)); st256(J.ub + 4ull * wva, fp::make(1)); }
                    if (b01) { st256(J.lb + 4ull * wvb, fp::make(0)); st256(J.ub + 4ull * wvb, fp::make(1)); }
                    if (xa_w) { st256(J.lb + 4ull * wva, xlb0); st256(J.ub + 4ull * wva, xub0); }
                    if (xb_w) { st256(J.lb + 4ull * wvb, xlb1); st256(J.ub + 4ull * wvb, xub1); }
                    if (r2) {
                        st256(J.values + 8ull * rx, ld256(J.vals + 4ull * validx));
                        st256(J.values + 8ull * rx + 4, ld256(J.vals + 4ull * (validx + 1)));
                        J.nvalues[rx] = 2;
                        J.abz[rx] = -1;
                        solved[row] = 1;
                    }
                    if (fo_.r3v) { st256(J.values + 8ull * wva, xlb0); J.nvalues[wva] = 1; }
                    if (fo_.v6a) { st256(J.values + 8ull * wva, xub0); st256(J.values + 8ull * wva + 4, xlb0); J.nvalues[wva] = 2; }
                    if (fo_.v6b) { st256(J.values + 8ull * wvb, xub1); st256(J.values + 8ull * wvb + 4, xlb1); J.nvalues[wvb] = 2; }
                    if (flip_w) { if (flip_lds) flipL[row] = flip_new; else flipG[row] = flip_new; }
                    C.steps += d_steps; C.nuniq += d_nuniq;
                    C.hits[0] += d_h0; C.hits[1] += d_h1; C.hits[2] += fo_.d_h2; C.hits[3] += d_h3; C.hits[4] += d_h4; C.hits[5] += fo_.d_h5;
                    stQ(row, (uint16_t)0);                      // (:817) popped: its own events may queue it again
                }
                wg_fence();
                QState qq;
                qq.head = head + 1; qq.tail = tail; qq.evout = nullptr; qq.nev = 0; qq.emit = 0;
                const uint32_t nev0 = rdlane(nev, 0);
#pragma unroll
                for (uint32_t k = 0; k < 5; ++k)
                    if (k < nev0) requeue(J, qq, rdlane(ev[k], 0));
                for (uint32_t i = rank; i < NMARK; i += NT) { Tm.key[i] = 0u; Tm.val[i] = 0xFFFFFFFFu; }
                wg_fence();
                if (rank == 0) why[6] += 1;
                *out_tail = qq.tail;
                *out_examined = 1u | 0x80000000u;      // (not a dependency: see the end of this function)
                return 1;
            }
        }
        if (f0 < c) c = f0;
    }
    if (rank >= c) ncand = 0;
    uint32_t M;
    uint32_t cbase = excl_scan(ncand, &M);
    if (M > MAXC) {       // (many events with full fan-out) keep the ranks whose candidates fit
        const uint32_t f0 = first_rank(rank < c && cbase + ncand > MAXC, 3);       // >= 1: one row has at most 15 candidates
        if (f0 < c) c = f0;
        if (rank >= c) ncand = 0;
        cbase = excl_scan(ncand, &M);
    }
    W2T(4);        // fan-out lists + candidate scan
    const bool in = mine && rank < c;
    // ---- commit the prefix (ranks below c), every lane its own pop
    if (in) {
        my_pops++;
        my_nnz += nnz_long ? nnz_long : (biglin ? lenC : nE);
        w2_min(Tr, row + 1u, rank);
    }
    if (in && live) {
        if (wa) stF(wva, wfa);
        if (wb) stF(wvb, wfb);
        if (a01) { st256(J.lb + 4ull * wva, fp::make(0)); st256(J.ub + 4ull * wva, fp::make(1)); }
        if (b01) { st256(J.lb + 4ull * wvb, fp::make(0)); st256(J.ub + 4ull * wvb, fp::make(1)); }
        if (xa_w) { st256(J.lb + 4ull * wva, xlb0); st256(J.ub + 4ull * wva, xub0); }
        if (xb_w) { st256(J.lb + 4ull * wvb, xlb1); st256(J.ub + 4ull * wvb, xub1); }
        if (r2) {        // make_values (:921-927)
            st256(J.values + 8ull * rx, ld256(J.vals + 4ull * validx));
            st256(J.values + 8ull * rx + 4, ld256(J.vals + 4ull * (validx + 1)));
            J.nvalues[rx] = 2;
            J.abz[rx] = -1;
            solved[row] = 1;
        }
        if (fo_.r3v) { st256(J.values + 8ull * wva, xlb0); J.nvalues[wva] = 1; }
        if (fo_.v6a) { st256(J.values + 8ull * wva, xub0); st256(J.values + 8ull * wva + 4, xlb0); J.nvalues[wva] = 2; }
        if (fo_.v6b) { st256(J.values + 8ull * wvb, xub1); st256(J.values + 8ull * wvb + 4, xlb1); J.nvalues[wvb] = 2; }
        if (flip_w) { if (flip_lds) flipL[row] = flip_new; else flipG[row] = flip_new; }
        C.steps += d_steps; C.nuniq += d_nuniq;
        C.hits[0] += d_h0; C.hits[1] += d_h1; C.hits[2] += fo_.d_h2; C.hits[3] += d_h3; C.hits[4] += d_h4; C.hits[5] += fo_.d_h5;
    }
    sync();
    W2T(5);        // commit
    // ---- REQUEUE resolution in sequential order (rank, emission index, position in the variable's row list)
    uint32_t new_tail = tail;
    if (M) {
        uint32_t tg[15], st[15];
        bool el[15];
#pragma unroll
        for (uint32_t k = 0; k < 5; ++k) {
            const uint32_t nf = (in && live && k < nev) ? fo[k].x : 0u;
            tg[3 * k] = fo[k].y; tg[3 * k + 1] = fo[k].z; tg[3 * k + 2] = fo[k].w;
#pragma unroll
            for (uint32_t p = 0; p < 3; ++p) { el[3 * k + p] = p < nf; st[3 * k + p] = el[3 * k + p] ? (uint32_t)ldQ(tg[3 * k + p]) : 1u; }
        }
        // eligible: not queued at all, or itself a row of the prefix popped at my rank or before
        uint32_t j = cbase;
        uint32_t jj[15];
#pragma unroll
        for (uint32_t i = 0; i < 15; ++i) {
            jj[i] = j;
            if (!el[i]) continue;
            ++j;
            const uint32_t rk = w2_get(Tr, tg[i] + 1u);
            el[i] = rk != 0xFFFFFFFFu ? rk <= rank : st[i] == 0u;
            if (el[i]) w2_min(Tt, tg[i] + 1u, jj[i]);
        }
        sync();
        uint32_t nwin = 0;
#pragma unroll
        for (uint32_t i = 0; i < 15; ++i) {
            if (el[i]) el[i] = w2_get(Tt, tg[i] + 1u) == jj[i];
            nwin += el[i] ? 1u : 0u;
        }
        uint32_t W;
        uint32_t o = tail + excl_scan(nwin, &W);
#pragma unroll
        for (uint32_t i = 0; i < 15; ++i)
            if (el[i]) { queue[o & qmask] = tg[i]; stQ(tg[i], (uint16_t)1); ++o; }
        new_tail = tail + W;
    }
    // rows of the prefix that nobody re-queued are out of the queue now
    if (in && (M == 0 || w2_get(Tt, row + 1u) == 0xFFFFFFFFu)) stQ(row, (uint16_t)0);
    sync();
    W2T(6);        // push resolution
    // ---- leave the tables clean
    for (uint32_t i = rank; i < NS; i += NT) { tb[i] = 0u; tb[NS + i] = 0xFFFFFFFFu; }
    wg_fence();
    if (WG) __syncthreads();
    W2T(7);        // wipe + final fence
    *out_tail = new_tail;
    // rows the round looked at: a prefix shorter than THIS is a dependency (the caller's window adapts to it). A prefix that ends
    // in front of a row or an event the round does not take (general shape, long row list, candidate table full) says nothing
    // about dependencies: reported as examined == committed, with bit 31 set (a wide frontier of such rows is the
    // multi-workgroup round's business, see queue_phase)
    *out_examined = (c < n && c != fb_) ? (c | 0x80000000u) : cmax;
    return c;
}

}  // namespace ecne
