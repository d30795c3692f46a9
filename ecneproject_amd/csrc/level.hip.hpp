// level.hip.hpp — level rounds: the dependency levels of a deep, narrow circuit (Poseidon / MiMC / EdDSA: 2-8 queued rows, hundreds to
// thousands of levels) executed round after round by ONE wavefront without leaving its loop -- one lane per queued row, the
// whole window decided in registers, committed in queue order.
// Part of the gfx950 device code of libecne_hip (see kernels.hip.hpp for the overview).
//
// Why: tests/tools/window_rounds.py replays the oracle's pop trace under "a round takes the first 64 queue entries and commits the
// longest prefix without a true dependency": Poseidon 408 rounds for 1 774 pops, EdDSAMiMCSponge 3 564 for 28 073. The chain
// executor (chain.hip.hpp) pops those rows one by one at 0.65-0.85 us each -- bound by instruction issue on its single
// wavefront, all 64 lanes on ONE row -- and the fast wavefront round (wave2.hip.hpp) pays 8 us of stages plus 2 us of policy
// around every round (stage clocks in LDS mode, tools/gp_w2prof.sh: flag bytes 0.9, decisions 2.0, marks + check 1.1, fan-out
// 0.6, commit 0.7, push resolution 2.1, wipe 0.4 us). A level costs here what ONE pop's dependent chain costs:
//   1. the window's rows from an LDS mirror of the queue ring (no memory trip: the rows were pushed by this loop);
//   2. record + descriptor + solved byte of every row (one trip to the L2, all loads in flight together);
//   3. the flag bytes of the row's variables (ds_read, in flight together) -> fast_decide() in registers;
//   4. write marks in a small hashed LDS table (min rank per slot; a collision only ends the prefix early, never wrongly late),
//      every lane looks its read set up: blocked iff an EARLIER rank writes what it reads;
//   5. inline fan-out lists foi[v] of the events (second trip to the L2), in_queue tags of the targets (LDS; the rows of the
//      prefix carry 2 + rank while they are being popped, rounds.hip.hpp's convention): eligible candidates are compacted in
//      candidate order, the lowest one per target wins (compared in registers), winners go to the mirror;
//   6. flag bytes / tags committed with ds_write; nothing is stored to device memory in a round of products, sums and x == y
//      rows (a store in front of the next level's loads costs its acknowledgement: the memory counter is in order).
// What the loop does not take -- another shape, a row without a record, a bound of the third kind, R7 / R8 in reach, an event with
// more than three rows, errors -- ends the window in front of it; at rank 0 the loop returns and the caller pops that one row with
// the chain executor. Decisions are fast_decide()'s (fastrow.hip.hpp), statement for statement what the fast wavefront round
// commits; the schedule is a prefix of the FIFO order in every round, so the result is the sequential one (DESIGN.md "Schedule").
#pragma once
#include "wave2.hip.hpp"

namespace ecne {

#define ECNE_LV_MARKS 256u      // hashed write marks (u32 rank, 0xFFFFFFFF = none)
#define ECNE_LV_QM 256u         // mirror of the queue ring: positions head .. head + 255
#define ECNE_LV_CAND 960u       // dense list of eligible candidates of one round (64 rows x 15)
static_assert(4u * (ECNE_LV_MARKS + ECNE_LV_QM + ECNE_LV_CAND) <= ECNE_W2_BYTES, "the level rounds' tables live in the fast wavefront round's LDS block");
#ifndef ECNE_LV_WIDE_AVAIL
#define ECNE_LV_WIDE_AVAIL 192u   // more rows than this queued: a wide frontier, the round schedule's business (unless it asks for level rounds: wide_ok)
#endif

#ifndef ECNE_LVG_WIDE_AVAIL
#define ECNE_LVG_WIDE_AVAIL 128u  // ... on the master of a team: its solo drain rounds take 512 rows per level and win from about 250 queued rows on
#endif                            //     (45 chains side by side, ~360 rows queued: 186 ms with level rounds up to 192 rows, 168 with solo drains)
#ifndef ECNE_SOLO_RATIO
#define ECNE_SOLO_RATIO 8       // (rounds.hip.hpp: solo drains when a narrow round committed less than 1 / ECNE_SOLO_RATIO of a well-filled window)
#endif
__device__ __forceinline__ bool level_rounds_on(const Job& J) { return (J.lv_off & 1u) == 0; }
// LV_FAT: more rows queued than a crew round takes (crew.hip.hpp) -- level rounds until LV_NARROW: the frontier is narrow again
// (LV_FAT / LV_NARROW leave the LDS tables and the queue mirror in place for the other loop -- `warm`, mtop_io; LV_NARROW_COLD: narrow again
//  behind a pop of the general executor, tables restored)
enum : uint32_t { LV_EMPTY = 0, LV_DECLINED = 1, LV_WIDE = 2, LV_ROUNDS = 3, LV_REFILL = 4, LV_CUT = 5, LV_FAT = 6, LV_NARROW = 7, LV_NARROW_COLD = 8 };
#ifndef ECNE_CREW_MAX
#define ECNE_CREW_MAX 8u          // rows per crew round = wavefronts of the workgroup
#endif
#ifndef ECNE_CREW_ENTER
#define ECNE_CREW_ENTER 4u        // level rounds hand over to crew rounds when at most this many rows are queued (a frontier that hovers around eight rows would change loops every round)
#endif

// Wavefront 0 of a single-workgroup job whose flags / in_queue tags are LDS-resident (chain_ok). head / tail: the queue cursors, in
// and out. Returns why it stopped (LV_*); *n_rounds = rounds run.
// LDS = false: the master of a multi-workgroup job (flags / tags in device memory; level rounds in place of its fast wavefront rounds):
// tags are read before anything is stored in a round (a prefix row's rank comes from the window in registers, not from its tag), a
// round resolves at most 64 candidates, stores are fenced at the end of every round; cut_exit: a well-filled window whose prefix a
// dependency cut to an eighth returns LV_CUT (chains side by side: the caller's solo drain rounds take those).
// the fast wavefront round's LDS block as that round expects it (wavefront 0; after a level_rounds call that ended LV_DECLINED under `semi`)
__device__ __forceinline__ void lv_tables_restore(const Job& J) {
    uint32_t* const tb = (uint32_t*)(ecne_dyn_lds + J.lds_w2_off);
    const uint32_t NS = ECNE_W2_SLOTS(0);
    for (uint32_t i = (uint32_t)lane_id(); i < NS; i += 64) { tb[i] = 0u; tb[NS + i] = 0xFFFFFFFFu; }
    wg_fence();
}
template <bool LDS>
__device__ __noinline__ uint32_t level_rounds(const Job& J, uint32_t& head_io, uint32_t& tail_io, uint32_t max_rounds, bool wide_ok, bool cut_exit, LaneCtr& C,
                                              uint32_t& my_pops, uint32_t& my_nnz, uint32_t* n_rounds, unsigned long long* prof, bool narrow_exit = false,
                                              bool warm = false, uint32_t* mtop_io = nullptr, bool semi = false, bool semi_in = false) {
    const uint32_t lane = (uint32_t)lane_id();
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
    uint8_t* const Fl = (uint8_t*)(ecne_dyn_lds + (LDS ? J.lds_flags_off : 0u));
    uint16_t* const Ql = (uint16_t*)(ecne_dyn_lds + (LDS ? J.lds_inq_off : 0u));
    ECNE_GLOBAL uint8_t* const Fg = LDS ? (ECNE_GLOBAL uint8_t*)nullptr : (ECNE_GLOBAL uint8_t*)uni(J.flags);
    ECNE_GLOBAL uint16_t* const Qg = LDS ? (ECNE_GLOBAL uint16_t*)nullptr : (ECNE_GLOBAL uint16_t*)uni(J.inq);
    auto ldF = [&](uint32_t v) -> uint8_t { if constexpr (LDS) return Fl[v]; else return Fg[v]; };
    auto stF = [&](uint32_t v, uint8_t f) { if constexpr (LDS) Fl[v] = f; else Fg[v] = f; };
    auto ldQ = [&](uint32_t r) -> uint32_t { if constexpr (LDS) return Ql[r]; else return Qg[r]; };
    auto stQ = [&](uint32_t r, uint32_t x) { if constexpr (LDS) Ql[r] = (uint16_t)x; else Qg[r] = (uint16_t)x; };
    const bool flip_lds = LDS && J.lds_flip_off != 0xFFFFFFFFu;
    uint8_t* const flipL = (uint8_t*)(ecne_dyn_lds + (flip_lds ? J.lds_flip_off : 0u));
    ECNE_GLOBAL uint8_t* const flipG = flip_lds ? (ECNE_GLOBAL uint8_t*)nullptr : (ECNE_GLOBAL uint8_t*)uni(J.flip3);
    uint32_t* const tb = (uint32_t*)(ecne_dyn_lds + J.lds_w2_off);
    uint32_t* const wm = tb;                                   // write marks
    uint32_t* const qm = tb + ECNE_LV_MARKS;                   // queue mirror
    uint32_t* const cl = tb + ECNE_LV_MARKS + ECNE_LV_QM;      // candidate list
    auto wslot = [](uint32_t v) -> uint32_t { return (v * 2654435761u) >> 24; };

    uint32_t head = head_io, tail = tail_io, rounds = 0, why = LV_EMPTY;
    // (counters in registers: the by-reference ones live in the caller's frame -- a scratch round trip per increment)
    uint32_t c_steps = 0, c_nuniq = 0, c_h0 = 0, c_h1 = 0, c_h3 = 0, c_h4 = 0, c_pops = 0, c_nnz = 0;
#ifdef ECNE_LVPROF
    unsigned long long lvt_last = wall_clock64();
#define LVT(k) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); if (lane == 0) { const unsigned long long t_ = wall_clock64(); prof[k] += t_ - lvt_last; lvt_last = t_; } } while (0)
#define LVCOUNT(k) do { if (lane == 0) prof[k] += 1; } while (0)
#else
#define LVT(k) do { } while (0)
#define LVCOUNT(k) do { } while (0)
#endif
    // ---- entry: the tables of the fast wavefront round become ours (clean on entry, restored on exit); the mirror is filled from the ring
    // (warm: the crew rounds left marks and mirror as this loop keeps them)
    // positions head .. mtop - 1 of the queue are mirrored in LDS (at most 256); what is queued behind them lives in the ring only. A
    // push lands in the mirror while the mirror holds everything that is queued, else in the ring (wide_ok: a long queue is worked
    // off 256 positions at a time -- the loop returns LV_REFILL when the mirrored part is used up and is entered again).
    uint32_t mtop;
    // (semi_in: the call before this one ended in front of a row it does not take and left the tables as they were -- lv_tables_restore -- the
    //  marks are all taken back, only the mirror has to be filled again)
    if (warm) mtop = *mtop_io;
    else {
        if (!semi_in) for (uint32_t i = lane; i < ECNE_LV_MARKS; i += 64) wm[i] = 0xFFFFFFFFu;
        wg_fence();
        mtop = head + ((tail - head) < ECNE_LV_QM ? (tail - head) : ECNE_LV_QM);
        for (uint32_t i = lane; i < mtop - head; i += 64) qm[(head + i) & (ECNE_LV_QM - 1)] = queue[(head + i) & qmask];
        lds_fence();
    }
    LVT(0);        // entry
    const ECNE_GLOBAL uint32_t* const fo_rows = (const ECNE_GLOBAL uint32_t*)uni(J.fo_rows);
    // exclusive prefix sum of a small per-lane count (< 2^B) over the wavefront by bit planes: ballots and v_mbcnt, no cross-lane moves
    auto scan_bits = [&](uint32_t x, int B, uint32_t* total) -> uint32_t {
        uint32_t off = 0, tot = 0;
        for (int b = 0; b < B; ++b) {
            const uint64_t m = __ballot((x >> b) & 1u);
            off += (uint32_t)__popcll(m & lanes_below()) << b;
            tot += (uint32_t)__popcll(m) << b;
        }
        *total = tot;
        return off;
    };
    // largest value (< 16) over the lanes for which `on` holds
    auto max4 = [&](uint32_t x, bool on) -> uint32_t {
        uint64_t mask = __ballot(on);
        uint32_t r = 0;
        for (int b = 3; b >= 0; --b) {
            const uint64_t m = __ballot(on && ((x >> b) & 1u)) & mask;
            if (m) { r |= 1u << b; mask = m; }
        }
        return r;
    };
    while (head != tail) {
        const uint32_t avail = tail - head;
        if (!wide_ok && avail > (LDS ? ECNE_LV_WIDE_AVAIL : ECNE_LVG_WIDE_AVAIL)) { why = LV_WIDE; break; }      // a wide frontier: the rounds on the whole workgroup first
        if (rounds >= max_rounds) { why = LV_ROUNDS; break; }
        if (narrow_exit && rounds && avail <= ECNE_CREW_ENTER) { why = LV_NARROW; break; }      // a narrow frontier again: crew rounds (crew.hip.hpp)
        uint32_t n = avail < 64u ? avail : 64u;
        if (head + n > mtop) {                       // the window reaches beyond the mirror
            if (mtop - head < 16u && mtop != tail) { why = LV_REFILL; break; }
            n = mtop - head;
        }
        const uint32_t rank = lane;
        const bool mine = rank < n;
        // ---- 1, 2: my row
        uint32_t row = mine ? qm[(head + rank) & (ECNE_LV_QM - 1)] : 0u;
        u32x4 w4[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}}, ri4[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
        uint8_t is_solved = 0, flip_in = 0;
        if (mine) {
#pragma unroll
            for (int i = 0; i < 4; ++i) w4[i] = rec[4u * row + (uint32_t)i];
            ri4[0] = rinfo[2u * row]; ri4[1] = rinfo[2u * row + 1u];
            is_solved = solved[row];
            flip_in = flip_lds ? flipL[row] : flipG[row];
        }
        LVT(1);        // record + descriptor
        uint32_t w[16];
#pragma unroll
        for (int i = 0; i < 4; ++i) { w[4 * i] = w4[i].x; w[4 * i + 1] = w4[i].y; w[4 * i + 2] = w4[i].z; w[4 * i + 3] = w4[i].w; }
        const uint32_t shape = ri4[0].x, rx = ri4[0].y, kpos = ri4[0].z, kneg = ri4[0].w, k1 = ri4[1].x, k2 = ri4[1].y, validx = ri4[1].z;
        const uint32_t nA = w[0] & 0xFFu, nB = (w[0] >> 8) & 0xFFu, nCc = (w[0] >> 16) & 0xFFu, nE = nA + nB + nCc;
        const bool norec = (w[0] >> 24) == 0;
        const bool xy = (shape & (SH_R5 | SH_R4_T | SH_R4_T2 | SH_R3)) == (SH_R5 | SH_R4_T | SH_R4_T2);
        const bool f1 = (shape & SH_HAS_AB) && !(shape & SH_C_EMPTY);
        const bool f2 = (shape & SH_C_EMPTY) != 0;
        const bool f4 = !(shape & (SH_HAS_AB | SH_C_EMPTY | SH_R3 | SH_R4_T | SH_R4_T2 | SH_R5 | SH_R6));
        const bool live = mine && !is_solved;
        // A long linear row (no record: the 1 025-term sum of a decoder, the 88 bits of a decomposition) is re-queued by each of its
        // terms and nearly all of those pops do nothing; words 1 and 2 of its record line hold the watched pair that says why
        // (long_row_walk, fastrow.hip.hpp). While the pair still does, the pop is settled here from two flag bytes, at any rank.
        const uint32_t lenC = ri4[1].w;
        const bool lr4 = long_r4(shape);
        const bool biglin = live && norec && (f4 || lr4) && lenC > 15;
        bool bl_nop = false, watched = false;
        uint32_t wv0 = w[1], wv1 = w[2];
        if (__ballot(biglin)) {
            const uint32_t h0 = w[1], h1 = w[2];
            watched = biglin && h0 < 0xFFFFFFFEu;
            const uint8_t g0 = ldF(watched ? h0 : 0u), g1 = ldF(watched ? h1 : 0u);
            bl_nop = (biglin && h0 == 0xFFFFFFFEu && f4) || (watched && long_watch_holds(g0, g1, lr4));
            watched = watched && bl_nop;
            // (round 5) ... and a decomposition whose pivot and lowest bit are not unique, the pivot's bounds cut already (long_r4_idle):
            // that pop has read those two
            if (biglin && lr4 && !bl_nop && long_r4_idle(J, shape, ldF(kpos), ldF(kneg), kpos, kneg, lenC)) {
                bl_nop = watched = true; wv0 = kpos; wv1 = kneg;
            }
        }
        bool slow = mine && !bl_nop && (norec || (shape & SH_BIG) || (!is_solved && !(xy || f1 || f2 || f4)));
        // ---- 3: products and plain sums walk their entries (R1, :827-873): flag bytes from LDS, counted on the fly. The loop is
        // unrolled over the record's 15 slots and ends, for the whole wavefront, at the longest row of the window.
        const bool walk = live && !slow && !xy && !f2;
        const uint32_t maxE = max4(nE, walk);
        // (written with masks and arithmetic, not with branches: the compiler turns a chain of small `if`s on per-lane conditions into
        //  exec-mask bookkeeping on the scalar unit, ~40 instructions and four branches per entry; this is ~20 straight-line VALU ones)
        const uint32_t nEw = walk ? nE : 0u, nAB = nA + nB;
        uint32_t cnt = 0, u = 0, uf32 = 0, nonfinal = 0, nuab32 = 0, notk32 = 0;
#pragma unroll
        for (uint32_t g = 0; g < 4; ++g) {          // four entries per step: their flag bytes are in flight together
            if (4 * g >= maxE) break;
            uint32_t fg[4];
#pragma unroll
            for (uint32_t t = 0; t < 4; ++t) {
                const uint32_t e = 4 * g + t;
                if (e >= 15) { fg[t] = 3; continue; }
                const uint32_t on = (uint32_t)((int32_t)(e - nEw) >> 31);            // all ones iff e < nEw
                fg[t] = ldF(w[1 + e] & on);
            }
#pragma unroll
            for (uint32_t t = 0; t < 4; ++t) {
                const uint32_t e = 4 * g + t;
                if (e >= 15) continue;
                const uint32_t f = fg[t];
                const uint32_t on = (uint32_t)((int32_t)(e - nEw) >> 31);
                const uint32_t inC = ~(uint32_t)((int32_t)(e - nAB) >> 31);           // all ones iff e >= nA + nB
                nonfinal |= (((((f & 3u) ^ 3u) + 3u) >> 2) & on & 1u) << e;
                const uint32_t nu = ~f & on & 1u;                                     // not unique
                nuab32 |= nu & ~inC;
                const uint32_t cn = nu & inC;
                const uint32_t m = 0u - (cn & ((cnt - 1u) >> 31));                    // the first non-unique variable of C
                u = (u & ~m) | (w[1 + e] & m);
                uf32 = (uf32 & ~m) | (f & m);
                cnt += cn;
                notk32 |= cn & ((f >> 1) ^ 1u);
            }
        }
        const bool nuab = nuab32 != 0, notknown = (notk32 & 1u) != 0;
        const uint8_t uf = (uint8_t)uf32;
        FastOut D;
        D.slow = slow; D.reason = 7;
        if (walk) {
            if (!nuab && cnt == 1) {
                D.wva = u; D.wfa = (uint8_t)(uf | 3); D.wa = true;
                D.ev[0] = u; D.nev = 1;
                D.d_nuniq = 1; D.d_steps = 1; D.d_h0 = 1;
            } else if (f4 && cnt > 0 && !notknown) { D.slow = true; D.reason = 5; }       // R7 / R8 in reach (:1235-1348): the general executor decides
        }
        LVT(2);        // flag bytes (+ the walk's decision)
        // ---- bit checks and x == y rows: fast_decide (fastrow.hip.hpp) on their two or three flag bytes -- only when the window holds one
        const bool oth = live && !slow && (xy || f2);
        if (__ballot(oth)) {
            uint8_t fa = 3, fb = 3, fx = 3;
            const bool on_xy = oth && xy, on_x = oth && f2 && (shape & SH_R2);
            const uint8_t a_ = ldF(on_xy ? k1 : 0u), b_ = ldF(on_xy ? k2 : 0u), x_ = ldF(on_x ? rx : 0u);
            if (on_xy) { fa = a_; fb = b_; }
            if (on_x) fx = x_;
            FastIn fin;
            fin.shape = shape; fin.rx = rx; fin.kpos = kpos; fin.kneg = kneg; fin.k1 = k1; fin.k2 = k2; fin.nA = nA; fin.nB = nB; fin.nE = nE;
#pragma unroll
            for (int i = 0; i < 16; ++i) fin.w[i] = w[i];
#pragma unroll
            for (int i = 0; i < 15; ++i) fin.fl[i] = 3;
            fin.fa = fa; fin.fb = fb; fin.fx = fx; fin.flip_in = flip_in;
            fin.live = oth; fin.xy = xy; fin.f2 = f2; fin.f4 = false; fin.bigsum = false; fin.r4s = false; fin.r3f = false; fin.r3x = false; fin.r6f = false; fin.validx = validx;
            FastOut D2;
            D2.slow = false; D2.reason = 7;
            // (x == y rows with a bound of the third kind go through the limbs in fast_decide: not here -- the chain executor's general path)
            if (on_xy && (((fa | fb) & 8u) || k1 == k2 || nE != 2)) { D2.slow = true; D2.reason = 3; }
            fast_decide(J, fin, D2);
            if (oth) D = D2;
        }
        slow = D.slow;
        // ---- the window ends in front of the first row this loop does not take
        uint32_t cmax = n;
        {
            const uint64_t m = __ballot(slow);
            if (m) {
                const uint32_t fs = (uint32_t)(__ffsll((long long)m) - 1);
                if (fs == 0) { why = LV_DECLINED; LVCOUNT(8 + (rdlane(D.reason, 0) & 7u)); break; }
                cmax = fs;
            }
        }
        LVT(3);        // decisions
        const bool cand = mine && rank < cmax;
        uint32_t c = cmax;
        // ---- 4: write marks; blocked iff an earlier rank writes what I read (a window of one row has nothing to check)
        if (cmax > 1) {
            const uint32_t sa = wslot(D.wva), sb = wslot(D.wvb);
            const bool ma = cand && live && D.wa, mb = cand && live && D.wb;
            if (ma) atomicMin(&wm[sa], rank);
            if (mb) atomicMin(&wm[sb], rank);
            bool blocked = false;
            if (__ballot(cand && oth)) {
                const bool c2 = cand && oth && f2 && (shape & SH_R2), cxy = cand && oth && xy;
                const uint32_t m0 = wm[wslot((c2 || cxy) ? (cxy ? k1 : rx) : 0u)], m1 = wm[wslot(cxy ? k2 : 0u)];
                if (c2) blocked = m0 < rank;
                if (cxy) blocked = m0 < rank || m1 < rank;
            }
            const uint32_t cw_ = (cand && walk) ? nonfinal : 0u;
            uint32_t blk = 0;
#pragma unroll
            for (uint32_t g = 0; g < 4; ++g) {
                if (4 * g >= maxE) break;
                uint32_t mg[4];
#pragma unroll
                for (uint32_t t = 0; t < 4; ++t) {
                    const uint32_t e = 4 * g + t;
                    if (e >= 15) { mg[t] = 0xFFFFFFFFu; continue; }
                    mg[t] = wm[wslot(w[1 + e] & (0u - ((cw_ >> e) & 1u)))];
                }
#pragma unroll
                for (uint32_t t = 0; t < 4; ++t) { const uint32_t e = 4 * g + t; if (e < 15) blk |= (mg[t] < rank ? 1u : 0u) & (cw_ >> e); }
            }
            if (blk & 1u) blocked = true;
            if (__ballot(cand && watched)) {       // the watched pair of a long row is what its empty pop has read
                const bool on = cand && watched;
                const uint32_t m0 = wm[wslot(on ? wv0 : 0u)], m1 = wm[wslot(on ? wv1 : 0u)];
                if (on && (m0 < rank || m1 < rank)) blocked = true;
            }
            const uint64_t m = __ballot(blocked);
            if (m) { const uint32_t fb_ = (uint32_t)(__ffsll((long long)m) - 1); if (fb_ < c) c = fb_; }      // >= 1: rank 0 is never blocked
            if (ma) wm[sa] = 0xFFFFFFFFu;       // (marks are the round's: the writers take them back)
            if (mb) wm[sb] = 0xFFFFFFFFu;
        }
        LVT(4);        // marks + check
        // ---- 5: fan-out of the events of the prefix: inline lists (up to three rows) arrive with foi[v], longer ones are copied from fo_rows
        u32x4 fo[5] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
        uint32_t ncand = 0;
        bool bigev = false;
        const uint32_t nev = (mine && rank < c && live) ? D.nev : 0u;
        const uint32_t maxev = max4(nev, true);
#pragma unroll
        for (uint32_t k = 0; k < 5; ++k) { if (k >= maxev) break; if (k < nev) fo[k] = foi[D.ev[k]]; }
#pragma unroll
        for (uint32_t k = 0; k < 5; ++k) { if (k >= maxev) break; if (k < nev) { ncand += fo[k].x; if (fo[k].x > 3u) bigev = true; } }
        LVT(5);        // fan-out lists
        // candidate numbering in (rank, emission, position) order; the mirror holds 256 queue positions, so the prefix ends where its
        // pushes (at most: every candidate wins) would not fit behind what stays queued
        uint32_t M = 0, cb = 0;
        const bool anybig = __ballot(bigev) != 0;
        if (maxev) {
            cb = anybig ? wave_excl_scan(ncand, &M) : scan_bits(ncand, 4, &M);
            if (M > ECNE_LV_CAND) {                  // the candidate list holds 960 entries: the prefix ends where they would not fit
                const uint64_t m = __ballot(rank < c && cb + ncand > ECNE_LV_CAND);
                if (m) {
                    const uint32_t f0 = (uint32_t)(__ffsll((long long)m) - 1);
                    if (f0 == 0) { why = LV_DECLINED; LVCOUNT(14); break; }          // (one row's lists alone: the chain executor walks them)
                    if (f0 < c) c = f0;
                }
                if (rank >= c) ncand = 0;
                cb = wave_excl_scan(ncand, &M);
            }
        }
        if constexpr (!LDS) {
            if (M > 64u) {                           // device-memory tags: one block of 64 candidates per round (all tags are read before any is stored)
                const uint64_t m = __ballot(rank < c && cb + ncand > 64u);
                if (m) {
                    const uint32_t f0 = (uint32_t)(__ffsll((long long)m) - 1);
                    if (f0 == 0) { why = LV_DECLINED; LVCOUNT(14); break; }
                    if (f0 < c) c = f0;
                }
                if (rank >= c) ncand = 0;
                cb = wave_excl_scan(ncand, &M);
            }
            // chains side by side: a well-filled window cut to an eighth by a dependency -- the caller's solo drain rounds take those
            if (cut_exit && n >= 32u && ECNE_SOLO_RATIO * c <= n && c < cmax) { why = LV_CUT; break; }
        }
        const bool in = mine && rank < c;
        const bool inl = in && live;
        // ---- 6: commit the prefix: every lane its own pop; the rows of the prefix carry 2 + rank while the pushes are resolved
        if (in) {
            c_pops++;
            c_nnz += bl_nop ? lenC : nE;
            if constexpr (LDS) stQ(row, 2u + rank);
        }
        if (inl) {
            if (D.wa) stF(D.wva, D.wfa);
            if (D.wb) stF(D.wvb, D.wfb);
            if (D.a01) { st256(J.lb + 4ull * D.wva, fp::make(0)); st256(J.ub + 4ull * D.wva, fp::make(1)); }
            if (D.b01) { st256(J.lb + 4ull * D.wvb, fp::make(0)); st256(J.ub + 4ull * D.wvb, fp::make(1)); }
            if (D.r2) {        // make_values (:921-927)
                st256(J.values + 8ull * rx, ld256(J.vals + 4ull * validx));
                st256(J.values + 8ull * rx + 4, ld256(J.vals + 4ull * (validx + 1)));
                J.nvalues[rx] = 2;
                J.abz[rx] = -1;
                solved[row] = 1;
            }
            if (D.flip_w) { if (flip_lds) flipL[row] = D.flip_new; else flipG[row] = D.flip_new; }
            c_steps += D.d_steps; c_nuniq += D.d_nuniq;
            c_h0 += D.d_h0; c_h1 += D.d_h1; c_h3 += D.d_h3; c_h4 += D.d_h4;
        }
        LVT(6);        // commit
        // ---- REQUEUE resolution in sequential order (rank, emission index, position in the variable's row list)
        uint32_t new_tail = tail;
        bool requeued = false;
        if (M) {
            // the candidate list: rank << 24 | target row, in candidate order
            uint32_t base = cb;
#pragma unroll
            for (uint32_t k = 0; k < 5; ++k) {
                if (k >= maxev) break;
                const uint32_t nf = (inl && k < nev) ? fo[k].x : 0u;
                if (nf >= 1 && nf <= 3) {
                    cl[base] = (rank << 24) | fo[k].y;
                    if (nf >= 2) cl[base + 1] = (rank << 24) | fo[k].z;
                    if (nf >= 3) cl[base + 2] = (rank << 24) | fo[k].w;
                }
                if (anybig) {      // a long list: the wavefront copies it, lanes across its positions
                    for (uint64_t bm = __ballot(nf > 3u); bm; bm &= bm - 1) {
                        const uint32_t src = (uint32_t)(__ffsll((long long)bm) - 1);
                        const uint32_t n_ = rdlane(nf, src), off = rdlane(fo[k].y, src), b_ = rdlane(base, src);
                        for (uint32_t p = lane; p < n_; p += 64) cl[b_ + p] = (src << 24) | fo_rows[off + p];
                    }
                }
                base += nf;
            }
            // eligible: not queued, or itself a row of the prefix popped at the candidate's rank or before; the lowest eligible
            // candidate per target wins -- 64 candidates at a time, each compared with the eligible ones below it (registers); a
            // winner's tag is set at once, so the later blocks see it queued
            for (uint32_t b0 = 0; b0 < M; b0 += 64) {
                const uint32_t jj = b0 + lane;
                const uint32_t pk = cl[jj < M ? jj : 0u];
                const uint32_t t = pk & 0xFFFFFFu, rk = pk >> 24;
                const uint32_t st = ldQ(t);
                bool el;
                if constexpr (LDS) el = jj < M && (st == 0u || (st >= 2u && st - 2u <= rk));
                else {
                    uint32_t prk = 0xFFFFFFFFu;                       // the target's own rank if it is a row of the prefix (registers: the window's rows)
                    for (uint32_t k = 0; k < c; ++k) { const uint32_t rowk = rdlane(row, k); if (rowk == t) prk = k; }
                    el = jj < M && (prk != 0xFFFFFFFFu ? prk <= rk : st == 0u);
                }
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
                    if (mtop == new_tail && pos - (head + c) < ECNE_LV_QM) {
                        qm[pos & (ECNE_LV_QM - 1)] = t;
                        // (the master of a team: the ring keeps every queue position that was ever pushed -- the positions popped since the last
                        //  P3 pass are that pass's work list, p3p4_incremental in k_solve.hip.hpp; the round stores to device memory anyway)
                        if constexpr (!LDS) queue[pos & qmask] = t;
                    } else queue[pos & qmask] = t;
                    stQ(t, 1u);
                }
                if constexpr (!LDS) {                 // (one block) a row of the prefix that a winner re-queued keeps its tag
                    for (uint64_t mm = wmask; mm; mm &= mm - 1) { const uint32_t tw = rdlane(t, (uint32_t)(__ffsll((long long)mm) - 1)); if (in && tw == row) requeued = true; }
                }
                {
                    const uint32_t nw = (uint32_t)__popcll(wmask);
                    if (mtop == new_tail) { const uint32_t room = ECNE_LV_QM - (new_tail - (head + c)); mtop += nw < room ? nw : room; }
                    new_tail += nw;
                }
            }
        }
        // rows of the prefix that nobody re-queued are out of the queue now
        if constexpr (LDS) { if (in && ldQ(row) >= 2u) stQ(row, 0u); }
        else {
            if (in && !requeued) stQ(row, 0u);
            wg_fence();                               // the round's stores have landed before the next round's loads
            if ((rounds & 63u) == 63u) job_heartbeat(J);
        }
        head += c;
        tail = new_tail;
        ++rounds;
        LVT(7);        // push resolution
    }
    // ---- exit: what is queued goes to the ring; the tables are left as the fast wavefront round expects them (LV_NARROW: the crew rounds go on with them)
    lds_fence();
    if (why == LV_NARROW) { *mtop_io = mtop; wg_fence(); }
    else {
        for (uint32_t i = lane; i < mtop - head; i += 64) queue[(head + i) & qmask] = qm[(head + i) & (ECNE_LV_QM - 1)];
        lds_fence();
        // (LV_DECLINED with `semi`: the caller pops the one row and comes right back -- the 1 792 words of the tables are restored by
        //  whoever leaves for good, lv_tables_restore)
        if (!(semi && why == LV_DECLINED)) {
            const uint32_t NS = ECNE_W2_SLOTS(0);
            for (uint32_t i = lane; i < NS; i += 64) { tb[i] = 0u; tb[NS + i] = 0xFFFFFFFFu; }
        }
        wg_fence();
    }
    LVT(15);       // exit
    C.steps += c_steps; C.nuniq += c_nuniq; C.hits[0] += c_h0; C.hits[1] += c_h1; C.hits[3] += c_h3; C.hits[4] += c_h4;
    my_pops += c_pops; my_nnz += c_nnz;
    head_io = head; tail_io = tail;
    *n_rounds = rounds;
    return why;
}

}  // namespace ecne
