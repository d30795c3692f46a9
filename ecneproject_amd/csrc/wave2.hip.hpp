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

// ---- what one pop of a row of the four common shapes does, decided in registers from the row record and the flag bytes
// (x == y rows with a bound of the third kind: from the limbs). Nothing is written: the caller commits FastOut for the rows
// that make it into the prefix. Shared by the fast wavefront / workgroup round and the multi-workgroup round.
struct FastIn {
    uint32_t shape, rx, kpos, kneg, k1, k2, nA, nB, nE;
    uint32_t w[16];                 // row record: w[1 + e] = variable of entry e (A, B, C)
    uint8_t fl[15], fa, fb, fx;     // flag bytes: of the entries (products, sums), of k1 / k2 (x == y), of x (bit check)
    uint8_t flip_in;
    bool live, xy, f2, f4, bigsum;
    bool r4s = false;               // a short binary-decomposition row (R4 shape, l > 2): taken while R4's precondition fails
    bool r3f = false;               // a constant row x = c (R3 shape and none of R4..R6's)
    uint32_t validx = 0;            // the row's constants in J.vals (R3: c)
};
struct FastOut {
    uint32_t wva = 0, wvb = 0;      // variables whose flag byte (and maybe bounds) this pop changes
    uint8_t wfa = 0, wfb = 0;
    bool wa = false, wb = false, a01 = false, b01 = false, r2 = false, flip_w = false;
    bool xa_w = false, xb_w = false;            // x == y rows decided on the limbs: new bounds of k1 / k2 (a constant row: of x)
    bool r3v = false;                           // a constant row: values[x] = {xlb0} (:955-961)
    uint32_t d_h2 = 0;
    fp::u256 xlb0 = fp::make(0), xub0 = fp::make(0), xlb1 = fp::make(0), xub1 = fp::make(0);
    uint8_t flip_new = 0;
    uint32_t ev[5] = {0, 0, 0, 0, 0}, nev = 0;  // REQUEUE events, in the reference's order
    uint32_t d_steps = 0, d_nuniq = 0, d_h0 = 0, d_h1 = 0, d_h3 = 0, d_h4 = 0;
    bool slow = false;              // not settled here: the general executor takes the row
    uint32_t reason = 7;
};
__device__ __forceinline__ void fast_decide(const Job& J, const FastIn& I, FastOut& O) {
    const bool live = I.live, xy = I.xy, f2 = I.f2, f4 = I.f4, bigsum = I.bigsum, r4s = I.r4s;
    const uint32_t shape = I.shape, rx = I.rx, kpos = I.kpos, kneg = I.kneg, k1 = I.k1, k2 = I.k2, nA = I.nA, nB = I.nB, nE = I.nE;
    const uint32_t* const w = I.w;
    const uint8_t* const fl = I.fl;
    uint8_t fa = I.fa, fb = I.fb;
    const uint8_t fx = I.fx, flip_in = I.flip_in;
    uint32_t &wva = O.wva, &wvb = O.wvb, &nev = O.nev, &reason = O.reason;
    uint8_t &wfa = O.wfa, &wfb = O.wfb, &flip_new = O.flip_new;
    bool &wa = O.wa, &wb = O.wb, &a01 = O.a01, &b01 = O.b01, &r2 = O.r2, &flip_w = O.flip_w, &xa_w = O.xa_w, &xb_w = O.xb_w, &slow = O.slow;
    fp::u256 &xlb0 = O.xlb0, &xub0 = O.xub0, &xlb1 = O.xlb1, &xub1 = O.xub1;
    uint32_t* const ev = O.ev;
    uint32_t &d_steps = O.d_steps, &d_nuniq = O.d_nuniq, &d_h0 = O.d_h0, &d_h1 = O.d_h1, &d_h3 = O.d_h3, &d_h4 = O.d_h4;
    auto emit = [&](uint32_t v) {
        if (nev == 0) ev[0] = v; else if (nev == 1) ev[1] = v; else if (nev == 2) ev[2] = v; else if (nev == 3) ev[3] = v; else ev[4] = v;
        ++nev;
    };
    if (live && !slow) {
        if (f2) {
            // R2 check_quadratic (:875-942); errors are the general executor's business
            if (shape & SH_R2_BOUNDSERR) { slow = true; reason = 2; }
            else if ((shape & SH_R2) && !(fx & 2)) {
                if (shape & SH_R2_DIV0) { slow = true; reason = 2; }
                else {
                    wva = rx; wa = true; r2 = true;
                    wfa = (uint8_t)((fx | 2) & ~16u);
                    if (shape & SH_R2_IS01) { wfa = (uint8_t)((wfa & ~12u) | 4u); a01 = true; }
                    emit(rx);
                    d_steps = 1; d_h1 = 1;
                }
            }
        } else if (xy && ((fa | fb) & 8u) && k1 != k2 && nE == 2) {
            // x == y with a bound that is neither [0,1] nor [0,p-1] (a constant wired on, say): the same three rules on the
            // limbs themselves, statement for statement exec_xy_lane() (rules_lane.hip.hpp)
            const bool sw = (shape & SH_R56_SWAP) != 0;
            const uint8_t fa_in = fa, fb_in = fb;
            fp::u256 lb0 = (fa & 8u) ? ld256(J.lb + 4ull * k1) : fp::make(0), ub0 = (fa & 8u) ? ld256(J.ub + 4ull * k1) : ((fa & 4u) ? fp::make(1) : fp::pminus1());
            fp::u256 lb1 = (fb & 8u) ? ld256(J.lb + 4ull * k2) : fp::make(0), ub1 = (fb & 8u) ? ld256(J.ub + 4ull * k2) : ((fb & 4u) ? fp::make(1) : fp::pminus1());
            if (((fa ^ fb) & 1u)) {                               // R1
                if (!(fa & 1)) { fa |= 3; emit(k1); } else { fb |= 3; emit(k2); }
                d_nuniq++; d_steps++; d_h0++;
            }
            {                                                     // R4, l == 2
                flip_new = (uint8_t)(flip_in ^ 1);
                flip_w = true;
                const uint32_t new_key = flip_new ? kneg : kpos;
                const bool n_is_a = new_key == k1;
                uint8_t fn = n_is_a ? fa : fb, fo_ = n_is_a ? fb : fa;
                const fp::u256 lbn = n_is_a ? lb0 : lb1, ubn = n_is_a ? ub0 : ub1;
                if (fo_ & 4) {
                    if (!(fp::is_zero(lbn) && fp::is_one(ubn)) && fp::cmp(ubn, fp::make(1)) > 0) {
                        if (n_is_a) { lb0 = fp::make(0); ub0 = fp::make(1); xa_w = true; } else { lb1 = fp::make(0); ub1 = fp::make(1); xb_w = true; }
                        fn = (uint8_t)((fn & ~12u) | 4u | 2u);
                        d_steps++; d_h3++;
                        emit(new_key);
                    }
                    if ((fn & 1) && !(fo_ & 1)) {
                        fo_ |= 3;
                        d_nuniq++; d_steps++; d_h3++;
                        emit(n_is_a ? k2 : k1);
                    }
                }
                if (n_is_a) { fa = fn; fb = fo_; } else { fb = fn; fa = fo_; }
            }
            if (!fp::eq(ub1, ub0) || !fp::eq(lb1, lb0) || ((fa ^ fb) & 1u)) {      // R5
                bool cha = false, chb = false;
                if ((fa ^ fb) & 1u) { fa |= 3; d_nuniq += 2; cha = chb = true; }
                const fp::u256 mn = fp::cmp(ub0, ub1) <= 0 ? ub0 : ub1;
                const fp::u256 mx = fp::cmp(lb0, lb1) >= 0 ? lb0 : lb1;
                const bool na = fp::cmp(ub0, mn) > 0 || fp::cmp(lb0, mx) < 0, nb = fp::cmp(ub1, mn) > 0 || fp::cmp(lb1, mx) < 0;
                if (na) { lb0 = mx; ub0 = mn; xa_w = true; fa = (uint8_t)((fa & ~12u) | bounds_class_bits(mx, mn) | 2u); }
                if (nb) { lb1 = mx; ub1 = mn; xb_w = true; fb = (uint8_t)((fb & ~12u) | bounds_class_bits(mx, mn) | 2u); }
                cha |= na; chb |= nb;
                const uint32_t nset = (cha ? 1u : 0u) + (chb ? 1u : 0u);
                d_steps += nset;
                if (nset) d_h4++;
                if (sw) { if (chb) emit(k2); if (cha) emit(k1); }
                else { if (cha) emit(k1); if (chb) emit(k2); }
            }
            xlb0 = lb0; xub0 = ub0; xlb1 = lb1; xub1 = ub1;
            wva = k1; wfa = fa; wa = fa != fa_in || xa_w;
            wvb = k2; wfb = fb; wb = fb != fb_in || xb_w;
            // R7 / R8 in reach? R7 with two non-unique, known variables and coefficients +-1 fires iff the first one (C order)
            // has ub <= lb (:1267-1269); R8 needs a group tag on every non-unique variable
            const bool nua = !(fa & 1), nub = !(fb & 1);
            if ((nua || nub) && !((nua && !(fa & 2)) || (nub && !(fb & 2)))) {
                const bool tagged = !((nua && !(fa & 16)) || (nub && !(fb & 16)));
                const bool first_is_a = !sw;
                const bool r7 = nua && nub && (first_is_a ? fp::cmp(ub0, lb0) <= 0 : fp::cmp(ub1, lb1) <= 0);
                if (tagged || r7) { slow = true; reason = 4; }
            }
        } else if (I.r3f) {
            // a constant row c_x * x + c_1 = 0 (R3 check_linear :949-988; R1 first, nothing else has anything to do afterwards:
            // x ends up unique, so R7 / R8 find no non-unique variable). fx = x's flag byte.
            uint8_t f = fx;
            uint32_t cnt = 0;
#pragma unroll
            for (uint32_t e = 0; e < 15; ++e) if (e < nE && !(fl[e] & 1)) ++cnt;
            if (cnt == 1 && !(f & 1)) {                                   // R1 (:827-873): x is the one non-unique variable
                f |= 3; emit(rx);
                O.d_nuniq++; O.d_steps++; O.d_h0++;
            } else if (cnt != 0) { slow = true; reason = 1; }             // (a non-unique constant wire: never seen; general executor)
            if (!slow) {
                // (all loads first; the pop of a constant row whose x already holds c -- every pop after the first -- writes nothing)
                const fp::u256 tv = ld256(J.vals + 4ull * I.validx);
                const uint8_t nv = J.nvalues[rx];
                const fp::u256 va = ld256(J.values + 8ull * rx), lbx = ld256(J.lb + 4ull * rx), ubx = ld256(J.ub + 4ull * rx);
                const bool same = nv == 1 && fp::eq(va, tv);
                const bool bsame = fp::eq(lbx, tv) && fp::eq(ubx, tv);
                bool new_info = false;
                if (!same) { O.d_steps++; O.d_h2++; new_info = true; O.r3v = true; }
                if (!(f & 1)) { O.d_nuniq++; new_info = true; }
                f = (uint8_t)(((f | 3) & ~12u) | bounds_class_bits(tv, tv));
                xlb0 = tv; xub0 = tv; xa_w = !bsame;
                wva = rx; wfa = f; wa = f != fx || !same || !bsame;
                if (new_info) emit(rx);
            }
        } else if (xy) {
            if (((fa | fb) & 8u) || k1 == k2 || nE != 2) { slow = true; reason = 3; }
            else {
                const bool sw = (shape & SH_R56_SWAP) != 0;          // C order starts with k2
                const uint8_t fa_in = fa, fb_in = fb;
                // R1 (:827-873)
                if (((fa ^ fb) & 1u)) {
                    if (!(fa & 1)) { fa |= 3; emit(k1); } else { fb |= 3; emit(k2); }
                    d_nuniq++; d_steps++; d_h0++;
                }
                // R4 (:991-1076), l == 2: the row is negated on every visit, the pivot alternates
                {
                    flip_new = (uint8_t)(flip_in ^ 1);
                    flip_w = true;
                    const uint32_t new_key = flip_new ? kneg : kpos;
                    const bool n_is_a = new_key == k1;
                    uint8_t fn = n_is_a ? fa : fb, fo_ = n_is_a ? fb : fa;
                    if (fo_ & 4) {
                        if (!(fn & 4)) {
                            fn = (uint8_t)((fn & ~12u) | 4u | 2u);
                            if (n_is_a) a01 = true; else b01 = true;
                            d_steps++; d_h3++;
                            emit(new_key);
                        }
                        if ((fn & 1) && !(fo_ & 1)) {
                            fo_ |= 3;
                            d_nuniq++; d_steps++; d_h3++;
                            emit(n_is_a ? k2 : k1);
                        }
                    }
                    if (n_is_a) { fa = fn; fb = fo_; } else { fb = fn; fa = fo_; }
                }
                // R5 (:1078-1146): bounds are [0,1] or [0,p-1] here, equal iff the class bits agree
                if (((fa ^ fb) & 4u) || ((fa ^ fb) & 1u)) {
                    bool cha = false, chb = false;
                    if ((fa ^ fb) & 1u) { fa |= 3; d_nuniq += 2; cha = chb = true; }        // key_1 written twice (sic, :1107-1108)
                    const bool na = ((fa ^ fb) & 4u) && !(fa & 4u), nb = ((fa ^ fb) & 4u) && !(fb & 4u);
                    if (na) { fa = (uint8_t)((fa & ~12u) | 4u | 2u); a01 = true; }
                    if (nb) { fb = (uint8_t)((fb & ~12u) | 4u | 2u); b01 = true; }
                    cha |= na; chb |= nb;
                    const uint32_t nset = (cha ? 1u : 0u) + (chb ? 1u : 0u);
                    d_steps += nset;
                    if (nset) d_h4++;
                    if (sw) { if (chb) emit(k2); if (cha) emit(k1); }
                    else { if (cha) emit(k1); if (chb) emit(k2); }
                }
                wva = k1; wfa = fa; wa = fa != fa_in || a01;
                wvb = k2; wfb = fb; wb = fb != fb_in || b01;
                // R7 / R8 (:1235-1348) in reach (see chain.hip.hpp): the general executor decides
                const bool nua = !(fa & 1), nub = !(fb & 1);
                if ((nua || nub) && !((nua && (fa & 18u) != 18u) || (nub && (fb & 18u) != 18u))) { slow = true; reason = 4; }
            }
        } else {
            // products and plain sums: R1 (:827-873)
            bool nuab = false, notknown = false;
            uint32_t cnt = 0, u = 0;
            uint8_t uf = 0;
#pragma unroll
            for (uint32_t e = 0; e < 15; ++e) {
                if (e >= nE) continue;
                const uint8_t f = fl[e];
                if (e < nA + nB) nuab |= !(f & 1);
                else if (!(f & 1)) { if (!cnt) { u = w[1 + e]; uf = f; } ++cnt; if (!(f & 2)) notknown = true; }
            }
            if (bigsum) { if (!(cnt >= 2 && notknown)) { slow = true; reason = 0; } }
            else if (!nuab && cnt == 1) {
                wva = u; wfa = (uint8_t)(uf | 3); wa = true;
                emit(u);
                d_nuniq = 1; d_steps = 1; d_h0 = 1;
            } else if ((f4 || r4s) && cnt > 0 && !notknown) { slow = true; reason = 5; }       // R7 / R8 in reach
            if (r4s && !slow) {
                // R4 checkBinary (:991-1076) on a decomposition of 3..15 terms: nothing happens while some variable other than
                // the pivot lacks bounds exactly [0,1] (:1020-1029) -- the usual state of such a row's pops; otherwise the
                // general executor does the arithmetic
                const uint32_t pivot = (shape & SH_R4_T) ? kpos : kneg;
                bool bad = false;
#pragma unroll
                for (uint32_t e = 0; e < 15; ++e)
                    if (e < nE && w[1 + e] != pivot && !(fl[e] & 4)) bad = true;
                if (!bad) { slow = true; reason = 1; }
            }
        }
    }
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
    const bool r3f = (shape & SH_R3) && !(shape & (SH_HAS_AB | SH_C_EMPTY | SH_R4_T | SH_R4_T2 | SH_R5 | SH_R6));
    const bool live = mine && !is_solved;
    // (a row without a record is declined even when it is solved: its pop still counts the row's non-zeros)
    // A long plain sum (no record: more than 15 terms, up to the 1 025 of a decoder) is re-queued by each of its terms and
    // nearly all of those pops do nothing: two of its variables non-unique, one of them not is_known -- R1 wants exactly
    // one, R7 wants all of them known, R8 all of them tagged. The lane looks at the first 8 terms; if they show that, the
    // pop is settled here (reading exactly those variables), anything else goes to the general executor.
    const bool bigsum = live && (w[0] >> 24) == 0 && f4 && lenC > 15;
    bool alldone = false;
    if (bigsum) {
        const ECNE_GLOBAL uint32_t* const colC = as_global(J.colC);
        const uint32_t c0 = as_global(J.rpC)[row];
        // ... or, once a full walk of the row has found two such terms anywhere in it, at those two (J.hint: "watched" terms)
        const uint32_t h0 = as_global(J.hint)[2u * row], h1 = as_global(J.hint)[2u * row + 1u];
        if (h0 == 0xFFFFFFFEu) { alldone = true; nA = 0; nB = 0; nCc = 0; nE = 0; }        // every term unique (for good): nothing can happen
        else if (h0 != 0xFFFFFFFFu) { w[1] = colC[h0]; w[2] = colC[h1]; nA = 0; nB = 0; nCc = 2; nE = 2; }
        else {
#pragma unroll
            for (uint32_t e = 0; e < 8; ++e) w[1 + e] = colC[c0 + e];
            nA = 0; nB = 0; nCc = 8; nE = 8;
        }
    }
    bool slow = mine && !bigsum && ((w[0] >> 24) == 0 || (!is_solved && ((shape & SH_BIG) || !(xy || f1 || f2 || f4 || r4s || r3f))));
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
    if (live && !slow && xy) { fa = ldF(k1); fb = ldF(k2); }
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
    fin.live = live; fin.xy = xy; fin.f2 = f2; fin.f4 = f4; fin.bigsum = bigsum && !alldone; fin.r4s = r4s && !nop_row; fin.r3f = r3f && !nop_row; fin.validx = validx;
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
        const bool cand0 = mine && !is_solved && (w[0] >> 24) == 0 && slow && (f4 || f1);
        if (rdlane(cand0 ? 1u : 0u, 0)) {
            const uint32_t row0 = rdlane(row, 0), shape0 = rdlane(shape, 0);
            const bool lin0 = !(shape0 & SH_HAS_AB);
            const ECNE_GLOBAL uint32_t* const rpA = as_global(J.rpA); const ECNE_GLOBAL uint32_t* const rpB = as_global(J.rpB);
            const ECNE_GLOBAL uint32_t* const rpC = as_global(J.rpC);
            const ECNE_GLOBAL uint32_t* const cA = as_global(J.colA); const ECNE_GLOBAL uint32_t* const cB = as_global(J.colB);
            const ECNE_GLOBAL uint32_t* const cC = as_global(J.colC);
            const uint32_t a0 = rpA[row0], a1 = rpA[row0 + 1], b0 = rpB[row0], b1 = rpB[row0 + 1], c0 = rpC[row0], c1 = rpC[row0 + 1];
            bool nu = false;
            for (uint32_t k = a0 + (uint32_t)lane; k < a1; k += 64) nu |= !(ldF(cA[k]) & 1);
            for (uint32_t k = b0 + (uint32_t)lane; k < b1; k += 64) nu |= !(ldF(cB[k]) & 1);
            const bool nuab = __ballot(nu) != 0;
            uint32_t cnt = 0, u = 0, uf = 0;
            uint32_t k_nu1 = 0xFFFFFFFFu, k_nu2 = 0xFFFFFFFFu, k_nk = 0xFFFFFFFFu;      // positions: first two non-unique terms, first one not is_known
            bool nk = false;
            // (four strides per trip: the loads of a trip are in flight together -- a 1 025-term sum is 5 dependent
            //  round trips instead of 17)
            for (uint32_t base = c0; base < c1; base += 256) {
                uint32_t v4[4];
                uint8_t f4_[4];
                bool act4[4];
#pragma unroll
                for (uint32_t t = 0; t < 4; ++t) { const uint32_t k = base + 64u * t + (uint32_t)lane; act4[t] = k < c1; v4[t] = act4[t] ? cC[k] : 1u; }
#pragma unroll
                for (uint32_t t = 0; t < 4; ++t) f4_[t] = act4[t] ? ldF(v4[t]) : (uint8_t)3;
#pragma unroll
                for (uint32_t t = 0; t < 4; ++t) {
                    const uint64_t m = __ballot(act4[t] && !(f4_[t] & 1));
                    const uint64_t mk = __ballot(act4[t] && !(f4_[t] & 1) && !(f4_[t] & 2));
                    if (m && cnt == 0) { const int src = __ffsll((long long)m) - 1; u = rdlane(v4[t], (uint32_t)src); uf = rdlane(f4_[t], (uint32_t)src); }
                    if (m && k_nu2 == 0xFFFFFFFFu) {
                        const uint32_t kb = base + 64u * t;
                        uint64_t mm = m;
                        if (k_nu1 == 0xFFFFFFFFu) { k_nu1 = kb + (uint32_t)(__ffsll((long long)mm) - 1); mm &= mm - 1; }
                        if (mm) k_nu2 = kb + (uint32_t)(__ffsll((long long)mm) - 1);
                    }
                    if (mk && k_nk == 0xFFFFFFFFu) k_nk = base + 64u * t + (uint32_t)(__ffsll((long long)mk) - 1);
                    cnt += (uint32_t)__popcll(m);
                    nk |= mk != 0;
                }
            }
            const bool notknown = __ballot(nk) != 0;
            // two terms that keep this row's pops empty for as long as they stay as they are: remembered (see bigsum above)
            const bool bigsum0 = rdlane(bigsum ? 1u : 0u, 0) != 0;
            if (bigsum0 && cnt >= 2 && notknown && lane == 0) {
                ECNE_GLOBAL uint32_t* const hint = as_global(J.hint);
                hint[2u * row0] = k_nk;
                hint[2u * row0 + 1u] = k_nk == k_nu1 ? k_nu2 : k_nu1;
            }
            if (bigsum0 && (cnt == 0 || (cnt == 1 && !nuab)) && lane == 0) as_global(J.hint)[2u * row0] = 0xFFFFFFFEu;      // (after R1, below:) every term unique, for good (reset by the next solve's setup)
            const bool reach78 = lin0 && cnt > 0 && !(cnt == 1 && !nuab) && !notknown;      // R7 / R8 could fire: the general executor decides
            if (!reach78) {
                if (lane == 0) {
                    slow = false;
                    nnz_long = (a1 - a0) + (b1 - b0) + (c1 - c0);
                    if (!nuab && cnt == 1) {
                        wva = u; wfa = (uint8_t)(uf | 3u); wa = true;
                        emit(u);
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
                if (e < nE && ((fl[e] & 3) != 3 || nop_row || r4s || r3f) && w2_get(Tm, w[1 + e] + 1u) < rank) blocked = true;
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
                    my_nnz += nnz_long ? nnz_long : (bigsum ? lenC : nE);
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
                    if (flip_w) { if (flip_lds) flipL[row] = flip_new; else flipG[row] = flip_new; }
                    C.steps += d_steps; C.nuniq += d_nuniq;
                    C.hits[0] += d_h0; C.hits[1] += d_h1; C.hits[2] += fo_.d_h2; C.hits[3] += d_h3; C.hits[4] += d_h4;
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
        my_nnz += nnz_long ? nnz_long : (bigsum ? lenC : nE);
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
        if (flip_w) { if (flip_lds) flipL[row] = flip_new; else flipG[row] = flip_new; }
        C.steps += d_steps; C.nuniq += d_nuniq;
        C.hits[0] += d_h0; C.hits[1] += d_h1; C.hits[2] += fo_.d_h2; C.hits[3] += d_h3; C.hits[4] += d_h4;
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
