// chain.hip.hpp — the chain executor: strictly sequential queue pops (:805-1349) of a single-workgroup job on ONE
// wavefront, built for dependency chains (Poseidon / MiMC / EdDSA: a frontier of 1-8 rows, thousands of levels deep).
// Part of the gfx950 device code of libecne_hip (see kernels.hip.hpp for the overview).
//
// What a pop costs is the number of DEPENDENT memory round trips on its critical path (tools/micro/lat.hip, one
// wavefront, 2.4 GHz: LDS through ds_read 27 ns, L2 hit 103 ns, the same LDS through a generic pointer 106 ns,
// beyond the L2 290 ns). exec_row() walks CSR arrays in L2 and needs about twelve of them per pop (2.4 us). Here:
//   * the queue's next 64 entries live in the REGISTERS of the wavefront (lane i = queue position base + i; pushes
//     are mirrored into the window as well as stored to the ring), so a pop reads its row with v_readlane;
//   * the row arrives in ONE round trip: lanes 0..15 load the 64-byte record rec[row] (lengths + up to 15 variable
//     ids), lanes 16..23 its RowInfo, lane 24 the solved byte, in the same load instruction group;
//   * lane 1 + e then owns entry e of the row: it reads the variable's flag byte from LDS (ds_read_u8) and, at the
//     same time and speculatively, the variable's inline fan-out foi[v] (its REQUEUE list when v becomes unique);
//   * the rule is decided with ballots; REQUEUE tests and sets the in_queue tags in LDS.
// The four row shapes that make up > 99.9 % of the rows of the circomlib circuits are executed here -- products
// a*b = c (R1 only), bit checks (R2), x == y wiring (R1, R4 with l = 2, R5) and plain sums (R1) -- statement for
// statement as exec_row() does; anything else (and any of these when R7 / R8 could fire, or when a bound is neither
// [0,1] nor the initial [0,p-1]) goes through exec_row() / exec_r78_wave() on the same state.
#pragma once
#include "fastrow.hip.hpp"

namespace ecne {

// the chain executor's preconditions: one workgroup, flags and in_queue tags resident in LDS, row records uploaded.
// (Also the precondition of every executor that keeps pushes in the LDS queue mirror only -- crew_rounds, level_rounds<true> --; k_solve's
//  p3p4_incremental relies on their being unreachable for a team's master, see the guard at its call.)
__device__ __forceinline__ bool chain_ok(const Job& J) {
    return J.nwg == 1 && J.rec != nullptr && J.lds_flags_off != 0xFFFFFFFFu && J.lds_inq_off != 0xFFFFFFFFu;
}

struct ChainQ {          // the queue as the chain executor sees it (wave-uniform except win)
    uint32_t head, tail, wbase;
    uint32_t win;        // lane i: queue[(wbase + i) & qmask] for wbase + i < tail
};

__device__ __forceinline__ void chain_window_load(const ECNE_GLOBAL uint32_t* queue, uint32_t qmask, ChainQ& cq) {
    cq.wbase = cq.head;
    const uint32_t pos = cq.head + (uint32_t)lane_id();
    cq.win = (int32_t)(cq.tail - pos) > 0 ? queue[pos & qmask] : 0u;
}

// push row r (wave-uniform) at the tail. A position inside the register window is written to the ring only when the
// window is handed back (chain_window_flush): a store per pop would sit in front of the next pop's loads (the memory
// counter is in-order) and its acknowledgement takes as long as a load.
__device__ __forceinline__ void chain_push(ECNE_GLOBAL uint32_t* queue, uint32_t qmask, ChainQ& cq, uint32_t r) {
    const uint32_t pos = cq.tail;
    if (pos - cq.wbase < 64u) { if ((uint32_t)lane_id() == pos - cq.wbase) cq.win = r; }
    else if (lane_id() == 0) queue[pos & qmask] = r;
    cq.tail = pos + 1;
}
// the unpopped part of the window goes to the ring (before anybody else reads or extends the queue)
__device__ __forceinline__ void chain_window_flush(ECNE_GLOBAL uint32_t* queue, uint32_t qmask, const ChainQ& cq) {
    const uint32_t pos = cq.wbase + (uint32_t)lane_id();
    if ((int32_t)(pos - cq.head) >= 0 && (int32_t)(cq.tail - pos) > 0) queue[pos & qmask] = cq.win;
}

// REQUEUE(v) (:628-633 lists, `if !in_queue[r] push!`), v's inline fan-out given wave-uniform
__device__ __forceinline__ void chain_requeue(const ECNE_GLOBAL uint32_t* fo_rows, ECNE_GLOBAL uint32_t* queue, uint32_t qmask, uint16_t* inq, ChainQ& cq,
                                              uint32_t n, uint32_t r0, uint32_t r1, uint32_t r2) {
    if (n <= 3) {
        if (n >= 1 && inq[r0] == 0) { if (lane_id() == 0) inq[r0] = 1; chain_push(queue, qmask, cq, r0); }
        if (n >= 2 && inq[r1] == 0) { if (lane_id() == 0) inq[r1] = 1; chain_push(queue, qmask, cq, r1); }
        if (n >= 3 && inq[r2] == 0) { if (lane_id() == 0) inq[r2] = 1; chain_push(queue, qmask, cq, r2); }
        return;
    }
    const int lane = lane_id();
    for (uint32_t base = 0; base < n; base += 64) {      // r0 = offset of the list in fo_rows
        const uint32_t k = base + (uint32_t)lane;
        const bool act = k < n;
        const uint32_t r = act ? fo_rows[r0 + k] : 0u;
        const bool push = act && inq[r] == 0;
        const uint64_t m = __ballot(push);
        if (!m) continue;
        const uint32_t pos = cq.tail + (uint32_t)__popcll(m & lanes_below());
        if (push) { inq[r] = 1; if (pos - cq.wbase >= 64u) queue[pos & qmask] = r; }
        // mirror into the register window: position p belongs to lane p - wbase
        for (uint64_t mm = m; mm; mm &= mm - 1) {
            const int src = __ffsll((long long)mm) - 1;
            const uint32_t rr = rdlane(r, (uint32_t)src), pp = rdlane(pos, (uint32_t)src);
            if ((uint32_t)lane == pp - cq.wbase) cq.win = rr;
        }
        cq.tail += (uint32_t)__popcll(m);
    }
}

// Up to max_pops strictly sequential pops on wavefront 0 (all 64 lanes), stopping early when the queue runs empty,
// an error is raised, or -- stop_avail != 0 -- more than stop_avail rows are waiting (a frontier that wide is the
// round schedule's business). Requires flags and inq resident in LDS (J.lds_flags_off / J.lds_inq_off) and J.rec / J.foi.
// Returns 1 when it stopped in front of a live long row (stop_big): the caller pops that one with the whole workgroup -- its events
// resolved in parallel -- where this loop would re-queue them one after the other (a binary decomposition of 89 terms whose R4 / R7
// make every term unique: 67 us here, 12-16 us there).
__device__ __noinline__ uint32_t chain_pops(const Job& J, QState& q, uint32_t max_pops, uint32_t stop_avail, unsigned long long* hits,
                                            unsigned long long& steps, unsigned long long& nuniq, unsigned long long& pops,
                                            unsigned long long& pop_nnz, bool stop_big = false) {
    const int lane = lane_id();
    uint8_t* const F = (uint8_t*)(ecne_dyn_lds + J.lds_flags_off);
    uint16_t* const Q = (uint16_t*)(ecne_dyn_lds + J.lds_inq_off);
    // base pointers as wave-uniform scalars (they come out of the Job in LDS: without this every address is 64-bit vector math)
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
    ECNE_GLOBAL uint8_t* const solved = (ECNE_GLOBAL uint8_t*)uni(J.solved);      // (solved / flip3 / queue stay in device memory, see k_solve)
    // the orientation bytes of the x == y rows (R4, :1001-1011): in LDS when they fit, else device memory
    const bool flip_lds = J.lds_flip_off != 0xFFFFFFFFu;
    uint8_t* const flipL = (uint8_t*)(ecne_dyn_lds + (flip_lds ? J.lds_flip_off : 0u));
    ECNE_GLOBAL uint8_t* const flipG = flip_lds ? (ECNE_GLOBAL uint8_t*)nullptr : (ECNE_GLOBAL uint8_t*)uni(J.flip3);
    // counters live in registers here (the by-reference ones are memory: one round trip per increment) and are
    // folded in before every call that takes the references and at the end
    uint32_t c_steps = 0, c_nuniq = 0, c_pops = 0, c_nnz = 0, c_h0 = 0, c_h1 = 0, c_h3 = 0, c_h4 = 0;
    auto flush = [&]() {
        steps += c_steps; nuniq += c_nuniq; pops += c_pops; pop_nnz += c_nnz;
        hits[0] += c_h0; hits[1] += c_h1; hits[3] += c_h3; hits[4] += c_h4;
        c_steps = c_nuniq = c_pops = c_nnz = c_h0 = c_h1 = c_h3 = c_h4 = 0;
    };
    ChainQ cq;
    cq.head = q.head; cq.tail = q.tail;
    wg_fence();
    chain_window_load(queue, qmask, cq);
    uint32_t done = 0, at_big = 0;
    bool stop = false;
    // The row of a pop -- record (lanes 0..15), descriptor (lanes 16..23), solved and orientation bytes -- is fetched
    // one pop AHEAD whenever the queue already holds the next entry: static data plus two bytes only the pop of that
    // very row changes, so the early copy cannot go stale. Keyed by queue position (pf_pos).
    uint32_t pf_pos = 0xFFFFFFFFu, pf_row = 0, pf_w = 0;
    uint8_t pf_solved = 0, pf_flip = 0;
    auto fetch_row = [&](uint32_t pos) {
        uint32_t i = pos - cq.wbase;
        if (i >= 64) {              // beyond the window: the next 64 entries (our own stores have to have landed)
            chain_window_flush(queue, qmask, cq);
            wg_fence();
            chain_window_load(queue, qmask, cq);
            i = pos - cq.wbase;
        }
        pf_pos = pos;
        pf_row = rdlane(cq.win, i);
        pf_w = 0;
        if (lane < 16) pf_w = rec[16u * pf_row + (uint32_t)lane];
        else if (lane < 24) pf_w = rinfo[8u * pf_row + (uint32_t)(lane - 16)];
        pf_solved = solved[pf_row];
        pf_flip = flip_lds ? flipL[pf_row] : flipG[pf_row];
    };
    while (cq.head != cq.tail && done < max_pops && !stop) {
        if (stop_avail && cq.tail - cq.head > stop_avail) break;
#ifdef ECNE_AVAILHIST
        if (threadIdx.x == 0) { const uint32_t a_ = cq.tail - cq.head; pop_prof().acc[a_ <= 1 ? 0 : a_ <= 2 ? 1 : a_ <= 4 ? 2 : a_ <= 8 ? 3 : a_ <= 16 ? 4 : a_ <= 64 ? 5 : a_ <= 512 ? 6 : 7]++; }
#endif
        ECNE_PT(7);
        if (pf_pos != cq.head) fetch_row(cq.head);
        if (stop_big && (rdlane(pf_w, 16) & SH_BIG) && !pf_solved) { at_big = 1; break; }
        const uint32_t row = pf_row;
        const uint32_t w = pf_w;
        const uint8_t is_solved = pf_solved, flip_in = pf_flip;
        cq.head++;
        ++done;
        if ((int32_t)(cq.tail - cq.head) > 0 && cq.head - cq.wbase < 64) fetch_row(cq.head);   // the next pop's row, if it is known already
        if (lane == 0) Q[row] = 0;                      // (:817) in_queue[row] = false
        const uint32_t w0 = rdlane(w, 0);
        ECNE_PT(0);
        const uint32_t shape = rdlane(w, 16);
        const uint32_t nA = w0 & 0xFFu, nB = (w0 >> 8) & 0xFFu, nCc = (w0 >> 16) & 0xFFu;
        const uint32_t n = nA + nB + nCc;
        const uint32_t e = (uint32_t)lane - 1u;          // my entry of the row, if any
        const bool valid = (w0 >> 24) != 0 && e < n;
        const uint32_t v = valid ? w : 1u;
        uint8_t f = 3;
        u32x4 fo = {0, 0, 0, 0};
        if (valid) { f = F[v]; fo = foi[v]; }
        const bool xy = (shape & (SH_R5 | SH_R4_T | SH_R4_T2 | SH_R3)) == (SH_R5 | SH_R4_T | SH_R4_T2);
        const bool f1 = (shape & SH_HAS_AB) && !(shape & SH_C_EMPTY);
        const bool f2 = (shape & SH_C_EMPTY) != 0;
        const bool f4 = !(shape & (SH_HAS_AB | SH_C_EMPTY | SH_R3 | SH_R4_T | SH_R4_T2 | SH_R5 | SH_R6));
        // x == y rows: lanes 1 and 2 own the two variables in C order; a bound that is neither [0,1] nor [0,p-1]
        // (flag bit 3) sends the row to the general executor, which compares the limbs
        const uint32_t k1 = rdlane(w, 20), k2 = rdlane(w, 21);
        const bool xy_slow = xy && (((rdlane(f, 1) | rdlane(f, 2)) & 8u) || k1 == k2 || n != 2);
        if ((w0 >> 24) == 0 || (shape & SH_BIG) || !(xy || f1 || f2 || f4) || xy_slow) {
            // ---- not one of the fast shapes: the general executor on the same state (generic pointers)
            c_pops++;
            c_nnz += (J.rpA[row + 1] - J.rpA[row]) + (J.rpB[row + 1] - J.rpB[row]) + (J.rpC[row + 1] - J.rpC[row]);
            if (is_solved) continue;
#ifdef ECNE_ROUNDLOG
            if (lane == 0) printf("RG row %u shape %x rec %u n %u xyslow %d\n", row, shape, w0 >> 24, (J.rpA[row + 1] - J.rpA[row]) + (J.rpB[row + 1] - J.rpB[row]) + (J.rpC[row + 1] - J.rpC[row]), (int)xy_slow);
#endif
            QState qq;
            qq.head = cq.head; qq.tail = cq.tail; qq.evout = nullptr; qq.nev = 0; qq.emit = 0;
            flush();
            chain_window_flush(queue, qmask, cq);
            wg_fence();
            exec_row(J, qq, row, hits, steps, nuniq);
            wg_fence();
            cq.tail = qq.tail;
            chain_window_load(queue, qmask, cq);
            if (J.ctr->error) stop = true;
            ECNE_PT(4);
            continue;
        }
        c_pops++;
        c_nnz += n;
        if (is_solved) continue;                         // (:818-820)
        const bool inC = valid && e >= nA + nB;
        const uint64_t m_nuab = __ballot(valid && !inC && !(f & 1));
        const uint64_t m_nuc = __ballot(inC && !(f & 1));
        ECNE_PT(1);
        // REQUEUE of the variable lane `src` owns
        auto requeue_lane = [&](int src) {
            chain_requeue(fo_rows, queue, qmask, Q, cq, rdlane(fo.x, (uint32_t)src), rdlane(fo.y, (uint32_t)src),
                          rdlane(fo.z, (uint32_t)src), rdlane(fo.w, (uint32_t)src));
        };
        // ---- R1 check_unique (:827-873), every fast shape
        bool r1_fired = false;
        if (!m_nuab && __popcll(m_nuc) == 1) {
            const int src = __ffsll((long long)m_nuc) - 1;
            if (lane == src) { f |= 3; F[v] = f; }
            c_nuniq++; c_steps++; c_h0++;
            requeue_lane(src);
            r1_fired = true;
        }
        ECNE_PT(2);
        if (f2) {
            // ---- R2 check_quadratic (:875-942); C is empty, so nothing else can apply to this row
            if (shape & SH_R2_BOUNDSERR) { raise_ranked(J, cq.head - 1, K_EBOUNDS); stop = true; continue; }
            if (shape & SH_R2) {
                const uint32_t x = rdlane(w, 17);
                const uint64_t mx = __ballot(valid && v == x);
                const int src = __ffsll((long long)mx) - 1;            // a lane that owns x (x is in A or B)
                const uint8_t fx = (uint8_t)rdlane(f, (uint32_t)src);
                if (!(fx & 2)) {
                    if (shape & SH_R2_DIV0) { raise_ranked(J, cq.head - 1, K_EDIVZERO); stop = true; continue; }
                    const uint32_t validx = rdlane(w, 22);
                    if (lane == 0) {
                        st256(J.values + 8ull * x, ld256(J.vals + 4ull * validx));
                        st256(J.values + 8ull * x + 4, ld256(J.vals + 4ull * (validx + 1)));
                        J.nvalues[x] = 2;
                        J.abz[x] = -1;
                        uint8_t nf = (uint8_t)((fx | 2) & ~16u);
                        if (shape & SH_R2_IS01) {                      // make_bounds (:923-927)
                            st256(J.lb + 4ull * x, fp::make(0));
                            st256(J.ub + 4ull * x, fp::make(1));
                            nf = (uint8_t)((nf & ~12u) | 4u);
                        }
                        F[x] = nf;
                        solved[row] = 1;
                    }
                    requeue_lane(src);
                    c_steps++; c_h1++;
                }
            }
            continue;
        }
        if (f1) continue;                                 // (:944-946) a non-zero A or B: R3..R8 never run
        // ---- linear rows. R7 and R8 need every non-unique variable of C to be is_known (R8 through its group tag:
        // only P4 sets one, together with is_known; R2 clears it) -- decided after the fast rules from the flag bytes.
        if (xy) {
            // x == y (:991-1146 with l == 2)
            const bool sw = (shape & SH_R56_SWAP) != 0;        // C order starts with k2
            const int l1 = sw ? 2 : 1, l2 = sw ? 1 : 2;        // lanes owning k1 / k2
            uint8_t fa = (uint8_t)rdlane(f, (uint32_t)l1), fb = (uint8_t)rdlane(f, (uint32_t)l2);   // (R1's update included)
            const uint8_t fa_in = fa, fb_in = fb;
            const uint32_t kpos = rdlane(w, 18), kneg = rdlane(w, 19);
            // R4 (:991-1076), l == 2: the row is negated on every visit, the pivot alternates
            {
                const uint8_t o = (uint8_t)(flip_in ^ 1);
                if (lane == 0) { if (flip_lds) flipL[row] = o; else flipG[row] = o; }
                const uint32_t new_key = o ? kneg : kpos;
                const bool n_is_a = new_key == k1;
                uint8_t& fn = n_is_a ? fa : fb;
                uint8_t& fo_ = n_is_a ? fb : fa;
                if (fo_ & 4) {                                 // the other variable has bounds exactly [0,1]
                    if (!(fn & 4)) {                           // pivot still [0,p-1]: ub > 1 -> [0,1]  (:1035-1046)
                        if (lane == 0) { st256(J.lb + 4ull * new_key, fp::make(0)); st256(J.ub + 4ull * new_key, fp::make(1)); }
                        fn = (uint8_t)((fn & ~12u) | 4u | 2u);
                        c_steps++; c_h3++;
                        if (lane == 0) F[new_key] = fn;
                        requeue_lane(n_is_a ? l1 : l2);
                    }
                    if ((fn & 1) && !(fo_ & 1)) {              // pivot unique: the other one becomes unique (:1049-1067)
                        fo_ |= 3;
                        if (lane == 0) F[n_is_a ? k2 : k1] = fo_;
                        c_nuniq++; c_steps++; c_h3++;
                        requeue_lane(n_is_a ? l2 : l1);
                    }
                }
            }
            // R5 (:1078-1146): bounds are [0,1] or [0,p-1] here, equal iff the class bits agree
            if (((fa ^ fb) & 4u) || ((fa ^ fb) & 1u)) {
                bool cha = false, chb = false;
                if ((fa ^ fb) & 1u) { fa |= 3; c_nuniq += 2; cha = chb = true; }        // key_1 written twice (sic, :1107-1108)
                // mn = min ub, mx = max lb: differing classes mean [0,1] vs [0,p-1] -> the wide one is narrowed
                const bool wa = ((fa ^ fb) & 4u) && !(fa & 4u), wb = ((fa ^ fb) & 4u) && !(fb & 4u);
                if (wa) { fa = (uint8_t)((fa & ~12u) | 4u | 2u); if (lane == 0) { st256(J.lb + 4ull * k1, fp::make(0)); st256(J.ub + 4ull * k1, fp::make(1)); } }
                if (wb) { fb = (uint8_t)((fb & ~12u) | 4u | 2u); if (lane == 0) { st256(J.lb + 4ull * k2, fp::make(0)); st256(J.ub + 4ull * k2, fp::make(1)); } }
                cha |= wa; chb |= wb;
                const uint32_t nset = (cha ? 1u : 0u) + (chb ? 1u : 0u);
                c_steps += nset;
                if (nset) c_h4++;
                if (lane == 0) { if (fa != fa_in) F[k1] = fa; if (fb != fb_in) F[k2] = fb; }
                if (sw) { if (chb) requeue_lane(l2); if (cha) requeue_lane(l1); }
                else { if (cha) requeue_lane(l1); if (chb) requeue_lane(l2); }
            }
            // R7 / R8 (:1235-1348). With both coefficients +-1 and bounds [0,1] or [0,p-1], R7's link test fails for two
            // non-unique variables (ratio 1 <= ub - lb) and a single one was R1's; R8 needs a group tag (flag bit 4)
            // on every non-unique variable.
            const bool nua = !(fa & 1), nub = !(fb & 1);
            if ((nua || nub) && !((nua && (fa & 18u) != 18u) || (nub && (fb & 18u) != 18u))) {
                QState qq;
                qq.head = cq.head; qq.tail = cq.tail; qq.evout = nullptr; qq.nev = 0; qq.emit = 0;
                flush();
                chain_window_flush(queue, qmask, cq);
                wg_fence();
                exec_r78_wave(J, qq, row, hits, steps, nuniq);
                wg_fence();
                cq.tail = qq.tail;
                chain_window_load(queue, qmask, cq);
                ECNE_PT(5);
            }
            ECNE_PT(3);
            continue;
        }
        // ---- plain sum (no R3..R6 shape): after R1 only R7 / R8 are left
        {
            uint64_t m_nu = m_nuc, m_nk = __ballot(inC && !(f & 1) && !(f & 2));
            if (r1_fired) { m_nu = 0; m_nk = 0; }
            if (m_nu && !m_nk) {
                QState qq;
                qq.head = cq.head; qq.tail = cq.tail; qq.evout = nullptr; qq.nev = 0; qq.emit = 0;
                flush();
                chain_window_flush(queue, qmask, cq);
                wg_fence();
                exec_r78_wave(J, qq, row, hits, steps, nuniq);
                wg_fence();
                cq.tail = qq.tail;
                chain_window_load(queue, qmask, cq);
            }
        }
    }
    flush();
    chain_window_flush(queue, qmask, cq);
    wg_fence();
    q.head = cq.head;
    q.tail = cq.tail;
    return at_big;
}

}  // namespace ecne
