// k_solve.hip.hpp — k_solve: setup, outer loop with P1/P2, the queue phase, P3, P4, P5, verdict counts (SPMD over the workgroups of a job).
// Part of the gfx950 device code of libecne_hip (see kernels.hip.hpp for the overview).
#pragma once
#include "rounds.hip.hpp"
#include "oob.hip.hpp"

namespace ecne {

// P4 (:1425-1483) of one outer iteration, all workgroups of the job. A function of its own (like P3's phase 1): with the sweep's locals
// in the kernel body the compiler kept them alive across the whole outer loop -- spilled at its top by every thread of the team.
// Returns true when a job barrier saw an error (the caller leaves the outer loop).
__device__ __noinline__ bool p4_phase(const Job& J, ChunkShared& s_chunk, QState& q, QState* s_q, uint32_t* s_scan, unsigned long long* hits,
                                      unsigned long long& steps, unsigned long long outer, uint32_t my_rank, int* s_err) {
    const int tid = threadIdx.x, lane = lane_id(), w = wave_id();
    const bool master = my_rank == 0;
    const uint32_t gtid = my_rank * ECNE_WG + tid, gstride = J.nwg * ECNE_WG;
    Counters* const ctr = J.ctr;
    // ================= P4 ABZ tagging (:1425-1483): marking and tagging on all workgroups.
    // p4_b[i] / p4_s[i]: B variable and slope variable (bit 31: no slope -> DivideError) of the i-th
    // statically eligible row. A row tags b iff b is not unique, still untagged, and the row is the
    // FIRST such row of b in index order (varmin[b]).
    {
        // (ids above num_variables, one workgroup: the lowest row whose visit reads such a state (:1432, :1443) -- BoundsError there unless a
        //  lower row divides by zero (:1467); every thread evaluates the same few records, oob.hip.hpp)
        const uint32_t r_oob = J.oob ? oob_p4_first(J) : 0xFFFFFFFFu;
        bool my_live = false;      // a candidate whose b is neither unique nor tagged: some row will tag in this pass
        for (uint32_t i = gtid; i < J.nP4; i += gstride) {
            const uint32_t b = J.p4_b[i];
            if (J.flags[b] & 1) continue;
            if (r_oob != 0xFFFFFFFFu && J.p4_list[i] >= r_oob) continue;
            if (J.p4_s[i] & 0x80000000u) { raise(J, K_EDIVZERO); continue; }
            if (ld_agent(&J.varmin[b]) > i) atomicMin(&J.varmin[b], i);
            if (J.abz[b] == -1) my_live = true;
        }
        if (my_live) __hip_atomic_store(&ctr->p4_live, (unsigned)outer, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (stamped with the outer iteration: never reset)
        if (r_oob != 0xFFFFFFFFu) { __syncthreads(); if (tid == 0) raise(J, K_EBOUNDS); }      // (after every DivideError of a lower row: raise keeps the first code)
        if (job_barrier(J, s_err)) return true;
        // No live candidate anywhere: no row can tag, nothing to re-queue -- the pass ends at this barrier (every workgroup reads the
        // same word). The minima stay: as long as b is not unique its first candidate row is the same row.
        if (ld_agent(&ctr->p4_live) == (unsigned)outer) {
        uint32_t my_fired = 0;
        for (uint32_t i = gtid; i < J.nP4; i += gstride) {
            const uint32_t b = J.p4_b[i];
            if ((J.flags[b] & 1) || ld_agent(&J.varmin[b]) != i || J.abz[b] != -1) continue;
            J.abz[b] = (int32_t)(J.p4_s[i] & 0x7FFFFFFFu);
            J.flags[b] |= 2 | 16;    // is_known; bit 4: carries a group tag (abz != -1)
            J.fired[i] = 1;          // by list position; cleared again when the events are collected
            ++my_fired;
        }
        {   // one device atomic per workgroup (the first sweep of a large circuit tags tens of thousands of rows)
            uint32_t wg_fired;
            wg_exclusive_scan(my_fired, s_scan, &wg_fired);
            if (tid == 0 && wg_fired) atomicAdd(&ctr->p4_nfired, wg_fired);
        }
        if (master && tid == 0) ctr->p4_tail = q.tail;   // (thread 0 holds the queue cursor) for p4's job-wide REQUEUE
        if (job_barrier(J, s_err)) return true;
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
            const uint32_t tail0 = ld_agent(&ctr->p4_tail);
            const uint32_t T = gstride;
            int err = 0;
            // 1. the event list: B variables of the fired rows, ascending
            const uint32_t iper = (J.nP4 + T - 1) / T;
            const uint32_t i0 = gtid * iper < J.nP4 ? gtid * iper : J.nP4, i1 = (gtid + 1) * iper < J.nP4 ? (gtid + 1) * iper : J.nP4;
            uint32_t cnt = 0;
            for (uint32_t i = i0; i < i1; ++i) cnt += J.fired[i];
            uint32_t nev = 0;
            uint32_t o = team_exclusive_scan(J, s_chunk, my_rank, cnt, 0, &nev, s_err, &err);
            if (!err) {
                for (uint32_t i = i0; i < i1; ++i)
                    if (J.fired[i]) { J.events[o++] = J.p4_b[i]; J.fired[i] = 0; }
                err = job_barrier(J, s_err);
            }
            // 2. candidates
            uint32_t M = 0, e0 = 0, e1 = 0, cbase = 0;
            if (!err) {
                const uint32_t eper = (nev + T - 1) / T;
                e0 = gtid * eper < nev ? gtid * eper : nev;
                e1 = (gtid + 1) * eper < nev ? (gtid + 1) * eper : nev;
                uint32_t deg = 0;
                for (uint32_t e = e0; e < e1; ++e) { const uint32_t v = J.events[e]; deg += J.fo_ptr[v + 1] - J.fo_ptr[v]; }
                cbase = team_exclusive_scan(J, s_chunk, my_rank, deg, 1, &M, s_err, &err);
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
                err = job_barrier(J, s_err);
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
                    const uint32_t wbase = team_exclusive_scan(J, s_chunk, my_rank, nwin, 0, &W, s_err, &err);
                    if (!err) {
                        uint32_t oq = tail0 + wbase;
                        for (uint32_t jj = j0; jj < j1; ++jj) {
                            const uint32_t cw = J.cand[jj];
                            const uint32_t t = cw & 0x7FFFFFFFu;
                            if (cw & 0x80000000u) { J.queue[oq & J.qmask] = t; J.inq[t] = 1; ++oq; }
                            J.best[t] = 0xFFFFFFFFu;
                        }
                        err = job_barrier(J, s_err);
                    }
                }
                if (!err) {
                    p4_done = true;
                    if (master) {
                        if (w == 0 && lane == 0) { q.tail = tail0 + W; *s_q = q; ctr->p4_nfired = 0; }
                        __syncthreads();
                        q = *s_q;
                        steps += nev;
                        if (w == 0) hits[11] += nev;
                    }
                }
            } else if (!err) {
                // (a B variable with a huge fan-out) the master replays the events one by one
                if (master) {
                    if (w == 0) {
                        for (uint32_t e = 0; e < nev; ++e) requeue(J, q, J.events[e]);
                        if (lane == 0) { *s_q = q; ctr->p4_nfired = 0; }
                    }
                    __syncthreads();
                    q = *s_q;
                    steps += nev;
                    if (w == 0) { hits[11] += nev; hits[15]++; }
                }
                err = job_barrier(J, s_err);
                if (!err) p4_done = true;
            }
            if (err) p4_err = true;
        }
        if (p4_err) return true;
        if (master && p4_fired != 0 && !p4_done) {
            __syncthreads();
            if (tid == 0) ctr->p4_nfired = 0;
            // wave 0 owns the queue cursor during P1-P3; every master thread needs it now
            if (w == 0 && lane == 0) *s_q = q;
            __syncthreads();
            q = *s_q;
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
                unsigned long long p4_fb = 0;
                for (uint32_t eb = 0; eb < nev; eb += 4096) {
                    const uint32_t cnt = (nev - eb) < 4096u ? (nev - eb) : 4096u;
                    tl = resolve_pushes(J, s_chunk, J.events + eb, false, cnt, -1, 0, tl, &p4_fb);      // (every thread counts a fallback in a copy of its own)
                }
                q.tail = tl;
                steps += nev;
                if (w == 0) { hits[11] += nev; hits[15] += p4_fb; }
            }
        }
        }      // (a live candidate)
    }
    return false;
}

// One evaluated row into the caches (p3k / p3h / p3h2) and the group table: k = its non-unique variables of C (0: not eligible, or none
// left), h / h2 the key of that set. A FULL pass (INC = false) starts from an empty table. An INCREMENTAL pass (INC = true: the table and the
// caches still hold the last pass, p3p4_incremental below) first takes the row's old status out of its group. Returns bit 0: a one-variable
// group (fires at its first row, :1402), bit 1: the row's group could be complete with it, bit 2: a group of two and more.
template <bool INC>
__device__ __forceinline__ uint32_t p3_apply(const Job& J, uint32_t r, uint32_t k, uint64_t h, uint64_t h2, uint32_t my_rank, uint32_t ht_cap, uint32_t* s_htn) {
    uint32_t st = 0;
    if constexpr (INC) {
        if (J.p3k[r] >= 2) {
            const uint32_t so = ht_slot(J, J.p3h[r], J.p3h2[r], false);
            if (so != 0xFFFFFFFFu) atomicSub(&J.ht_new[so], 1u);
        }
    }
    J.p3k[r] = (uint8_t)(k > 255 ? 255 : k);
    if (k == 1) st |= 1u;
    else if (k >= 2) {
        J.p3h[r] = h; J.p3h2[r] = h2;   // only read for k >= 2
        bool created = false;
        const uint32_t s = ht_slot(J, h, h2, true, &created);
        if (s != 0xFFFFFFFFu) {
            // "hot" only when this group could be complete with this member (k rows counting the frozen ones): otherwise nobody has
            // to look for trigger rows this pass
            const uint32_t before = atomicAdd(&J.ht_new[s], 1u);
            if (before + 1 + ld_agent(&J.ht_frozen[s]) >= k) st |= 2u;
        }
        st |= 4u;
        if (created) {   // remembered, so that only the slots in use are wiped afterwards
            const uint32_t pos = atomicAdd(s_htn, 1u);
            if (pos < ht_cap) J.ht_list[(size_t)my_rank * ht_cap + pos] = s; else raise(J, K_ECAPACITY);      // (an incremental pass is only started with room for every row it looks at)
        }
    }
    return st;
}

// P3 status of a LONG row, one wavefront: lanes across its entries. Returns "a variable of A or B is not unique" (not eligible, :1366-1382).
__device__ __forceinline__ bool p3_eval_long(const Job& J, uint32_t r, uint32_t& k, uint64_t& h, uint64_t& h2) {
    const int lane = lane_id();
    bool nuab = false;
    for (uint32_t e = J.rpA[r] + lane; e < J.rpA[r + 1]; e += 64) nuab |= !(J.flags[J.colA[e]] & 1);
    for (uint32_t e = J.rpB[r] + lane; e < J.rpB[r + 1]; e += 64) nuab |= !(J.flags[J.colB[e]] & 1);
    k = 0; h = 0; h2 = 0;
    for (uint32_t e = J.rpC[r] + lane; e < J.rpC[r + 1]; e += 64) {
        const uint32_t v = J.colC[e];
        if (!(J.flags[v] & 1)) { ++k; h += mixA(v); h2 += mixB(v); }
    }
    for (int d = 32; d >= 1; d >>= 1) { k += __shfl_xor(k, d, 64); h += __shfl_xor(h, d, 64); h2 += __shfl_xor(h2, d, 64); }
    h = mixA(h + k);
    return __ballot(nuab) != 0;
}

// P3, phase 1 of a pass (:1357-1386): every row from f on that is not dead is evaluated against the current state -- eligibility, the
// number k of non-unique variables of C, the hash of that set -- and counted into its group. A function of its own: called once per
// pass from the kernel body, whose 256 live registers made every call of the per-row evaluation a spill / reload of dozens of them.
__device__ __noinline__ void p3_phase1(const Job& J, uint32_t f, uint32_t gtid, uint32_t gstride, uint32_t my_rank, uint32_t ht_cap, uint32_t* s_htn,
                           bool& my_any, bool& my_hot) {
    const uint32_t nC = J.nC;
    const int lane = lane_id(), w = wave_id();
    for (uint32_t r4 = (f & ~3u) + 4u * gtid; r4 < nC; r4 += 4u * gstride) {
        const uint32_t dead4 = *reinterpret_cast<const uint32_t*>(J.rdead + r4);   // padded to a multiple of 4
        if (((dead4 | (dead4 >> 1)) & 0x01010101u) == 0x01010101u) continue;       // all four dead or long
        uint32_t want = 0, k4[4];
        uint64_t h4[4], g4[4];
        for (uint32_t r = r4 < f ? f : r4; r < r4 + 4 && r < nC; ++r)
            if (!((dead4 >> (8 * (r - r4))) & 3)) want |= 1u << (r - r4);      // dead: every variable unique already (p3k[r] stays 0); long: below
        p3_eval4(J, r4, want, k4, h4, g4);       // the four rows' loads in flight together
        for (uint32_t r = r4 < f ? f : r4; r < r4 + 4 && r < nC; ++r) {
            if (!((want >> (r - r4)) & 1u)) continue;
            uint32_t k = k4[r - r4];
            if (k == 0) J.rdead[r] = 1;        // eligible with no unknown left: nothing can change for this row
            if (k == 0xFFFFFFFFu) k = 0;
            const uint32_t st = p3_apply<false>(J, r, k, h4[r - r4], g4[r - r4], my_rank, ht_cap, s_htn);
            if (st & 1u) atomicMin(&J.ctr->p3_cand1, r);
            if (st & 2u) my_hot = true;
            if (st & 4u) my_any = true;
        }
    }
    // long rows: one WAVEFRONT per row, lanes across its entries (one lane walking a 1 025-term row made
    // its whole workgroup -- and with it every workgroup of the job -- wait ~70 us per sweep)
    for (uint32_t li = my_rank * ECNE_NWAVES + (uint32_t)w; li < J.nLong; li += J.nwg * ECNE_NWAVES) {
        const uint32_t r = J.long_list[li];
        if (r < f || (J.rdead[r] & 1)) continue;               // (wave-uniform)
        uint32_t k; uint64_t h, h2;
        const bool inelig = p3_eval_long(J, r, k, h, h2);
        if (lane == 0) {
            if (inelig) k = 0;
            else if (k == 0) J.rdead[r] |= 1;
            const uint32_t st = p3_apply<false>(J, r, k, h, h2, my_rank, ht_cap, s_htn);
            if (st & 1u) atomicMin(&J.ctr->p3_cand1, r);
            if (st & 2u) my_hot = true;
            if (st & 4u) my_any = true;
        }
    }
}

// P3 (:1357-1417) and P4 (:1425-1483) of an outer iteration in which only a few rows were popped: the MASTER workgroup alone, no job barrier.
// What either sweep finds is a function of the unique bits (P3: which rows have A and B unique and which variables of C are not; P4: a
// statically eligible row whose B variable is neither unique nor tagged) and of the group tags R2 resets (:916, :925) -- and every rule that
// changes either re-queues the rows of the variable it changed (:861-866, :928-933, ...), so the rows whose status can differ from what the
// last pass saw are rows POPPED since then: queue positions [h0, h1) of the ring (every executor on a team leaves the popped rows there).
// The last P3 pass ended without a firing and left its group table and the per-row caches standing (p3_tbl = 1); this pass moves the popped
// rows from their old groups to their new ones. If one of them now is a one-variable group or completes a group -- or a P4 candidate has
// come alive -- nothing is decided here: the full passes run (with everybody, in row order, from a clean table). Otherwise both sweeps would
// have found nothing to do, exactly as the full passes would: no firing needs k rows of one group, and the counts are the full pass's counts.
// Returns non-zero when the full passes are needed.
__device__ __noinline__ int p3p4_incremental(const Job& J, uint32_t h0, uint32_t h1, uint32_t stamp, uint32_t ht_cap, uint32_t* s_htn, uint32_t* s_flag, uint32_t* s_long) {
    const int tid = threadIdx.x, lane = lane_id(), w = wave_id();
    if (tid == 0) { s_flag[0] = 0; s_long[64] = 0; }
    __syncthreads();
    uint32_t bad = 0;
    for (uint32_t n = (uint32_t)tid; n < h1 - h0; n += ECNE_WG) {
        const uint32_t r = J.queue[(h0 + n) & J.qmask];
        if (atomicExch(&J.p3stamp[r], stamp) == stamp) continue;      // popped more than once since the last pass
        const RowInfo ri = J.rinfo[r];
        if (ri.shape & SH_P4) {      // (:1437-1469) some row would tag in this pass: a candidate whose B variable is neither unique nor tagged
            const uint32_t b = ri.kpos;
            if (!(J.flags[b] & 1) && J.abz[b] == -1) bad = 1;
        }
        const uint8_t dead = J.rdead[r];
        if (dead & 1) continue;                                       // every variable unique, for good
        if (dead & 2) {                                               // a long row: one wavefront each, below
            const uint32_t pos = atomicAdd(&s_long[64], 1u);
            if (pos < 64) s_long[pos] = r; else bad = 1;
            continue;
        }
        uint32_t k; uint64_t h, h2;
        p3_eval(J, r, k, h, h2);
        if (k == 0) J.rdead[r] = 1;
        if (k == 0xFFFFFFFFu) k = 0;
        if (p3_apply<true>(J, r, k, h, h2, 0, ht_cap, s_htn) & 3u) bad = 1;      // a one-variable group, or a group that could be complete
    }
    __syncthreads();
    const uint32_t nl = s_long[64] < 64u ? s_long[64] : 64u;
    for (uint32_t li = (uint32_t)w; li < nl; li += ECNE_NWAVES) {
        const uint32_t r = s_long[li];
        uint32_t k; uint64_t h, h2;
        const bool inelig = p3_eval_long(J, r, k, h, h2);
        if (lane == 0) {
            if (inelig) k = 0;
            else if (k == 0) J.rdead[r] |= 1;
            if (p3_apply<true>(J, r, k, h, h2, 0, ht_cap, s_htn) & 3u) bad = 1;
        }
    }
    if (bad) atomicOr(&s_flag[0], 1u);
    __syncthreads();
    const int need_full = (int)s_flag[0];
    __syncthreads();
    return need_full;
}

// P3 (:1357-1417) of one outer iteration, all workgroups of the job: evaluation passes (p3_phase1), the master's search for the
// earliest firing row, the freeze / restart behind a firing, the wipe of the group table. A function of its own (register budget,
// see p4_phase). Returns true when a job barrier saw an error.
// keep_ok (a team, round 5): a pass that ends without a firing leaves the group table and the per-row caches standing (ctr->p3_tbl = 1) for
// the incremental passes of the iterations the master runs alone (p3p4_incremental); a pass that finds such a table wipes it first.
__device__ __noinline__ bool p3_phase(const Job& J, QState& q, unsigned long long* hits, unsigned long long& steps, uint32_t my_rank, uint32_t ht_cap,
                                      uint32_t* s_htn, unsigned long long* s_steps, uint32_t* m_rows, uint32_t* m_vars, unsigned long long* tk, int* s_err, bool keep_ok) {
    const int tid = threadIdx.x, lane = lane_id(), w = wave_id();
    const bool master = my_rank == 0;
    const uint32_t nC = J.nC;
    const uint32_t gtid = my_rank * ECNE_WG + tid, gstride = J.nwg * ECNE_WG;
    Counters* const ctr = J.ctr;
    uint32_t f = 0;   // rows < f are frozen (already swept in this pass)
    bool p3_err = false, keep = false;
    if (keep_ok && ld_agent(&ctr->p3_tbl) != 0) {
        // (every workgroup reads the word the master wrote before the barrier that let them out of the queue phase; the master writes it
        //  again at the end of this function, at least one barrier from here)
        __syncthreads();
        const uint32_t nmine = *s_htn < ht_cap ? *s_htn : ht_cap;
        for (uint32_t i = tid; i < nmine; i += ECNE_WG) {
            const uint32_t s = J.ht_list[(size_t)my_rank * ht_cap + i];
            J.ht_key[s] = 0; J.ht_key2[s] = 0; J.ht_new[s] = 0; J.ht_frozen[s] = 0;
        }
        __syncthreads();
        if (tid == 0) *s_htn = 0;
        if (job_barrier(J, s_err)) return true;
    }
    for (;;) {
        if (master && tid == 0) tk[6]++;
        // (ids above num_variables) the lowest row from f on whose visit reads such a state enters as a row that "fires": the
        // pass then examines what comes before it, and the master raises instead of firing it (phase 3)
        if (J.oob && master && tid == 0) { const uint32_t ro = oob_p3_first(J, f); if (ro != 0xFFFFFFFFu) atomicMin(&ctr->p3_cand1, ro); }
        // phase 1: evaluate rows >= f against the current state
        // (the dead-row bytes are read four rows at a time: most of a large system is dead or idle)
        bool my_any = false, my_hot = false;   // (one store per thread at the end, not one per row, to the two flag words)
        p3_phase1(J, f, gtid, gstride, my_rank, ht_cap, s_htn, my_any, my_hot);
        if (my_any) __hip_atomic_store(&ctr->p3_any, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (my_hot) __hip_atomic_store(&ctr->p3_hot, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (job_barrier(J, s_err)) { p3_err = true; break; }
        const bool any = ld_agent(&ctr->p3_any) != 0;
        const bool hot = ld_agent(&ctr->p3_hot) != 0;
        // Nobody reported a one-variable group or a group that could be complete: nothing can fire in this pass, and every
        // workgroup sees that from the same two words (the master does not touch them on this path) -- the pass ends here,
        // without the master's search and the second barrier (most passes of most circuits: 28 of 28 on ecdsa_like(26)).
        if (!hot && ld_agent(&ctr->p3_cand1) == 0xFFFFFFFFu) {
            if (master && tid == 0 && any) ctr->p3_any = 0;      // (read again a whole outer iteration from now)
            keep = keep_ok && f == 0;                              // nothing fired in this pass: table and caches describe the state as it is
            break;
        }
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
            if (job_barrier(J, s_err)) { p3_err = true; break; }
        }
        // phase 3 (master, wave 0): find the earliest trigger row that passes the test
        if (master) {
            if (w == 0) {
                uint32_t nhot = hot ? ld_agent(&ctr->p3_nhot) : 0;
                if (nhot > J.hotcap) { raise(J, K_ECAPACITY); nhot = 0; }
                uint32_t best = ld_agent(&ctr->p3_cand1);   // k == 1: first arrival of a one-variable group always fires
                for (uint32_t a = 0; a < nhot; ++a) {
                    if ((a & 63u) == 0) job_heartbeat(J);      // (up to hotcap candidates, examined by the master alone)
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
                if (J.oob && best != 0xFFFFFFFFu && best == oob_p3_first(J, f)) raise(J, K_EBOUNDS);      // BoundsError at that row's visit (:1365)
                if (lane == 0) ctr->p3_fire = best;
            }
        }
        if (job_barrier(J, s_err)) { p3_err = true; break; }
        // ready for the next pass -- only now: the other workgroups decide from p3_hot / p3_cand1 whether this pass goes on
        // (above) at their own pace after the first barrier; all of them have done so once they are here
        if (master && tid == 0) { ctr->p3_cand1 = 0xFFFFFFFFu; ctr->p3_nhot = 0; ctr->p3_any = 0; ctr->p3_hot = 0; }
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
                if (lane == 0) *s_steps = steps;
            }
            __syncthreads();
            steps = *s_steps;
        }
        if (job_barrier(J, s_err)) { p3_err = true; break; }   // the firing's writes reach the helpers
        f = fire + 1;
    }
    if (p3_err) return true;
    if (keep_ok && master && tid == 0) ctr->p3_tbl = keep ? 1u : 0u;
    if (keep) return false;
    // leave the table clean for the next outer iteration: every workgroup wipes the slots it created
    __syncthreads();
    {
        const uint32_t nmine = *s_htn < ht_cap ? *s_htn : ht_cap;
        for (uint32_t i = tid; i < nmine; i += ECNE_WG) {
            const uint32_t s = J.ht_list[(size_t)my_rank * ht_cap + i];
            J.ht_key[s] = 0; J.ht_key2[s] = 0; J.ht_new[s] = 0; J.ht_frozen[s] = 0;
        }
        __syncthreads();
        if (tid == 0) *s_htn = 0;
    }
    return false;
}

// Setup (:593-704), all workgroups of the job: initial per-variable state, the known variables, the initial queue (rows with at most
// one variable outside known_variables, ascending), the L2 warm-up of a single-workgroup job. A function of its own (register budget).
__device__ __noinline__ void setup_phase(const Job& J, ChunkShared& s_chunk, QState& q, uint32_t my_rank, uint32_t* s_scan, uint32_t* s_u32, int* s_err) {
    const int tid = threadIdx.x;
    const bool master = my_rank == 0;
    const uint32_t nC = J.nC, nV = J.nV;
    const uint32_t gtid = my_rank * ECNE_WG + tid, gstride = J.nwg * ECNE_WG;
    Counters* const ctr = J.ctr;
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
    if (J.nwg > 1 && J.drain)      // epoch 0 = never marked (drain.hip.hpp)
        for (int p = 0; p < 6; ++p)
            for (uint32_t v = gtid; v <= nV; v += gstride) J.dmk[p][v] = 0;
    if (tid == 0) s_chunk.depoch = 0;
    for (uint32_t r = gtid; r < nC; r += gstride) { J.inq[r] = 0; J.solved[r] = 0; J.flip3[r] = 0; J.best[r] = 0xFFFFFFFFu; J.rdead[r] = 0; J.p3k[r] = 0; J.p3stamp[r] = 0; }
    // the watched pairs of the long rows (a cache in words 1, 2 of their otherwise unused record lines, fastrow.hip.hpp) start empty
    if (J.rec != nullptr)
        for (uint32_t r = gtid; r < nC; r += gstride)
            if ((J.rec[16ull * r] >> 24) == 0) const_cast<uint32_t*>(J.rec)[16ull * r + 1] = 0xFFFFFFFFu;
    for (uint32_t r = gtid; r < nC + J.nSp; r += gstride) J.fired[r] = 0;   // [nC..) = special_solved
    for (uint32_t s = gtid; s <= J.htmask; s += gstride) { J.ht_key[s] = 0; J.ht_key2[s] = 0; J.ht_new[s] = 0; J.ht_frozen[s] = 0; }
    if (master && tid == 0) { ctr->err_key = ~0ull; ctr->p3_cand1 = 0xFFFFFFFFu; ctr->p3_nhot = 0; ctr->p3_any = 0; ctr->p3_hot = 0; ctr->p3_fire = 0xFFFFFFFFu; ctr->q_cut = 0xFFFFFFFFu; ctr->d_cut[0] = ctr->d_cut[1] = 0xFFFFFFFFu; ctr->d_pend2[0] = ctr->d_pend2[1] = 0; ctr->d_flag[0] = ctr->d_flag[1] = 0; }
    job_barrier(J, s_err);
    for (uint32_t i = gtid; i < J.nLong; i += gstride) J.rdead[J.long_list[i]] = 2;   // bit 1: a long row (P3 evaluates it on a wavefront)
    for (uint32_t i = gtid; i < J.nKnown; i += gstride) {
        uint32_t v = J.knowns[i];
        J.flags[v] = 3;
        if (v == 1) { J.nvalues[1] = 1; st256(J.values + 8ull, fp::make(1)); }
    }
    job_barrier(J, s_err);
    // initial queue: rows with at most one variable outside known_variables, ascending (:621-627).
    // Every workgroup owns a contiguous block of rows: count, job-wide scan of the block totals, write.
    q.head = 0; q.tail = 0; q.evout = nullptr; q.nev = 0; q.emit = 0;
    {
        const uint32_t per = (nC + J.nwg - 1) / J.nwg;
        const uint32_t blk0 = my_rank * per < nC ? my_rank * per : nC;
        const uint32_t blk1 = (my_rank + 1) * per < nC ? (my_rank + 1) * per : nC;
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
        uint32_t base = team_exclusive_scan_any(J, s_chunk, my_rank, mine, &total_pushes, s_err, &scan_err);
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
    // ---------------- L2 warm-up. A workgroup meets most rows of a small system exactly once per visit and, after
    // a fresh upload or on another XCD than last time, every first touch of a row's descriptor, entries and
    // fan-out lists would be a round trip beyond this XCD's L2 in the middle of a dependency chain. One
    // streaming pass over the static arrays (a few MB at most, all 512 lanes) makes them L2 hits.
    if (J.nwg == 1 && J.warm_bytes) {
        const uint4* const w0 = (const uint4*)J.rpA;          // the static arrays are one contiguous carve, rpA first
        const uint32_t nq = J.warm_bytes / 16;
        uint32_t acc = 0;
        for (uint32_t i = tid; i < nq; i += ECNE_WG) { const uint4 x = w0[i]; acc ^= x.x ^ x.y ^ x.z ^ x.w; }
        if (acc == 0x9E3779B9u && nq == 0xFFFFFFFFu) s_u32[1] = acc;   // (keeps the loads alive)
    }
}

// P1 (:718-747) and P2 (:750-800): wavefront 0 of the master workgroup, in the reference's order (a function of its own: register budget)
__device__ __noinline__ void p12_phase(const Job& J, QState& q, unsigned long long* hits, unsigned long long& steps) {
    const int lane = lane_id();
    const uint32_t nC = J.nC;
    // P1 (:718-747). 64 specials are tested at a time, one per lane; the ones whose inputs are all
    // unique fire in index order, and after every firing the later lanes look again (its outputs
    // may complete their inputs), which is what the one-by-one sweep would have seen.
    if (J.oob) oob_p1(J, q, hits, steps);      // (ids above num_variables: one special at a time, oob.hip.hpp)
    else
    for (uint32_t base = 0; base < J.nSp; base += 64) {
        const uint32_t i = base + lane;
        int from = 0;
        job_heartbeat(J);
        for (;;) {
            bool can = i < J.nSp && lane >= from && !J.fired[nC + i];   // [nC..) = special_solved
            if (can) {
                // (four inputs per trip, ids first, then their flag bytes: two round trips per four inputs instead of per input)
                const uint32_t e1 = J.sp_in_ptr[i + 1];
                for (uint32_t e = J.sp_in_ptr[i]; e < e1 && can; e += 4) {
                    uint32_t vv[4];
#pragma unroll
                    for (uint32_t k = 0; k < 4; ++k) vv[k] = e + k < e1 ? J.sp_in[e + k] : 1u;
                    uint32_t all = 1;
#pragma unroll
                    for (uint32_t k = 0; k < 4; ++k) all &= e + k < e1 ? (uint32_t)J.flags[vv[k]] : 1u;      // (padding counts as unique whatever the constant wire's state is: a caller's known_variables need not hold it)
                    can = (all & 1) != 0;
                }
            }
            const uint64_t m = __ballot(can);
            if (!m) break;
            const int src = __ffsll((long long)m) - 1;
            const uint32_t is = base + (uint32_t)src;
            if (lane == 0) J.fired[nC + is] = 1;
            steps++; hits[8]++;
            p1_fire_outputs(J, q, is);
            from = src + 1;
        }
    }
    // P2 (:750-800): every (BigMultModP i, BigLessThan j) pair, from the two index lists
    if (J.oob) { if (!J.ctr->error) oob_p2(J, q, hits); }
    else
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
}

// P5 isZero pairs (:1492-1550), ascending over the static candidates: the master workgroup (a function of its own: register budget)
__device__ __noinline__ void p5_phase(const Job& J, QState& q, unsigned long long* hits, unsigned long long& steps, unsigned long long* s_steps, uint32_t my_rank) {
    const int lane = lane_id(), w = wave_id();
    const bool master = my_rank == 0;
    if (master) {
        if (w == 0) {
            // 64 candidates are tested at a time, one per lane; the ones that pass fire in index order,
            // and after every firing the later lanes look again (its newly unique y may complete their A)
            if (J.oob) oob_p5(J, q, hits, steps);
            else
            for (uint32_t base = 0; base < J.nP5; base += 64) {
                const uint32_t i = base + lane;
                const uint32_t r = i < J.nP5 ? J.p5_rows[i] : 0, y = i < J.nP5 ? J.p5_y[i] : 0;
                int from = 0;            // lanes below `from` are done
                if ((base & 4095u) == 0) job_heartbeat(J);
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
            if (lane == 0) *s_steps = steps;
        }
        __syncthreads();
        steps = *s_steps;
    }
}

// The queue phase (:805-1349) as strictly sequential pops on wavefront 0 of the master: queue_mode 1 (exec_row(): the reference's schedule
// verbatim -- debugging, parity runs, systems that name ids above num_variables, known_variables without the constant wire) and
// queue_mode 2 (the chain executor where it applies). A function of its own: inside the kernel body its calls cost the outer loop registers.
__device__ __noinline__ void seq_queue_phase(const Job& J, QState& q, unsigned long long* hits, unsigned long long& steps, unsigned long long& nuniq,
                                             unsigned long long& pops, unsigned long long& pop_nnz) {
    const int tid = threadIdx.x, lane = lane_id();
    const uint32_t nC = J.nC;
    (void)tid;
    const unsigned long long pop_cap = 4096ull + 64ull * (J.rpA[nC] + J.rpB[nC] + J.rpC[nC]);
#ifdef ECNE_POPPROF
    if (tid == 0) { pop_prof().last = wall_clock64(); }
#endif
    if (J.queue_mode == 2 && chain_ok(J)) {
        // QUEUE, strictly sequential pops on the chain executor (chain.hip.hpp)
        while (q.head != q.tail && !J.ctr->error) {
            if (pops > pop_cap) { raise(J, K_ENOCONVERGE); break; }
            chain_pops(J, q, 1u << 16, 0, hits, steps, nuniq, pops, pop_nnz);
        }
        return;
    }
    // QUEUE (:805-1349), strictly sequential pops (debug / parity reference schedule)
    while (q.head != q.tail && !J.ctr->error) {
        if (pops > pop_cap) { raise(J, K_ENOCONVERGE); break; }
        ECNE_PT(7);
        uint32_t row = J.queue[q.head & J.qmask];
        q.head++;
        if (lane == 0) J.inq[row] = 0;
        wg_fence();
        ECNE_PT(0);
        pops++;
        pop_nnz += (J.rpA[row + 1] - J.rpA[row]) + (J.rpB[row + 1] - J.rpB[row]) + (J.rpC[row + 1] - J.rpC[row]);
        if (J.solved[row]) continue;
        ECNE_PT(1);
        if (J.oob) {      // a row that names an id above num_variables: BoundsError at the first read of that state, else nothing
            const int k = oob_pop(J, row);
            if (k == 2) { raise_ranked(J, q.head - 1, K_EBOUNDS); break; }
            if (k == 1) continue;
        }
        // (a caller's known_variables without the constant wire, variable 1 not known yet: R2 as the reference states it)
        if ((J.lv_off & 8u) && !(J.flags[1] & 2) && (J.rinfo[row].shape & SH_C_EMPTY)) { if (r2_constant_wire_free(J, q, row, hits, steps)) break; }
        else exec_row(J, q, row, hits, steps, nuniq);
    }
}

// ---------------------------------------------------------------------------------------- k_solve
struct WgDesc { uint32_t job, rank; };
enum : uint32_t { TEAM_FULL = 1u, TEAM_EXIT = 2u };      // Counters.team_cmd
#ifndef ECNE_INC_MAX
#define ECNE_INC_MAX 4096u      // an outer iteration with at most this many pops is finished by the master alone (p3p4_incremental)
#endif

// TEAM = false: the kernel of single-workgroup jobs. It has no code for helper workgroups, rounds on all workgroups or drain
// rounds: with that code in the same kernel (registers of the callee chain, 1.2 KB more scratch per lane) every phase of a
// small solve -- setup, P3, the chain executor's bursts -- measured 4-9 % slower (tools/ab_old_new.py).
template <bool TEAM>
__device__ __forceinline__ void k_solve_body(const Job* jobs, const WgDesc* wgs) {
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
    // (workgroup-uniform values as scalars: read from the Job in LDS or from memory they would sit in a vector register of every thread
    //  across the whole outer loop -- and be spilled there, see the phase clocks below)
    auto uni32 = [](uint32_t x) -> uint32_t { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); };
    auto uniptr = [](const void* p) -> uint64_t {
        const uint64_t x = (uint64_t)p;
        return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(x >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)x);
    };
    const uint32_t nC = uni32(J.nC), nV = uni32(J.nV);
    const uint32_t my_rank = uni32(me.rank);
    const bool master = my_rank == 0;
    const bool seq_mode = J.queue_mode == 1 || J.queue_mode == 2;   // strictly sequential pops: 1 = exec_row(), 2 = the chain executor where it applies
    // ---------------- LDS residency. A job run by ONE workgroup keeps the two arrays every pop reads and writes
    // -- the unique / is_known flag bytes and the in_queue tags -- in the CU's LDS when they fit the launch's
    // dynamic LDS. The Job's pointers are generic (flat) pointers, so every rule, sweep and REQUEUE below works on
    // either memory unchanged (an LDS access through a generic pointer costs what an L2 hit costs, 106 ns); the
    // chain executor (chain.hip.hpp) reads them with ds_read (27 ns). The host reads the flags afterwards: they
    // are copied back before the verdict count.
    uint8_t* const g_flags = J.flags;
    // dynamic LDS, from the top: tables of the fast wavefront round (always, 7 KB), below them those of the fast workgroup
    // round (56 KB) when there is room -- always on a multi-workgroup job, on a single-workgroup job after its state
    if (tid == 0) {
        J.lds_w2_off = J.lds_bytes >= ECNE_W2_BYTES ? J.lds_bytes - ECNE_W2_BYTES : 0xFFFFFFFFu;
        J.lds_w2b_off = 0xFFFFFFFFu;
        uint32_t off = 0;
        const uint32_t cap = J.lds_bytes > ECNE_W2_BYTES ? J.lds_bytes - ECNE_W2_BYTES : 0u;
        if (J.nwg == 1 && cap) {
            auto take = [&](size_t bytes) -> unsigned char* {
                const uint32_t b = ((uint32_t)bytes + 15u) & ~15u;
                if (bytes > cap || off + b > cap) return nullptr;
                unsigned char* p = ecne_dyn_lds + off;
                off += b;
                return p;
            };
            if (unsigned char* p = take((size_t)nV + 1)) { J.flags = (uint8_t*)p; J.lds_flags_off = (uint32_t)(p - ecne_dyn_lds); }
            if (unsigned char* p = take(2ull * (nC ? nC : 1))) { J.inq = (uint16_t*)p; J.lds_inq_off = (uint32_t)(p - ecne_dyn_lds); }
            if (unsigned char* p = take(nC ? nC : 1)) { J.flip3 = (uint8_t*)p; J.lds_flip_off = (uint32_t)(p - ecne_dyn_lds); }
        }
        if (cap >= off + ECNE_W2_BYTES_BIG) J.lds_w2b_off = cap - ECNE_W2_BYTES_BIG;
    }
    __syncthreads();
    if (J.lds_w2_off != 0xFFFFFFFFu) w2_tables_init(J.lds_w2_off, false);
    if (J.lds_w2b_off != 0xFFFFFFFFu) w2_tables_init(J.lds_w2b_off, true);
    if (tid < 8) { s_chunk.qt[tid] = 0; s_chunk.mt[tid] = 0; }
    if (tid < 16) s_chunk.sd[tid] = 0;
#ifdef ECNE_POPPROF
    if (tid < 8) pop_prof().acc[tid] = 0;
#endif
    const uint32_t gtid = my_rank * ECNE_WG + tid, gstride = uni32(J.nwg) * ECNE_WG;   // job-wide thread index
    Counters* const ctr = (Counters*)uniptr(J.ctr);
    const uint32_t ht_cap = (nC + uni32(J.nwg) - 1) / uni32(J.nwg) + 2048;   // this workgroup's share of ht_list (its rows + slack)
    if (tid == 0) { s_htn = 0; job_barrier_init(); }
#ifdef ECNE_JITTER
    // a random start per workgroup; every other helper of a team is held back well past the master's first commands (0.1-0.3 ms)
    ECNE_JIT(8);
    if (tid == 0 && my_rank != 0 && ((blockIdx.x ^ g_jitter_seed) & 1u)) for (uint32_t i = 0; i < 32u + ((g_jitter_seed >> 3) & 63u); ++i) __builtin_amdgcn_s_sleep(127);
#endif
    // phase clocks (ecne_summary.phase_ms): thread 0 of the master keeps them, in LDS. As eight 64-bit accumulators and a time stamp in
    // the registers of EVERY thread they were live across the whole outer loop: 92 more register spills in this kernel, 0.46 GB of
    // scratch write-backs per ecdsa-scale launch (each job barrier's release writes the dirty scratch lines back) and 0.4 ms.
    __shared__ unsigned long long tk[8], t_last;
    if (tid == 0) { for (int i = 0; i < 8; ++i) tk[i] = 0; t_last = wall_clock64(); }
#define ECNE_TICK(slot) do { if (master && tid == 0) { const unsigned long long t_now = wall_clock64(); tk[slot] += t_now - t_last; t_last = t_now; } } while (0)

    // ---------------- setup (:593-704), all workgroups (setup_phase)
    QState q;
    setup_phase(J, s_chunk, q, my_rank, s_scan, s_u32, &s_err);
    ECNE_TICK(0);
    unsigned long long steps = 0, prev_steps = ~0ull, outer = 0;
    // statistics only the master's wavefront 0 keeps (its lanes in lockstep: they read the same word and write the same sum): in LDS,
    // not in the registers / scratch frame of every thread across the outer loop
    __shared__ unsigned long long s_stat[3], hits[16];
    // (P3's skip test below: steps -- P4's aside -- and num_unique when the last P3 pass began, steps P4 has counted so far, steps in front of the
    //  running P4. In LDS: three 64-bit values alive across the outer loop in the registers of every thread are three more pairs spilled at its top)
    __shared__ unsigned long long s_p3[4];
    if (tid == 0) { s_p3[0] = ~0ull; s_p3[1] = ~0ull; s_p3[2] = 0; s_p3[3] = 0; }
    unsigned long long &nuniq = s_stat[0], &pops = s_stat[1], &pop_nnz = s_stat[2];
    if (tid < 16) hits[tid] = 0;
    if (tid < 3) s_stat[tid] = 0;
    __syncthreads();
    // `steps` is the loop-control value: the master publishes it in ctr->sync_steps before each barrier

    // A TEAM (round 5). The helpers take part in an outer iteration only where there is work for them: the rounds on teams of the queue phase and
    // the full P3 / P4 sweeps. An iteration in which the master popped a few rows by itself is finished by the master alone -- P3 and P4 over the
    // popped rows only (p3p4_incremental), P5 -- and the next one begun, while the helpers go on waiting for the queue phase's next command
    // (`parked`): no job barrier in such an iteration (four before: loop top, end of the queue phase, P3, P4), no sweep over the whole system for
    // the three rows an adder's outputs wake up (ecdsa_like(26): 25 of 28 iterations). TEAM_FULL lets them out into the full sweeps of the
    // iteration the master names, TEAM_EXIT out of the loop.
    const bool team_phase = TEAM && J.nwg > 1 && !seq_mode;
    const bool p3_keep_ok = team_phase && !J.oob;
    bool parked = false;
    __shared__ uint32_t s_inc[4], s_longrows[65];      // s_inc[0]: result of the incremental pass, [1]: queue position up to which P3 has seen the pops
    __shared__ unsigned long long s_team[4], s_team_t0;           // ecne_summary.team
    if (tid == 0) { s_inc[1] = 0; s_team[0] = s_team[1] = s_team[2] = s_team[3] = 0; }
    auto team_leave = [&](uint32_t cmd) {      // master: the queue phase is over for the helpers (they wait at its command barrier)
        if (tid == 0) {
            ctr->team_cmd = cmd; ctr->team_outer = (unsigned)outer;
            __hip_atomic_store(&ctr->q_cmd[bar_local().gen & 1u][0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (the block of the barrier about to be arrived at)
        }
        return job_barrier(J, &s_err);
    };
    for (;;) {
        if (team_phase) {
            if (master) {
                const bool go = prev_steps != steps;      // (:708-711)
                if (!parked) {
                    if (tid == 0) { ctr->sync_steps = steps; ctr->sync_go = go ? 1u : 0u; }
                    if (job_barrier(J, &s_err)) break;
                } else if (!go) { team_leave(TEAM_EXIT); parked = false; }
                if (!go) break;
            } else {
                if (job_barrier(J, &s_err)) break;
                if (ld_agent(&ctr->sync_go) == 0u) break;
            }
        } else {
        if (master && tid == 0) ctr->sync_steps = steps;
        if (job_barrier(J, &s_err)) break;
        steps = __hip_atomic_load(&ctr->sync_steps, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (J.family) {
            // one of the independent parts of a file: the loop goes on while ANY part made progress (the file's one loop, :708-711)
            if (tid == 0) s_u32[2] = family_sync(J, prev_steps != steps, (uint32_t)outer);
            __syncthreads();
            const uint32_t go = s_u32[2];
            __syncthreads();
            if (go != 1u) break;
        } else if (prev_steps == steps) break;   // (:708-711)
        }
        prev_steps = steps;
        outer++;
        if (team_phase && master && tid == 0) s_team_t0 = wall_clock64();
        // ================= P1, P2 and the queue: master only, in the reference's order
        uint32_t team_io = 0;
        if (master) {
            if (w == 0) {
                p12_phase(J, q, hits, steps);
                if (seq_mode) seq_queue_phase(J, q, hits, steps, nuniq, pops, pop_nnz);      // (queue_mode 1 / 2: strictly sequential pops)
                if (lane == 0) { s_q = q; s_steps = steps; }
                if (lane == 0) tk[7] += wall_clock64() - t_last;      // diagnostics: P1 + P2 alone (phase_ms[7]); the slot-1 clock keeps running
            }
            __syncthreads();
            steps = s_steps;
            q = s_q;
            if (!seq_mode && wg_error(J, &s_err)) {
                // P1/P2 raised: the queue phase is skipped, but the helpers are waiting at its command
                // barrier — meet them there (they leave on the error snapshot)
                if (J.nwg > 1) job_barrier(J, &s_err);
                team_io = 2u;
            } else if (!seq_mode) {
                // QUEUE (:805-1349), chunk-parallel schedule; counters other than `steps` live in wave 0
                unsigned long long st2 = steps, nu2 = 0, pp2 = 0, pn2 = 0, ht2[16];
                for (int i = 0; i < 16; ++i) ht2[i] = 0;
                queue_phase_chunked<TEAM>(J, q, s_chunk, ht2, st2, nu2, pp2, pn2, &s_err, team_phase ? &team_io : nullptr);
                steps = st2;
                if (w == 0) { nuniq += nu2; pops += pp2; pop_nnz += pn2; for (int i = 0; i < 8; ++i) hits[i] += ht2[i]; for (int i = 13; i < 16; ++i) hits[i] += ht2[i]; }
            }
        }
        else if (!seq_mode) { if constexpr (TEAM) queue_phase_helper(J, s_chunk, my_rank, &s_err); }
        // A team leaves the queue phase through a job barrier of its own (the master's TEAM_FULL / TEAM_EXIT): that barrier has published the
        // phase's state changes and left every workgroup the same snapshot of the error word -- a second barrier here would only make
        // the helpers' counters visible, and those are folded in behind P3's first barrier instead (4.5 us per outer iteration).
        bool solo_done = false;      // (master) P3 and P4 of this iteration are done: nothing for them to do, found from the popped rows alone
        if (team_phase) {
            if (master) {
                parked = true;       // the helpers sit at the command barrier of the queue phase
                if (team_io & 2u) { parked = false; break; }                                      // (an error has sent them home already)
                if (wg_error(J, &s_err)) { team_leave(TEAM_FULL); parked = false; break; }          // (they leave on the barrier's error snapshot)
                ECNE_TICK(1);
                // the iteration on the master alone? No rounds on the team in this phase (the helpers' counters of such rounds have to be
                // folded in before the loop test, :708), P3's table kept from a pass without a firing, few pops since, room in this workgroup's
                // slot list for a new group per popped row
                const uint32_t h0 = s_inc[1], h1 = q.head;
                // INVARIANT the incremental pass rests on: its work list is ring positions [h0, h1), so every row popped since the last pass must
                // have been STORED in the device ring. The executors that push and pop through the LDS queue mirror without a ring store
                // (crew_rounds, level_rounds<true>, the chain executor) are only reachable with chain_ok(J) -- one workgroup, LDS-resident state --
                // which a team's master never has (nwg > 1); level_rounds<false> stores every push (level.hip.hpp). The guard below makes the
                // dependency explicit: should a mirror-only executor ever be enabled for a team master, the full sweeps run instead of a wrong verdict.
                if (!chain_ok(J) && !(team_io & 1u) && p3_keep_ok && ld_agent(&ctr->p3_tbl) == 1u && h1 - h0 <= ECNE_INC_MAX && h1 - h0 <= J.qmask && s_htn + (h1 - h0) <= ht_cap) {
                    if (tid == 0) { tk[6]++; s_team[1] += h1 - h0; }
                    solo_done = p3p4_incremental(J, h0, h1, (uint32_t)outer, ht_cap, &s_htn, &s_inc[0], s_longrows) == 0;
                    if (tid == 0 && solo_done) s_team[0]++;
                }
                __syncthreads();
                if (tid == 0) s_inc[1] = h1;
                if (!solo_done) { if (tid == 0) s_team[2]++; if (team_leave(TEAM_FULL)) { parked = false; break; } parked = false; }
            } else {
                if (s_err) break;
                if (ld_agent(&ctr->team_cmd) == TEAM_EXIT) break;
                outer = ld_agent(&ctr->team_outer);
            }
        }
        else if (job_barrier(J, &s_err)) break;      // (sequential modes, single-workgroup jobs: the helpers have been waiting here for the master's queue phase)
        if (!team_phase) ECNE_TICK(1);

        if (!solo_done) {
        // ================= P3 linear systems (:1357-1417): evaluation passes on all workgroups (p3_phase)
        // What a pass finds is a function of the unique bits alone (:1360-1386: A and B unique? which variables of C are not?). A pass that
        // ended without a firing therefore ends the same way while no variable has become unique since -- and every rule that makes one
        // unique counts a step or a unique variable (R1..R8, P1, P3, P4, P5). A single-workgroup job holds both counters complete at this
        // point: when neither has moved since the last pass began (so that pass fired nothing either) -- P4's own steps aside: it tags variables and makes them known
        // (:1455-1465), never unique -- the pass is skipped: the last outer iteration of every solve, the idle iterations of a converged
        // part of a split file (EdDSAPoseidon: 0.36 of 0.72 ms of P3; still counted as a pass).
        const bool p3_skip = J.nwg == 1 && !J.oob && outer > 1 && steps - s_p3[2] == s_p3[0] && nuniq == s_p3[1];
        // (the counters as the pass FINDS them: a pass that fires sweeps on behind the firing row only (:1388-1417) -- what its firing did to
        //  the rows in front is the next pass's business, and its own steps make the next comparison fail)
        if (J.nwg == 1) {
            __syncthreads();
            if (tid == 0) { s_p3[0] = steps - s_p3[2]; s_p3[1] = nuniq; }
            __syncthreads();
        }
        if (p3_skip) { if (tid == 0) tk[6]++; }
        else if (p3_phase(J, q, hits, steps, my_rank, ht_cap, &s_htn, &s_steps, m_rows, m_vars, tk, &s_err, p3_keep_ok)) break;
        if (team_phase && master) {
            // fold in what the helpers did during the rounds on teams: their atomics came before P3's first barrier (here, not inside
            // P3's loop: with the counters live across that loop the compiler spilled them, +0.27 GB of scratch writes per launch)
            steps += ctr->q_acc[0];
            if (w == 0) {
                nuniq += ctr->q_acc[1]; pops += ctr->q_acc[10]; pop_nnz += ctr->q_acc[11];
                for (int i = 0; i < 8; ++i) hits[i] += ctr->q_acc[2 + i];
            }
            __syncthreads();
            if (tid < 16) ctr->q_acc[tid] = 0;
        }
        ECNE_TICK(2);

        // ================= P4 ABZ tagging (:1425-1483): marking and tagging on all workgroups (p4_phase)
        if (tid == 0) s_p3[3] = steps;
        if (p4_phase(J, s_chunk, q, &s_q, s_scan, hits, steps, outer, my_rank, &s_err)) break;
        if (tid == 0) s_p3[2] += steps - s_p3[3];
        ECNE_TICK(3);
        } else ECNE_TICK(2);

        // ================= P5 isZero pairs (:1492-1550), ascending over the static candidates (master)
        p5_phase(J, q, hits, steps, &s_steps, my_rank);
        ECNE_TICK(4);
        if (team_phase && master && tid == 0 && solo_done) s_team[3] += wall_clock64() - s_team_t0;      // 100 MHz ticks of the iterations finished alone
    }
    if (team_phase && master && parked) team_leave(TEAM_EXIT);      // (left the loop with the helpers still waiting for a command: an error on the way)

    // ---------------- verdict counts (:1558-1597), all workgroups
    if (J.family && tid == 0 && (s_err || ctr->error)) __hip_atomic_store(&J.family->abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    job_barrier(J, &s_err);
    if (J.flags != g_flags) for (uint32_t v = tid; v <= nV; v += ECNE_WG) g_flags[v] = J.flags[v];   // (LDS-resident flags: the host reads them back)
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
#ifdef ECNE_POPPROF
                if (seq_mode) for (int i = 0; i < 8; ++i) ctr->qticks[i] = pop_prof().acc[i];
#endif
                for (int i = 0; i < 8; ++i) ctr->mticks[i] = s_chunk.mt[i];
                for (int i = 0; i < 16; ++i) ctr->sched[i] = s_chunk.sd[i];
                for (int i = 0; i < 4; ++i) ctr->team_stat[i] = team_phase ? s_team[i] : 0ull;
            }
        }
    }
}

#ifdef ECNE_TEAM_WAVES_PER_EU
__global__ __launch_bounds__(ECNE_WG) __attribute__((amdgpu_waves_per_eu(ECNE_TEAM_WAVES_PER_EU, ECNE_TEAM_WAVES_PER_EU))) void k_solve(const Job* jobs, const WgDesc* wgs) { k_solve_body<false>(jobs, wgs); }
#else
__global__ __launch_bounds__(ECNE_WG) void k_solve(const Job* jobs, const WgDesc* wgs) { k_solve_body<false>(jobs, wgs); }
#endif
#ifdef ECNE_TEAM_WAVES_PER_EU      // (experiment, docs/LAB.md round 6: the team kernel at 128 VGPRs = two workgroups per CU)
__global__ __launch_bounds__(ECNE_WG) __attribute__((amdgpu_waves_per_eu(ECNE_TEAM_WAVES_PER_EU, ECNE_TEAM_WAVES_PER_EU))) void k_solve_team(const Job* jobs, const WgDesc* wgs) { k_solve_body<true>(jobs, wgs); }
#else
__global__ __launch_bounds__(ECNE_WG) void k_solve_team(const Job* jobs, const WgDesc* wgs) { k_solve_body<true>(jobs, wgs); }
#endif

}  // namespace ecne
