// rounds.hip.hpp — the queue phase: multi-workgroup rounds, wavefront rounds, the master loop (single-workgroup rounds, bursts, long rows popped alone) and the helper loop.
// Part of the gfx950 device code of libecne_hip (see kernels.hip.hpp for the overview).
#pragma once
#include "job_barrier.hip.hpp"
#include "chain.hip.hpp"
#include "level.hip.hpp"
#include "crew.hip.hpp"

namespace ecne {

// ----------------------------------------------------------------- multi-workgroup queue round
// Same round as in queue_phase_chunked, but executed by ALL workgroups of the job on a window of up
// to nwg * 512 * 2 queue entries — for the thousand-row-wide frontiers of large circuits. The ranks are
// dealt out wavefront by wavefront: wave v of workgroup w owns the 64 * rpl ranks of block v * nwg + w, so
// every workgroup gets low and high ranks alike (the rows that actually fire sit at the low ranks: with
// one contiguous slice per workgroup the first workgroup worked five times longer than the others in the
// execute / expand steps). Cross-workgroup steps use job_barrier (6 per round)
// and two job-wide scans; everything a lane needs later (its rows, its events) it produced itself,
// except cand[] / best[] / inq[] / wmark, which are read after a barrier. Returns nonzero on error.
__device__ uint32_t team_exclusive_scan(const Job& J, ChunkShared& S, uint32_t wgrank, uint32_t x, uint32_t buf,
                                        uint32_t* total, int* s_err, int* err_out) {
    static_assert(ECNE_MAX_NWG <= 256, "the scan over the workgroup totals is one wavefront with four values per lane");
    uint32_t wgtot;
    const uint32_t local = wg_exclusive_scan(x, S.scan, &wgtot);
    if (threadIdx.x == 0) __hip_atomic_store(&J.ctr->q_part[buf][wgrank], wgtot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *err_out = job_barrier(J, s_err);
    // prefix over the workgroups: wavefront 0, four totals per lane (a loop over J.nwg values per thread cost 7 us at 248 workgroups)
    if (wave_id() == 0) {
        const uint32_t lane = (uint32_t)lane_id();
        uint32_t v[4], sum = 0, below = 0;
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k) v[k] = 4u * lane + k < J.nwg ? ld_agent(&J.ctr->q_part[buf][4u * lane + k]) : 0u;
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k) { if (4u * lane + k < wgrank) below += v[k]; sum += v[k]; }
        uint32_t tot;
        const uint32_t ex = wave_excl_scan(sum, &tot);
        if (lane == (wgrank >> 2)) S.bases[0] = ex + below;
        if (lane == 0) S.bases[1] = tot;
    }
    __syncthreads();
    const uint32_t pre = S.bases[0];
    *total = S.bases[1];
    __syncthreads();
    return pre + local;
}

// Job-wide exclusive scan in BLOCK order: wavefront v of workgroup w owns block b = v * nwg + w (64 lanes),
// the scan runs over blocks 0, 1, 2, ... and inside a block over the lanes. Used where the ranks of a round
// are dealt out to the workgroups wavefront by wavefront (queue_round_multi), so that the order of the
// scan is the order of the ranks. (Every workgroup scans all block totals, four per thread: a per-lane loop over
// them was 31 dependent-issue loads at 248 workgroups.)
__device__ uint32_t team_block_scan(const Job& J, ChunkShared& S, uint32_t wgrank, uint32_t x, uint32_t buf, uint32_t* total, int* s_err, int* err_out) {
    static_assert(ECNE_MAX_NWG * ECNE_NWAVES <= 4 * ECNE_WG, "four block totals per thread");
    const uint32_t lane = (uint32_t)lane_id(), b = (uint32_t)wave_id() * J.nwg + wgrank, nblocks = J.nwg * ECNE_NWAVES;
    uint32_t wtot;
    const uint32_t local = wave_excl_scan(x, &wtot);
    if (lane == 0) __hip_atomic_store(&J.ctr->q_blk[buf][b], wtot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *err_out = job_barrier(J, s_err);
    if (nblocks <= 64) {       // a small team (the master by itself: 8 blocks): one value per lane, no workgroup scan
        const uint32_t v = lane < nblocks ? ld_agent(&J.ctr->q_blk[buf][lane]) : 0u;
        uint32_t tot;
        const uint32_t ex = wave_excl_scan(v, &tot);
        *total = tot;
        return (uint32_t)__shfl((int)ex, (int)b, 64) + local;
    }
    const uint32_t t4 = 4u * (uint32_t)threadIdx.x;
    uint32_t v[4], sum = 0;
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) { v[k] = t4 + k < nblocks ? ld_agent(&J.ctr->q_blk[buf][t4 + k]) : 0u; sum += v[k]; }
    uint32_t tot;
    uint32_t run = wg_exclusive_scan(sum, S.scan, &tot);
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) {
        const uint32_t i = t4 + k;
        if (i < nblocks && i % J.nwg == wgrank) S.bases[i / J.nwg] = run;      // block i = wavefront (i / nwg) of this workgroup
        run += v[k];
    }
    __syncthreads();
    const uint32_t pre = S.bases[wave_id()];
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

#ifdef ECNE_FINE_TICKS
#define MTICK(slot) do { if (g == 0) { unsigned long long t_ = wall_clock64(); S.mt[slot] += t_ - mt_last; mt_last = t_; } } while (0)
#else
#define MTICK(slot) do { } while (0)
#endif
// Second half of a round on all workgroups, shared by queue_round_multi and queue_round_drain (drain.hip.hpp): the rows at ranks
// < c have been executed (events in evbuf / the long rows' pool slots, candidate counts in mycand) -- except, with big_exec, the
// registered long rows of the prefix, which are executed here first. REQUEUE resolution in sequential order: job-wide scan of the
// candidate counts, expansion, winners, queue write (4 job barriers, 2 of them carrying a scan; 2 when nobody re-queues anything).
__device__ __noinline__ int multi_finish(const Job& J, ChunkShared& S, uint32_t wgrank, uint32_t head, uint32_t tail, uint32_t c,
                                        uint32_t rpl, uint32_t r0, const uint32_t (&row)[2], uint32_t (&nev)[2], uint32_t bigsl,
                                        uint32_t mycand, bool big_exec, unsigned long long& mt_last, int* s_err,
                                        uint32_t* out_c, uint32_t* out_tail) {
    const int tid = threadIdx.x, lane = lane_id(), w = wave_id();
    Counters* const ctr = J.ctr;
    const uint32_t T = J.nwg * ECNE_WG, g = wgrank * ECNE_WG + tid;
    int err;
    if (S.bl_any) {   // (uniform per workgroup) long rows of the prefix: execute, then candidate offsets of their events
        if (big_exec) big_rows_exec(J, S, c, wgrank);
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
                J.evcnt[S.bl_rank[k]] = 0x80000000u | (wgrank * ECNE_BIGK + k);
                slot[0] = ne;
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
    const uint32_t cbase = team_block_scan(J, S, wgrank, mycand, 0, &M, s_err, &err);   // candidates in rank order
    MTICK(2);
    if (err) return err;
    if (M == 0) {
        // nobody re-queues anything (a block of empty pops -- the rows a multiplexer's sum re-queued, say): the rows of the prefix
        // leave the queue and that is all; the expansion, the winners' scan and two of the barriers drop out
#pragma unroll
        for (uint32_t sl = 0; sl < 2; ++sl)
            if (sl < rpl && r0 + sl < c) J.inq[row[sl]] = 0;
        if (g == 0) ctr->q_cut = 0xFFFFFFFFu;     // ready for the next multi round
        if (tid == 0) big_reset(S);
        if ((err = job_barrier(J, s_err))) return err;
        MTICK(5);
        *out_c = c;
        *out_tail = tail;
        return 0;
    }
    if (M > J.candcap) {
        // a variable with a huge fan-out: the master replays all events sequentially (rare)
        if (wgrank == 0) {
            if (w == 0) {
                QState qq;
                qq.head = 0; qq.tail = tail; qq.evout = nullptr; qq.nev = 0; qq.emit = 0;
                // event counts live in the executing lanes' registers: recount from the fan-out lists is not
                // possible, so each rank's count was also stored (evcnt[rank]; bit 31: a long row, its events are in the pool, their number in word 0 of the rank's slot)
                for (uint32_t r = 0; r < c; ++r) {
                    const uint32_t rr = J.queue[(head + r) & J.qmask];
                    if (lane == 0) J.inq[rr] = 0;
                    wg_fence();
                    uint32_t ne = J.evcnt[r];
                    const uint32_t* evs = J.evbuf + (size_t)r * ECNE_EVCAP;
                    if (ne & 0x80000000u) {   // a long row: its events are in the pool
                        evs = J.bigpool + (size_t)(ne & 0x7FFFFFFFu) * J.bigstride;
                        ne = J.evbuf[(size_t)r * ECNE_EVCAP];
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
    {
        const uint32_t nh = ld_agent(&ctr->q_nhuge);      // events published as huge: the whole team expands them now (rare: one more barrier)
        if (nh) {
            expand_huge_events(J, S, wgrank, nh);
            if ((err = job_barrier(J, s_err))) return err;
            if (g == 0) ctr->q_nhuge = 0;
        }
    }
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

__device__ __noinline__ int queue_round_multi(const Job& J, ChunkShared& S, uint32_t wgrank, uint32_t head, uint32_t tail,
                                             uint32_t n, LaneCtr& C, uint32_t& my_pops, uint32_t& my_nnz, int* s_err,
                                             uint32_t* out_c, uint32_t* out_tail) {
    const int tid = threadIdx.x, lane = lane_id(), w = wave_id();
    Counters* const ctr = J.ctr;
    const uint32_t T = J.nwg * ECNE_WG, g = wgrank * ECNE_WG + tid;
    const uint32_t rpl = (n + T - 1) / T;               // <= 2 by the caller's choice of n
    const uint32_t r0 = (((uint32_t)w * J.nwg + wgrank) * 64u + (uint32_t)lane) * rpl;   // block v * nwg + w, see above
    uint32_t row[2], shape[2], xv[2];
    uint32_t live = 0, noop = 0, noop_b = 0;
    int err;
    unsigned long long mt_last = wall_clock64();
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
    // ---- rows of the four common shapes take the RECORD path (one row per lane rounds only): the pop is decided in registers
    // from rec[row] and the flag bytes (fast_decide, wave2.hip.hpp) -- one round trip for the row, one for its flags -- instead
    // of three CSR walks (no-op test, access sets twice) plus the executor's own. Same hazard rules as for every other row of
    // the round: what the decision would write is marked, what it read is checked, the decision is taken again at commit time
    // (its inputs cannot have changed for a row of the prefix) and committed.
    bool fz = false;
    FastIn fin;
    uint32_t fz_wva = 0, fz_wvb = 0, fz_cls = 0;        // written variables; bit0/1: U / B class of wva, bit2/3: of wvb
    uint8_t fz_flip = 0;
    if (J.rec != nullptr && rpl == 1 && r0 < n && (live & 1u) && !(shape[0] & SH_BIG)) {
        const ECNE_GLOBAL u32x4* const rec = as_global(reinterpret_cast<const u32x4*>(J.rec));
        const ECNE_GLOBAL uint8_t* const Fg = as_global(J.flags);
        const RowInfo ri = J.rinfo[row[0]];
        u32x4 w4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) w4[i] = rec[4u * row[0] + (uint32_t)i];
        fz_flip = J.flip3[row[0]];
#pragma unroll
        for (int i = 0; i < 4; ++i) { fin.w[4 * i] = w4[i].x; fin.w[4 * i + 1] = w4[i].y; fin.w[4 * i + 2] = w4[i].z; fin.w[4 * i + 3] = w4[i].w; }
        fin.shape = ri.shape; fin.rx = ri.x; fin.kpos = ri.kpos; fin.kneg = ri.kneg; fin.k1 = ri.k1; fin.k2 = ri.k2;
        fin.nA = fin.w[0] & 0xFFu; fin.nB = (fin.w[0] >> 8) & 0xFFu; fin.nE = fin.nA + fin.nB + ((fin.w[0] >> 16) & 0xFFu);
        fin.xy = (ri.shape & (SH_R5 | SH_R4_T | SH_R4_T2 | SH_R3)) == (SH_R5 | SH_R4_T | SH_R4_T2);
        const bool f1 = (ri.shape & SH_HAS_AB) && !(ri.shape & SH_C_EMPTY);
        fin.f2 = (ri.shape & SH_C_EMPTY) != 0;
        fin.f4 = !(ri.shape & (SH_HAS_AB | SH_C_EMPTY | SH_R3 | SH_R4_T | SH_R4_T2 | SH_R5 | SH_R6));
        fin.live = true; fin.bigsum = false; fin.flip_in = fz_flip;
        if ((fin.w[0] >> 24) != 0 && (fin.xy || f1 || fin.f2 || fin.f4)) {
            const bool walk = !fin.xy && !fin.f2;
#pragma unroll
            for (uint32_t e = 0; e < 15; ++e) fin.fl[e] = (walk && e < fin.nE) ? Fg[fin.w[1 + e]] : (uint8_t)3;
            fin.fa = fin.fb = fin.fx = 3;
            if (fin.xy) { fin.fa = Fg[fin.k1]; fin.fb = Fg[fin.k2]; }
            if (fin.f2 && (ri.shape & SH_R2)) fin.fx = Fg[fin.rx];
            FastOut D;
            fast_decide(J, fin, D);
            if (!D.slow) {
                fz = true;
                fz_wva = D.wva; fz_wvb = D.wvb;
                const uint8_t ia = fin.xy ? fin.fa : fin.f2 ? fin.fx : (uint8_t)(D.wfa & ~3u), ib = fin.fb;     // flag bytes before (products / sums only set bits 0, 1)
                if (D.wa) fz_cls |= (((D.wfa ^ ia) & 3u) ? 1u : 0u) | ((((D.wfa ^ ia) & ~3u) || D.a01 || D.xa_w || D.r2) ? 2u : 0u);
                if (D.wb) fz_cls |= (((D.wfb ^ ib) & 3u) ? 4u : 0u) | ((((D.wfb ^ ib) & ~3u) || D.b01 || D.xb_w) ? 8u : 0u);
            }
        }
    }
    __syncthreads();
    // ---- mark (write sets)
#pragma unroll
    for (uint32_t sl = 0; sl < 2; ++sl) {
        if (sl >= rpl || r0 + sl >= n) continue;
        const uint32_t rank = r0 + sl;
        if (!(live & (1u << sl))) continue;
        if (sl == 0 && fz) {
            if (fz_cls & 1u) atomicMin(&J.wmarkU[fz_wva], rank);
            if (fz_cls & 2u) atomicMin(&J.wmarkB[fz_wva], rank);
            if (fz_cls & 4u) atomicMin(&J.wmarkU[fz_wvb], rank);
            if (fz_cls & 8u) atomicMin(&J.wmarkB[fz_wvb], rank);
            continue;
        }
        if (shape[sl] & SH_BIG) {   // plain long rows ride along, handled by this workgroup as a whole (see big_rows_*)
            if (!big_plain(shape[sl]) || !big_register(S, row[sl], rank)) atomicMin(&S.cut, rank);
            continue;
        }
        const RowInfo ri = J.rinfo[row[sl]];
        bool nb = false;
        if (row_is_noop(J, row[sl], ri, nb)) { noop |= 1u << sl; if (nb) noop_b |= 1u << sl; continue; }
        row_mark_global(J, row[sl], shape[sl], xv[sl], rank);
    }
    __syncthreads();
    if (S.bl_any) big_rows_mark(J, S);
    if ((err = job_barrier(J, s_err))) return err;
    MTICK(0);
    // ---- check (every lane keeps its own lowest cut; one LDS atomic per wavefront afterwards)
    uint32_t mycut = 0xFFFFFFFFu;
#pragma unroll
    for (uint32_t sl = 0; sl < 2; ++sl) {
        if (sl >= rpl || r0 + sl >= n || !(live & (1u << sl)) || (shape[sl] & SH_BIG)) continue;
        const uint32_t rank = r0 + sl;
        bool blocked = false;
        if (sl == 0 && fz) {
            // read set of the record path: the flag bytes of the entries that can still change (products, sums; a sum also reads
            // the group tag), flags and bounds of both variables (x == y), of x (bit check). An earlier writer blocks the row, a
            // later one cuts the prefix in front of itself (the rows of a prefix execute in place, concurrently).
            auto see = [&](uint32_t m) { if (m < rank) blocked = true; else if (m > rank && m < mycut) mycut = m; };
            if (fin.xy) { see(ld_agent(&J.wmarkU[fin.k1])); see(ld_agent(&J.wmarkB[fin.k1])); see(ld_agent(&J.wmarkU[fin.k2])); see(ld_agent(&J.wmarkB[fin.k2])); }
            else if (fin.f2) { if (fin.shape & SH_R2) { see(ld_agent(&J.wmarkU[fin.rx])); see(ld_agent(&J.wmarkB[fin.rx])); } }
            else {
#pragma unroll
                for (uint32_t e = 0; e < 15; ++e)
                    if (e < fin.nE && (fin.fl[e] & 3) != 3) { see(ld_agent(&J.wmarkU[fin.w[1 + e]])); if (fin.f4) see(ld_agent(&J.wmarkB[fin.w[1 + e]])); }
            }
            if (blocked && rank < mycut) mycut = rank;
            continue;
        }
        if (noop & (1u << sl)) {
            if (noop_b & (1u << sl)) blocked = row_noop_blocked_global(J, row[sl], rank);
        } else {
            const uint32_t m = row_check_global(J, row[sl], shape[sl], xv[sl], rank);
            if (m < mycut) mycut = m;
        }
        if (blocked && rank < mycut) mycut = rank;
    }
    { const uint32_t wm = wave_min(mycut); if (lane == 0 && wm != 0xFFFFFFFFu) atomicMin(&S.cut, wm); }
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
        if (sl == 0 && fz) {
            if (fz_cls & 1u) J.wmarkU[fz_wva] = 0xFFFFFFFFu;
            if (fz_cls & 2u) J.wmarkB[fz_wva] = 0xFFFFFFFFu;
            if (fz_cls & 4u) J.wmarkU[fz_wvb] = 0xFFFFFFFFu;
            if (fz_cls & 8u) J.wmarkB[fz_wvb] = 0xFFFFFFFFu;
            if (r0 >= c) continue;
            // commit: the decision again (its inputs are what they were: nobody in the prefix writes what this row reads)
            FastOut D;
            fast_decide(J, fin, D);
            J.inq[row[0]] = (uint16_t)2;
            J.prank[row[0]] = r0;
            my_pops++;
            my_nnz += fin.nE;
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
            uint32_t* ev = J.evbuf + (size_t)r0 * ECNE_EVCAP;
            nev[0] = D.nev;
#pragma unroll
            for (uint32_t e = 0; e < 5; ++e) if (e < D.nev) { ev[e] = D.ev[e]; mycand += J.fo_ptr[D.ev[e] + 1] - J.fo_ptr[D.ev[e]]; }
            J.evcnt[r0] = D.nev;
            continue;
        }
        if ((live & (1u << sl)) && !(shape[sl] & SH_BIG) && !(noop & (1u << sl))) row_unmark_global(J, row[sl], shape[sl], xv[sl]);
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
            else { C.rank = head + r0 + sl; exec_row_lane(J, row[sl], J.evbuf + (size_t)(r0 + sl) * ECNE_EVCAP, nev[sl], C); }
        }
        uint32_t* ev = J.evbuf + (size_t)(r0 + sl) * ECNE_EVCAP;
        for (uint32_t e = 0; e < nev[sl]; ++e) mycand += J.fo_ptr[ev[e] + 1] - J.fo_ptr[ev[e]];
        J.evcnt[r0 + sl] = nev[sl];      // for the sequential replay fallback
    }
    return multi_finish(J, S, wgrank, head, tail, c, rpl, r0, row, nev, bigsl, mycand, true, mt_last, s_err, out_c, out_tail);
}

}  // namespace ecne
#include "drain.hip.hpp"
namespace ecne {

// ------------------------------------------------------------------------------- wavefront round
// The same round as in queue_phase_chunked for a window of at most 64 queue entries, executed by ONE
// wavefront (lane = rank) without a single workgroup barrier: narrow dependency levels (a dozen rows
// wide) are the bulk of the rounds of a deep circuit and a workgroup round costs them ~20 us of barriers
// and idle lanes. Write-marks use the first 1024 slots of the LDS hash table (wiped afterwards), the
// REQUEUE events are resolved with wave scans. Returns the number of committed rows, or 0xFFFFFFFF
// without having touched anything when the window starts with a live long row (the caller's general path
// takes it). Wave 0 only, all 64 lanes.
#define ECNE_WSLOTS 1024
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
    if (mine && rank < c && live) {
        if (noop) { if ((shape & SH_R4_T) && (shape & SH_R4_T2)) J.flip3[row] ^= 1; }
        else { C.rank = head + rank; exec_row_lane(J, row, J.evbuf + (size_t)rank * ECNE_EVCAP, nev, C); }
    }
    wg_fence();   // the rank tags (and the state changes) are visible to every lane before the pushes are resolved
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
        // (best[] is reset only now: every candidate above was judged against the same minima, and those
        //  loads have all returned -- their values went into the ballots)
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
#ifndef ECNE_BURST_C
#define ECNE_BURST_C 8      // a round that commits fewer rows than this on a short queue switches to sequential bursts
#endif
#ifndef ECNE_CHAIN_AVAIL
#define ECNE_CHAIN_AVAIL 0      // queue length up to which the chain executor pops without trying a round first (measured: wave rounds win from 3-4 rows; 0 = off)
#endif
#ifndef ECNE_FAST_CHAIN
#define ECNE_FAST_CHAIN 3      // fast wavefront rounds the master of a team may run back to back in front of the one the policy looks at
#endif
#ifndef ECNE_CHAIN_BURST_C
#define ECNE_CHAIN_BURST_C 12
#endif
#ifndef ECNE_WGROW
#define ECNE_WGROW 2
#endif
#ifndef ECNE_WMIN
#define ECNE_WMIN 64
#endif
#ifndef ECNE_MULTI_MIN
#define ECNE_MULTI_MIN 128    // queued rows from which a round runs on all workgroups of the job (measured optimum)
#endif
#ifndef ECNE_V2_BACKLOG
#define ECNE_V2_BACKLOG 8192
#endif
#ifndef ECNE_V2_DECLINES
#define ECNE_V2_DECLINES 4
#endif
#ifndef ECNE_V2_WINDOW
#define ECNE_V2_WINDOW 128      // rows committed in a row without a dependency cut before a round goes to all workgroups (measured: 128 / 256 / 1024 -> 27.3 / 28.9 / 29.5 ms)
#endif
#ifndef ECNE_V2WG_AVAIL
#define ECNE_V2WG_AVAIL 2048     // ... and with the fast WORKGROUP round (512 rows in ~13 us) from this many queued rows
#endif
#ifndef ECNE_V2WG_WINDOW
#define ECNE_V2WG_WINDOW 2048
#endif
#ifndef ECNE_V2_AVAIL
#define ECNE_V2_AVAIL 256     // ... when the fast wavefront round is available: it takes 64 rows in ~9 us, a multi-workgroup round costs ~55 us
#endif
// (the same value on every workgroup of the job: the chained rounds derive their schedule from it)
__device__ __forceinline__ bool fast_wave_ok(const Job& J) { return J.rec != nullptr && J.lds_w2_off != 0xFFFFFFFFu; }
#ifndef ECNE_V2WG
#define ECNE_V2WG 0      // the fast round on the whole workgroup (512 rows, wave2.hip.hpp WG = true): bit-exact, but measured slower than
#endif                   // wavefront rounds + multi-workgroup rounds on every workload (ecdsa_like(26) 29.7 vs 27.3 ms, secp256k1 9.4 vs 8.4 ms): off
__device__ __forceinline__ bool fast_wg_ok(const Job& J) { return ECNE_V2WG && J.rec != nullptr && J.lds_w2b_off != 0xFFFFFFFFu; }
__device__ __forceinline__ uint32_t multi_min(const Job& J) { return fast_wg_ok(J) ? ECNE_V2WG_AVAIL : fast_wave_ok(J) ? ECNE_V2_AVAIL : ECNE_MULTI_MIN; }
// the adaptive single-workgroup window has to have grown this far (rounds committing everything they looked at, doubling
// it) before a round goes to all workgroups: with the fast round a multi-workgroup round pays from ~500 committed rows
__device__ __forceinline__ uint32_t multi_window_min(const Job& J) { return fast_wg_ok(J) ? ECNE_V2WG_WINDOW : fast_wave_ok(J) ? ECNE_V2_WINDOW : ECNE_MULTI_MIN; }
// drain rounds (drain.hip.hpp) instead of prefix rounds on all workgroups: the job has row records and the host did not switch them off
__device__ __forceinline__ bool drain_ok(const Job& J) { return J.drain != 0 && J.rec != nullptr; }
// test hook (ECNE_DRAIN=2): every frontier of two rows and more goes to a drain round, whatever the streak -- the level logic then
// sees the dependency chains and narrow windows the schedule normally keeps away from it
__device__ __forceinline__ bool drain_eager(const Job& J) { return (J.drain & 2u) && J.rec != nullptr && J.nwg > 1; }
__device__ __forceinline__ uint32_t multi_cap(const Job& J) { return J.nwg * ECNE_WG * (drain_ok(J) ? 1u : 2u); }
#ifndef ECNE_DRAIN_GROW
#define ECNE_DRAIN_GROW 6      // a drain that needed at most this many levels doubles the next window ...
#endif
#ifndef ECNE_DRAIN_GROW4
#define ECNE_DRAIN_GROW4 2     // ... at most this many: it quadruples
#endif
#ifndef ECNE_MWINDOW0
#define ECNE_MWINDOW0 16384   // window of the first PREFIX round on all workgroups of a queue phase (a dependency cuts it short: start small)
#endif
#ifndef ECNE_MWINDOW0_DRAIN
#define ECNE_MWINDOW0_DRAIN 131072   // ... of the first DRAIN round: the whole team's worth (a drain is not cut short by a dependency, it takes more levels;
                                     // ramping up from 16 384 cost three to five extra rounds per wide phase of ecdsa_like(26): 7.98 -> 7.50 ms)
#endif
#ifndef ECNE_DRAIN_SHRINK
#define ECNE_DRAIN_SHRINK 16   // ... one that needed more than this many halves it (a dependency chain inside the window: every level pays four job barriers)
#endif
__device__ __forceinline__ void drain_window_update(uint32_t levels, uint32_t nm, uint32_t cap_n, uint32_t& mwindow) {
    if (levels <= ECNE_DRAIN_GROW) {
        const uint32_t g = levels <= ECNE_DRAIN_GROW4 ? 4u : 2u;       // (a drain of one or two levels: the frontier is wide and flat, the window goes up in two steps instead of four)
        mwindow = (mwindow * g < cap_n) ? mwindow * g : cap_n;
    }
    else if (levels > ECNE_DRAIN_SHRINK) mwindow = nm / 2 > 1024u ? nm / 2 : 1024u;
}
#ifndef ECNE_SOLO_AVAIL
#define ECNE_SOLO_AVAIL 32     // solo drains: from this many queued rows on, when a fast round committed less than 1 / ECNE_SOLO_RATIO of them
#endif
#ifndef ECNE_SOLO_RATIO
#define ECNE_SOLO_RATIO 8
#endif
// ---- chained multi-workgroup rounds
// After a multi-workgroup round every workgroup of the job knows the new head, tail and prefix length, so the
// decision "the next round is a multi-workgroup round again, over nm rows" can be taken by every workgroup
// for itself: the master skips its command (one job barrier and the loads of its own window per round), the
// helpers go straight into the next round. Both sides run the two functions below on the same inputs; the
// state they read (queue head entry, its shape, its solved flag) was published by the round's last barrier and
// nobody writes between rounds. A chain ends after ECNE_CHAIN_MAX rounds so that the master's error / pop-cap
// polling keeps its cadence.
#ifndef ECNE_CHAIN_MAX
#define ECNE_CHAIN_MAX 32
#endif
__device__ __forceinline__ void multi_window_update(uint32_t cm, uint32_t nm, uint32_t cap_n, uint32_t& mwindow, uint32_t& window) {
    if (cm == nm) mwindow = (mwindow * 2 < cap_n) ? mwindow * 2 : cap_n;
    else if (cm < nm / 4) {
        const uint32_t wn = 4 * cm;
        if (wn >= 4096) mwindow = wn;
        else { mwindow = 4096; window = wn < ECNE_WMIN ? ECNE_WMIN : (wn < ECNE_RPL * ECNE_WG ? wn : ECNE_RPL * ECNE_WG); }
    }
}
// rows of the next chained round, 0 = back to the master's own loop (same tests as at its top)
__device__ __forceinline__ uint32_t multi_chain_next(const Job& J, uint32_t head, uint32_t tail, uint32_t window, uint32_t mwindow,
                                                     uint32_t c_last, uint32_t n_last) {
    // with the fast rounds at hand, a round that a dependency cut short hands the frontier back to them: the next prefix is
    // likely short as well -- on ecdsa_like(26) a 4 096-row window cut at row 3 968 (a multiplexer's 1 025-term sum) was
    // followed by a round of ONE row 78 times, 65 us each -- and they find that out for a fifth of the price
    if (fast_wave_ok(J) && c_last != n_last) return 0;
    const uint32_t avail = tail - head;
    const uint32_t n = avail < window ? avail : window;
    if (drain_eager(J)) { if (avail < 2) return 0; }
    else if (n <= 64 || avail < multi_min(J) || window < multi_window_min(J)) return 0;
    // (round 6: from STATIC data only. Every member of the team derives the chain's next step by itself behind the round's last barrier, and
    //  the master, once it has decided that the chain is over, goes on alone at once -- popping this very row. The test used to read
    //  J.solved[row0] as well: a member held up behind the barrier could then see the row solved by the master's pop and derive one more
    //  round that nobody else runs. A solved long row at the head now ends the chain too; the master pops it for nothing and commands the next one.)
    const uint32_t row0 = J.queue[head & J.qmask];
    const uint32_t shape0 = J.rinfo[row0].shape;
    if ((shape0 & SH_BIG) && !big_plain(shape0)) return 0;   // a long row that is popped alone
    const uint32_t cap_n = multi_cap(J);
    uint32_t nm = avail < cap_n ? avail : cap_n;
    if (nm > mwindow) nm = mwindow;
    return nm;
}

// ---- a chain of rounds on a TEAM: the first K workgroups of the job (K = J.nwg: everybody, the job's own barrier; K < J.nwg: a
// sub-team with a flat barrier of its own, the other workgroups wait at the job barrier for the next command; K = 1: the master
// by itself, "solo": its barriers are workgroup barriers). A frontier of 6 000 rows is 12 workgroups' worth of lanes: 248
// workgroups meeting six times per round cost 64 us per round, 43 cost 45. For the time of the chain the workgroup's own copy of
// the job (LDS, k_solve) says nwg = K, subteam = 1 -- every function of the round reads the team from there (a private copy of the
// 700-byte Job per thread cost 17 us per chain).
struct ChainState { uint32_t head, tail, window, mwindow, streak, rounds, sd[3]; unsigned long long rows; };
__device__ __noinline__ int multi_chain(const Job& J, uint32_t K, ChunkShared& S, uint32_t wgrank, uint32_t nm, uint32_t max_rounds,
                                       ChainState& st, LaneCtr& C, uint32_t& my_pops, uint32_t& my_nnz, int* s_err) {
    Job& Jm = const_cast<Job&>(J);           // (the kernel's __shared__ Job: not const there)
    const uint32_t nwg_full = J.nwg;
    const bool sub = K < nwg_full;
    if (sub) {
        __syncthreads();
        if (threadIdx.x == 0) { Jm.nwg = K; Jm.subteam = 1; }
        __syncthreads();
    }
    const uint32_t cap_n = K * ECNE_WG * (drain_ok(J) ? 1u : 2u);
    const bool drain = drain_ok(J);
    int rc = 0;
    for (uint32_t chain = 1;; ++chain) {       // chained rounds, see multi_chain_next
        uint32_t cm = 0, ntm = st.tail, levels = 0;
#ifdef ECNE_ROUNDLOG
        const unsigned long long rl_t0 = wall_clock64();
#endif
        if (drain ? queue_round_drain(J, S, wgrank, st.head, st.tail, nm, C, my_pops, my_nnz, s_err, &cm, &ntm, &levels)
                  : queue_round_multi(J, S, wgrank, st.head, st.tail, nm, C, my_pops, my_nnz, s_err, &cm, &ntm)) { rc = 1; break; }
#ifdef ECNE_ROUNDLOG
        if (wgrank == 0 && threadIdx.x == 0) printf("RL %s avail %u n %u c %u dt %llu levels %u team %u\n", K == 1 ? "solo" : "multi", st.tail - st.head, nm, cm, wall_clock64() - rl_t0, levels, K);
#endif
        if (wgrank == 0) job_heartbeat(J);      // (the workgroups outside a sub-team wait at the job barrier for the whole chain)
        st.head += cm;
        st.tail = ntm;
        st.rounds += 1;
        st.rows += cm;
        st.sd[cm < 64 ? 0 : cm < 4096 ? 1 : 2] += 1;
        st.streak = (cm == nm && (!drain || levels <= 2 || K > 1)) ? st.streak + cm : 0;
        // (a drain round is never cut by a dependency: cm < nm means a long row the round does not take sits at rank cm -- the
        //  window stays as it is; shrinking it there cost ecdsa_like(26) a ramp of three extra rounds, 4 096 -> 16 384 -> 65 536)
        if (drain) { if (cm == nm) drain_window_update(levels, nm, cap_n, st.mwindow); }
        else multi_window_update(cm, nm, cap_n, st.mwindow, st.window);
        if (K == 1 && (cm < 2 * levels || cm < nm)) break;      // a solo drain that runs fewer than two rows per level is a chain: back to the fast rounds
        if (sub && st.tail - st.head > 2 * cap_n) break;        // the frontier has outgrown the team: the master commands a larger one
        nm = chain < max_rounds ? multi_chain_next(J, st.head, st.tail, st.window, st.mwindow, cm, nm) : 0;
        if (!nm) break;
    }
    if (sub) {
        __syncthreads();
        if (threadIdx.x == 0) { Jm.nwg = nwg_full; Jm.subteam = 0; }
        __syncthreads();
    }
    return rc;
}
// team_io (a team's master, else null): out bit 0 = a chain of rounds ran on the helpers in this phase, bit 1 = an error seen at a job barrier
// has already sent the helpers home. The helpers are NOT told that the phase is over: the caller does that (team_leave, k_solve.hip.hpp) --
// or goes on alone, with them waiting for the next command.
template <bool TEAM>
__device__ __noinline__ void queue_phase_chunked(const Job& J, QState& q, ChunkShared& S, unsigned long long* hits,
                                    unsigned long long& steps, unsigned long long& nuniq,
                                    unsigned long long& pops, unsigned long long& pop_nnz, int* s_err, uint32_t* team_io = nullptr) {
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
    // With the chain executor a sequential pop costs about a fifth of exec_row()'s, so a round has to commit
    // more rows to be worth its latency, whatever the queue length; bursts grow further while the chain lasts.
    const bool chain = chain_ok(J);
    const bool v2 = fast_wave_ok(J);
    const bool lvg_ok = TEAM && J.nwg > 1 && v2 && level_rounds_on(J) && !drain_eager(J);      // ... and on the master of a team (device-memory state)
    uint32_t lvg_hold = 0;
    const bool lv_ok = chain && v2 && level_rounds_on(J);      // level rounds: LDS-resident state, row records, the fast round's LDS block
    uint32_t lv_burst = 1;
    uint32_t crew_pen = 0, crew_hold = 0;        // hand-overs to crew rounds that came back after a few rounds (a frontier that keeps widening): level rounds keep it for a while
    bool lv_wide = false, lv_chain = false;      // lv_chain: the burst that follows pops rows the level rounds declined (chain executor)
    bool head_big = false;                       // the chain executor stopped in front of a live long row: popped alone, by the whole workgroup
    const bool v2wg = v2 && fast_wg_ok(J);
    uint32_t streak = 0;             // rows committed in a row without a dependency cutting a round short (evidence for a wide independent frontier)
    bool declined_wide = false;      // the fast wavefront round keeps declining the head row of a wide frontier
    uint32_t declined_run = 0;
    const uint32_t burst_c = chain ? ECNE_CHAIN_BURST_C : ECNE_BURST_C, burst_avail = chain ? 0xFFFFFFFFu : 64u, burst_max = chain ? 4096u : 512u;
    uint32_t mwindow = (TEAM && drain_ok(J)) ? ECNE_MWINDOW0_DRAIN : ECNE_MWINDOW0;        // window of multi-workgroup rounds (adaptive like `window`)
    // Solo drain rounds: the master of a large job drains up to 512 rows by itself (queue_round_drain on a team of one: its job
    // barriers are workgroup barriers) when fast rounds keep committing a few rows of a full window -- many dependency chains side
    // by side, each with several rows queued (45 copies of a chained circuit in one file: ~5 of 64 rows per fast round).
    const bool solo_ok = TEAM && drain_ok(J) && J.nwg > 1 && !(J.drain & 4u);       // (multi-workgroup jobs: their state lives in device memory; ECNE_SOLO=0 switches them off)
    bool solo = false;
    uint32_t solo_cool = 0;
    bool helpers_released = false;   // an error seen at a job barrier has already sent the helpers home
    bool used_team = false;
    __syncthreads();
    while (q.head != q.tail) {
        // the error word is polled every 8th round (a raised error only has to stop the solve soon)
        job_heartbeat(J);   // "the master is alive"
#ifdef ECNE_ROUNDLOG
        unsigned long long rl_t0 = wall_clock64();      // (round log builds: every line's dt runs from here -- the prints themselves stay outside)
#endif
        if ((round++ & 7u) == 0 && wg_error(J, s_err)) break;
        if (pops_total > pop_cap) { raise(J, K_ENOCONVERGE); break; }
        uint32_t avail = q.tail - q.head;
        // chain executor: a short queue is popped sequentially right away (a round over a handful of rows costs more
        // than their pops, ~0.8 us each), until the queue is empty or a frontier has built up again
        uint32_t burst_stop = 0;
        if (chain && !burst && avail <= ECNE_CHAIN_AVAIL) { burst = 1u << 20; burst_stop = 2 * ECNE_CHAIN_AVAIL; }
        // ---- level rounds (level.hip.hpp): a single-workgroup job with LDS-resident state takes every frontier of up to 192 rows
        // level by level on wavefront 0, without coming back here between two levels. It returns in front of a row it does not
        // take (the chain executor pops that one; more of them in a row: longer bursts) or when the frontier has grown wide.
        if (lv_ok && !head_big && (lv_wide || (!burst && avail <= ECNE_LV_WIDE_AVAIL))) {
            // (lv_wide: a round on the workgroup was cut short by a dependency with many rows queued -- chains side by side; the level
            //  rounds work the queue off 64 rows at a time for as many rounds as the burst would have had pops)
            const uint32_t lv_max = lv_wide ? burst : (1u << 20);
            const bool wide = lv_wide;
            lv_wide = false; if (wide) burst = 0;
            // Narrow frontiers (up to eight queued rows) go to crew rounds -- one wavefront per row, all eight wavefronts (crew.hip.hpp);
            // wider ones to level rounds on wavefront 0 until they are narrow again.
            const bool crew = crew_on(J);
            uint32_t hd = q.head, tl = q.tail, nr = 0, why = LV_FAT, ngen = 0, bigf = 0, mtop = 0;
            bool warm = false;          // the other loop left the LDS tables and the queue mirror in place
            for (;;) {
                if (crew && tl != hd && tl - hd <= ECNE_CREW_MAX) {
                    uint32_t nr1 = 0, ng1 = 0, bg1 = 0;
                    const uint32_t left = lv_max > nr + ngen ? lv_max - nr - ngen : 1u;
#ifdef ECNE_ROUNDLOG
                    const unsigned long long cl_t0 = wall_clock64(); const uint32_t cl_h0 = hd, cl_a0 = tl - hd;
#endif
                    const bool handed = warm;
                    why = crew_rounds(J, S, hd, tl, left, wide, C, my_pops, my_nnz, &nr1, &ng1, &bg1, warm, mtop);
                    // (EdDSAPoseidon's frontier goes 2, 3, 4, 4, 12, 14 rows and round again: four crew rounds per hand-over gain what the two
                    //  hand-overs cost -- after two such visits in a row the level rounds keep the frontier for their next 64 rounds)
                    if (handed && why == LV_FAT && nr1 < 8u) { if (++crew_pen >= 2u) { crew_pen = 0; crew_hold = 64; } }
                    else if (nr1 >= 8u) crew_pen = 0;
#ifdef ECNE_ROUNDLOG
                    if (tid == 0) printf("RL crew avail %u n %u c %u dt %llu why %u gen %u\n", cl_a0, nr1, hd - cl_h0, wall_clock64() - cl_t0, why, ng1);
#endif
                    nr += nr1; ngen += ng1;
                    if (bg1) bigf = 1;
                    if (why != LV_FAT) break;
                    warm = true;
                }
#ifdef ECNE_ROUNDLOG
                const unsigned long long ll_t0 = wall_clock64();
#endif
                if (w == 0) {
                    // A row the level rounds do not take (a constant row, 1 = x + y, a short binary decomposition, a bound on the limbs) is
                    // popped right here by the general executor and the level rounds go on -- the way back through the policy and the chain
                    // executor's burst costs four times the pop. A live long row goes back: the whole workgroup pops it (head_big).
                    // (Tried: those shapes inside the level round through fast_decide, inline, out of line and as a second instantiation a
                    //  job switches to -- bit-exact, and slower every time: the loop is at its register limits, 248 VGPRs and scalar
                    //  registers already spilled to lanes; Poseidon 2.9 -> 3.4 us per level, secp256k1 7.2 -> 8.0 ms.)
                    uint32_t hd1 = hd, tl1 = tl, nr0 = 0, why1, left = lv_max > nr + ngen ? lv_max - nr - ngen : 1u, gdone = 0, gnnz = 0, big = 0, mt1 = mtop;
                    bool warm1 = warm, dirty = false;      // dirty: the last call left the LDS tables as they were (LV_DECLINED, semi)
                    unsigned long long st = 0, nu = 0, ht[16];
                    for (int i = 0; i < 16; ++i) ht[i] = 0;
                    for (;;) {
                        uint32_t nr1 = 0;
                        why1 = level_rounds<true>(J, hd1, tl1, left, wide, false, C, my_pops, my_nnz, &nr1, &S.sd[0], crew && crew_hold == 0, warm1, &mt1, true, dirty);
                        warm1 = false;
                        dirty = why1 == LV_DECLINED;
                        nr0 += nr1;
                        if (why1 != LV_DECLINED || gdone >= 256u) break;
                        const uint32_t rr = J.queue[hd1 & J.qmask];
                        const bool sv = J.solved[rr] != 0;
                        // (tried: the empty pops of a live long decomposition row settled here from one wavefront walk -- secp256k1 7.19 -> 7.05 ms,
                        //  EdDSAPoseidon 5.24 -> 5.37, EdDSAMiMCSponge 15.2 -> 15.5: the level rounds then go on where rounds on the workgroup do better)
                        bool nop_ = false;          // a long decomposition all of whose terms are unique: the pop is all that happens (long_r4_done)
                        if ((J.rinfo[rr].shape & SH_BIG) && !sv) {
                            const RowInfo ri_ = J.rinfo[rr];
                            // (not while a wide frontier is worked off 256 positions at a time: the pop by the workgroup hands that frontier to the rounds on
                            //  the workgroup, which do better on it -- EdDSAMiMCSponge 10.5 -> 11.1 ms with the row settled here)
                            nop_ = !wide && (long_r4_done(J, ri_.shape, ri_.kpos, ri_.kneg, ri_.lenC, J.rec[16ull * rr + 1])
                                             || (long_r4(ri_.shape) && long_r4_idle(J, ri_.shape, J.flags[ri_.kpos], J.flags[ri_.kneg], ri_.kpos, ri_.kneg, ri_.lenC)));
                            if (!nop_) { big = 1; break; }
                        }
                        hd1++;
                        if (lane == 0) J.inq[rr] = 0;
                        wg_fence();
                        ++gdone;
                        gnnz += (J.rpA[rr + 1] - J.rpA[rr]) + (J.rpB[rr + 1] - J.rpB[rr]) + (J.rpC[rr + 1] - J.rpC[rr]);
                        if (!sv && !nop_) {
                            QState qq;
                            qq.head = hd1; qq.tail = tl1; qq.evout = nullptr; qq.nev = 0; qq.emit = 0;
                            exec_row(J, qq, rr, ht, st, nu);
                            wg_fence();
                            tl1 = qq.tail;
                        }
                        if (J.ctr->error) { why1 = LV_ROUNDS; break; }
                        if (hd1 == tl1) { why1 = LV_EMPTY; break; }
                        if (crew && crew_hold == 0 && tl1 - hd1 <= ECNE_CREW_ENTER) { why1 = LV_NARROW_COLD; break; }
                        left = left > nr1 + 1u ? left - nr1 - 1u : 1u;
                    }
                    if (dirty) lv_tables_restore(J);      // (left in front of a row for good: a live long row, an error, 256 pops, the crew's turn)
                    if (lane == 0) {
                        S.acc[0] += st; S.acc[1] += nu;
                        for (int i = 0; i < 8; ++i) S.acc[2 + i] += ht[i];
                        S.acc[10] += gdone; S.acc[11] += gnnz;
                        S.head = hd1; S.tail = tl1; S.nbig = why1; S.bl_tmp[0] = nr0; S.bl_tmp[1] = big; S.bl_tmp[2] = gdone; S.bl_tmp[3] = mt1;
                    }
                }
                __syncthreads();
#ifdef ECNE_ROUNDLOG
                if (tid == 0) printf("RL lvl avail %u n %u c %u dt %llu why %u gen %u\n", tl - hd, S.bl_tmp[0], S.head - hd, wall_clock64() - ll_t0, S.nbig, S.bl_tmp[2]);
#endif
                why = S.nbig; nr += S.bl_tmp[0]; ngen += S.bl_tmp[2]; hd = S.head; tl = S.tail; mtop = S.bl_tmp[3];
                crew_hold = crew_hold > S.bl_tmp[0] ? crew_hold - S.bl_tmp[0] : 0u;
                if (S.bl_tmp[1]) bigf = 1;
                __syncthreads();
                warm = why == LV_NARROW;
                if (why != LV_NARROW && why != LV_NARROW_COLD) break;
            }
            const uint32_t done = hd - q.head;
            if (bigf) head_big = true;
#if defined(ECNE_FINE_TICKS) && !defined(ECNE_LVPROF)
            if (tid == 0) { S.sd[0] += nr; S.sd[1] += done; S.sd[2] += wall_clock64() - qt_last; }      // schedule diagnostics: level rounds in the fast rounds' slots
#endif
#ifdef ECNE_ROUNDLOG
            if (tid == 0) { const unsigned long long d_ = wall_clock64() - rl_t0; printf("RL levelsum avail %u n %u c %u dt %llu\n", avail, nr, done, d_); }
#endif
            pops_total += done;
            hits[13] += nr;
            q.head = hd; q.tail = tl;
            if (why == LV_DECLINED && !head_big) { burst = lv_burst; lv_chain = true; if (nr < 2 && lv_burst < 64u) lv_burst *= 2; else if (nr >= 2) lv_burst = 1; }
            QTICK(6);
            if (why == LV_REFILL) lv_wide = true, burst = lv_max > nr ? lv_max - nr : 1u;      // (the mirrored part is used up: the same call again)
            if (why != LV_WIDE) continue;
            avail = q.tail - q.head;
        }
        if (lv_ok && burst > 1 && !lv_chain && !head_big) { lv_wide = true; continue; }      // what would be a burst of the chain executor: level rounds instead
        lv_chain = false;
        if (head_big) burst = 0;
        // ---- the master of a multi-workgroup job: its narrow levels (the adders of ecdsa_like between two wide frontiers, a chained
        // circuit too large for one workgroup's LDS) as level rounds on device-memory state, in place of the fast wavefront rounds
        if constexpr (TEAM) {
            if (lvg_ok && lvg_hold) --lvg_hold;
            else if (lvg_ok && !burst && !solo && !declined_wide && avail <= ECNE_LVG_WIDE_AVAIL) {
                const bool cut_exit = solo_ok && solo_cool == 0;
                if (w == 0) {
                    uint32_t hd = q.head, tl = q.tail, nr = 0;
                    const uint32_t why = level_rounds<false>(J, hd, tl, 1u << 20, false, cut_exit, C, my_pops, my_nnz, &nr, &S.sd[0]);
                    if (lane == 0) { S.head = hd; S.tail = tl; S.nbig = why; S.bl_tmp[0] = nr; }
                }
                __syncthreads();
                const uint32_t why = S.nbig, nr = S.bl_tmp[0], done = S.head - q.head;
#if defined(ECNE_FINE_TICKS) && !defined(ECNE_LVPROF)
                if (tid == 0) { S.sd[0] += nr; S.sd[1] += done; S.sd[2] += wall_clock64() - qt_last; }      // schedule diagnostics: in the fast rounds' slots
#endif
#ifdef ECNE_ROUNDLOG
                if (tid == 0) { const unsigned long long d_ = wall_clock64() - rl_t0; printf("RL level avail %u n %u c %u dt %llu\n", avail, nr, done, d_); }
#endif
                pops_total += done;
                hits[13] += nr;
                q.head = S.head; q.tail = S.tail;
                declined_run = 0;                                     // (the streak -- evidence for a wide independent frontier -- is the wide rounds' own: left alone)
                if (solo_cool) solo_cool = solo_cool > nr ? solo_cool - nr : 0u;
                __syncthreads();
                if (why == LV_DECLINED) lvg_hold = 1;                 // the row at the head: one iteration of the policy below (fast round, general executor)
                else if (why == LV_CUT) solo = true;                   // chains side by side: solo drain rounds
                else if (why == LV_WIDE) lvg_hold = 0;
                QTICK(6);
                continue;
            }
        }
        if (burst) {
            // The last chunk round committed only a handful of rows (a dependency chain): pop the next
            // `burst` rows strictly sequentially on wave 0 (cheaper per pop than a round), then look again.
            if (w == 0) {
                QState qq = q;
                qq.evout = nullptr; qq.nev = 0; qq.emit = 0;
                unsigned long long st = 0, nu = 0, ht[16], pn = 0;
                for (int i = 0; i < 16; ++i) ht[i] = 0;
                uint32_t done = 0;
                if (chain_ok(J)) {      // flags / in_queue tags in LDS, rows in one line: the chain executor
                    unsigned long long pp = 0;
                    const uint32_t at_big = chain_pops(J, qq, burst, burst_stop, ht, st, nu, pp, pn, true);
                    done = (uint32_t)pp;
                    if (lane == 0) S.flag7 = at_big;
                } else
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
                    if (!chain_ok(J)) S.flag7 = 0;
                }
            }
            __syncthreads();
            q.head = S.head; q.tail = S.tail;
            pops_total += S.nbig;
            head_big = S.flag7 != 0;
#ifdef ECNE_W2PROF
            if (tid == 0) { S.sd[3] += 1; S.sd[4] += S.nbig; S.sd[5] += wall_clock64() - qt_last; }     // (profiling builds: sequential bursts instead of general wavefront rounds)
#endif
#ifdef ECNE_ROUNDLOG
            if (tid == 0) { const unsigned long long d_ = wall_clock64() - rl_t0; printf("RL burst avail %u n %u c %u dt %llu\n", avail, S.nbig, S.nbig, d_); }
#endif
            burst = 0;
            __syncthreads();
            QTICK(6);
            continue;
        }
        if (TEAM && solo) {
            const uint32_t row0 = J.queue[q.head & J.qmask];
            const uint32_t shape0 = J.rinfo[row0].shape;
            const bool head_alone = (shape0 & SH_BIG) && !J.solved[row0] && !big_plain(shape0);     // a long row that is popped alone
            const uint32_t smin = (v2 && avail >= ECNE_V2_BACKLOG) ? 64u : multi_window_min(J);
            if (avail < 2 || head_alone || (avail >= multi_min(J) && streak >= smin)) solo = false;        // (a wide independent frontier: all workgroups)
            else {
                const uint32_t ns = avail < (uint32_t)ECNE_WG ? avail : (uint32_t)ECNE_WG;
                ChainState st;
                st.head = q.head; st.tail = q.tail; st.window = window; st.mwindow = mwindow; st.streak = streak; st.rounds = 0; st.rows = 0;
                st.sd[0] = st.sd[1] = st.sd[2] = 0;
                if constexpr (TEAM) { if (multi_chain(J, 1u, S, 0, ns, 1u, st, C, my_pops, my_nnz, s_err)) break; }
                const uint32_t cm = st.head - q.head;
                q.head = st.head;
                q.tail = st.tail;
                pops_total += cm;
                hits[13]++;
                if (tid == 0) { S.sd[3] += 1; S.sd[4] += cm; S.sd[5] += wall_clock64() - qt_last; }     // schedule diagnostics: solo drains in the "general wavefront rounds" slots
                streak = st.streak;
                // a drain that ran fewer than two rows per level (streak reset) or stopped in front of a long row: back to the fast rounds and bursts for a while
                if (cm < ns || st.streak == 0) { solo = false; solo_cool = 8; }
                QTICK(6);
                continue;
            }
        }
        // adaptive window: examining rows that end up behind the cut is wasted work, so the window
        // follows the prefix lengths actually achieved (shrinks on short prefixes, doubles on full ones)
        uint32_t n = avail < window ? avail : window;
        // with the fast wavefront round a frontier below the multi-workgroup threshold is taken 64 rows at a time
        // (with a deep backlog one uncut fast round is evidence enough: the blocks of a multiplexer -- 4 096 independent rows, then
        //  a 1 025-term sum that waits for them -- come back every 70 us on ecdsa_like(26))
        const uint32_t streak_min = (v2 && avail >= ECNE_V2_BACKLOG) ? 64u : multi_window_min(J);
        if (v2 && J.nwg > 1 && !declined_wide && (avail < multi_min(J) || streak < streak_min)) {
            const uint32_t cap_ = v2wg ? (uint32_t)ECNE_WG : 64u;        // below the multi-workgroup threshold the fast rounds take the frontier
            n = avail < cap_ ? avail : cap_;
        }
        else if (TEAM && v2 && J.nwg > 1 && !declined_wide && avail > 64 && !drain_eager(J)) {
            // A wide frontier about to go to all workgroups: a long row the rounds do not take (a binary decomposition, say) among the
            // next 64 entries would end that round in front of it after everybody has loaded and marked a whole window (ecdsa_like(26):
            // 104 296 rows examined, 16 committed, 144 us) -- the fast round takes the rows in front of it instead.
            if (w == 0) {
                const uint32_t r_ = J.queue[(q.head + (uint32_t)lane) & J.qmask];
                const uint32_t sh_ = J.rinfo[r_].shape;
                const bool alone_ = (sh_ & SH_BIG) && !J.solved[r_] && !big_plain(sh_);
                const uint64_t m_ = __ballot(alone_) & ~1ull;      // (the row at the head itself: head_alone below)
                if (lane == 0) S.bl_tmp[0] = m_ ? (uint32_t)(__ffsll((long long)m_) - 1) : 0u;
            }
            __syncthreads();
            const uint32_t p_ = S.bl_tmp[0];
            __syncthreads();
            if (p_) n = p_;
        }
        if (v2wg && J.nwg == 1 && n > ECNE_WG) n = ECNE_WG;
        const bool eager = TEAM && drain_eager(J) && avail >= 2;
        if (n <= 64 && !declined_wide && !eager && !head_big) {
            // a narrow level: the whole round on wavefront 0, no workgroup barrier inside (queue_round_wave)
            if (w == 0) {
                uint32_t nt = q.tail, nx = n;
                uint32_t cw = 0xFFFFFFFFu;
                // On the master of a multi-workgroup job narrow levels follow each other (the adders of ecdsa_like: seven rounds of
                // ~10 rows per outer iteration). A round that committed its whole window and left another narrow frontier behind
                // has only one possible sequel in the policy below -- the next fast round -- so the wavefront runs it right away,
                // up to four in a row, without the workgroup's two barriers and the policy code in between (pre_*: what the
                // rounds in front of the last one did; applied by every thread below).
                uint32_t pre_cw = 0, pre_k = 0, hd = q.head, tl = q.tail, nn = n, sc_ = solo_cool;
                for (;;) {
                    nt = tl; nx = nn;
                    // the fast round on row records first; it declines (nothing touched) what it does not cover
                    if (v2)
                        cw = chain ? queue_round_fast<true, false>(J, S, hd, tl, nn, C, my_pops, my_nnz, &nt, &nx, &S.sd[6])
                                   : queue_round_fast<false, false>(J, S, hd, tl, nn, C, my_pops, my_nnz, &nt, &nx, &S.sd[6]);
                    // (committed everything it examined -- its whole window, or up to a row / an event it does not take, bit 31 -- and the
                    //  policy below would not start solo drains on it)
                    if (TEAM && J.nwg > 1 && cw < 0xFFFFFFFEu && cw != 0 && (nx & 0x7FFFFFFFu) == cw && pre_k < ECNE_FAST_CHAIN) {
                        const uint32_t av1 = tl - hd, av2 = nt - (hd + cw);
                        const bool to_solo = sc_ == 0 && solo_ok && av1 >= ECNE_SOLO_AVAIL && ECNE_SOLO_RATIO * cw <= (av1 < 64u ? av1 : 64u);
                        if (!to_solo && av2 >= 1u && av2 <= 64u) {
#ifdef ECNE_ROUNDLOG
                            if (lane == 0) { const unsigned long long t_ = wall_clock64(); printf("RL wave avail %u n %u c %u dt %llu\n", av1, nx & 0x7FFFFFFFu, cw, t_ - rl_t0); }
                            rl_t0 = wall_clock64();      // (the print itself is not the round's time)
#endif
                            if (sc_) --sc_;
                            pre_cw += cw; ++pre_k; hd += cw; tl = nt; nn = av2;
                            continue;
                        }
                    }
                    break;
                }
                const bool fast = cw < 0xFFFFFFFEu;
                // declined at rank 0: a single-workgroup job (or a long row) takes the general wavefront round; the master of a
                // multi-workgroup job pops that one row with the general executor (narrow level) or goes to a round on all
                // workgroups (wide frontier), see below
                if (cw == 0xFFFFFFFFu || (cw == 0xFFFFFFFEu && J.nwg == 1)) cw = queue_round_wave(J, S, hd, tl, nn, C, my_pops, my_nnz, &nt, &hits[15]);
                if (lane == 0) { S.nbig = cw; S.tail = nt; S.flag7 = fast ? 1u : 0u; S.bl_tmp[0] = nx; S.bl_tmp[1] = pre_cw; S.bl_tmp[2] = pre_k; S.bl_tmp[3] = tl; }
            }
            __syncthreads();
            if (TEAM && S.bl_tmp[2]) {       // the chained rounds in front of the last one: each committed all it looked at
                const uint32_t pre_cw = S.bl_tmp[1], pre_k = S.bl_tmp[2];
                q.head += pre_cw;
                q.tail = S.bl_tmp[3];
                pops_total += pre_cw;
                hits[13] += pre_k;
                declined_run = 0;
                streak += pre_cw;
                solo_cool = solo_cool > pre_k ? solo_cool - pre_k : 0u;
                for (uint32_t k_ = 0; k_ < pre_k; ++k_) window = (window * ECNE_WGROW < ECNE_RPL * ECNE_WG) ? window * ECNE_WGROW : ECNE_RPL * ECNE_WG;
#ifdef ECNE_FINE_TICKS
                if (tid == 0) { S.sd[0] += pre_k; S.sd[1] += pre_cw; }
#endif
                avail = q.tail - q.head;
                n = avail;
            }
            const uint32_t cw = S.nbig, ntw = S.tail, nx = S.bl_tmp[0] & 0x7FFFFFFFu;   // nx: rows the round examined (the fast round may stop short of n)
            // (bit 31: the round stopped in front of something it does not take, not at a dependency -- then nx == cw, and the
            //  streak below keeps counting; tried and dropped: treating short rounds of that kind like declines, 23.9 -> 25.5 ms)
            __syncthreads();
            if (cw == 0xFFFFFFFEu) {     // the fast round declined the row at the head
                // one sequential pop of that row, then look again -- unless the head of a wide frontier keeps being declined
                // (thousands of constant rows, say): then all workgroups take it
                if (avail >= multi_min(J) && ++declined_run >= ECNE_V2_DECLINES) { declined_wide = true; declined_run = 0; }
                else burst = 1;
                continue;
            }
            if (cw != 0xFFFFFFFFu) {
#ifdef ECNE_FINE_TICKS
                if (tid == 0) {    // schedule diagnostics: fast / general wavefront rounds, rows, ticks
                    const unsigned long long dt_ = wall_clock64() - qt_last;
                    const int b_ = S.flag7 ? 0 : 3;
                    S.sd[b_] += 1; S.sd[b_ + 1] += cw; S.sd[b_ + 2] += dt_;
                }
#endif
#ifdef ECNE_ROUNDLOG
                if (tid == 0) { const unsigned long long d_ = wall_clock64() - rl_t0; printf("RL wave avail %u n %u c %u dt %llu\n", avail, nx, cw, d_); }
#endif
                q.head += cw;
                q.tail = ntw;
                pops_total += cw;
                hits[13]++;
                declined_run = 0;
                streak = cw == nx ? streak + cw : 0;
                // a dependency cut a well-filled window short: several chains side by side -- the master drains the next windows by itself
                if (solo_cool) --solo_cool;
                else if (solo_ok && avail >= ECNE_SOLO_AVAIL && ECNE_SOLO_RATIO * cw <= (avail < 64u ? avail : 64u)) solo = true;      // (whatever stopped the round: a dependency, or a row the fast round only takes at rank 0)
                // (single-workgroup jobs: a short prefix goes to the chain executor whatever stopped it -- rows the fast round
                //  does not take are cheap there; the master of a large job only bursts on true dependency chains)
                if (cw < burst_c && (chain || cw < nx) && avail < burst_avail) { burst = next_burst; if (next_burst < burst_max) next_burst *= 2; }
                if (cw == nx) window = (window * ECNE_WGROW < ECNE_RPL * ECNE_WG) ? window * ECNE_WGROW : ECNE_RPL * ECNE_WG;
                else if (cw < nx / 4) { uint32_t wn = 4 * cw; window = wn < ECNE_WMIN ? ECNE_WMIN : wn; }
                else next_burst = 16;
                QTICK(6);
                continue;
            }
            // (the window starts with a live long row: the general path below takes this round)
        }
        if (v2wg && n > 64 && n <= ECNE_WG && !declined_wide && !head_big) {
            // ---- a medium frontier: the fast round on the whole workgroup (one row per thread), see wave2.hip.hpp
            uint32_t nt = q.tail, nx = n;
            const uint32_t cw = chain ? queue_round_fast<true, true>(J, S, q.head, q.tail, n, C, my_pops, my_nnz, &nt, &nx, &S.sd[6])
                                      : queue_round_fast<false, true>(J, S, q.head, q.tail, n, C, my_pops, my_nnz, &nt, &nx, &S.sd[6]);
            nx &= 0x7FFFFFFFu;
            if (cw == 0xFFFFFFFEu) {
                if (J.nwg > 1) {
                    if (avail >= multi_min(J) && ++declined_run >= ECNE_V2_DECLINES) { declined_wide = true; declined_run = 0; }
                    else burst = 1;
                } else burst = chain ? next_burst : 1;
                continue;
            }
            if (cw != 0xFFFFFFFFu) {
#ifdef ECNE_FINE_TICKS
                if (tid == 0) { const unsigned long long dt_ = wall_clock64() - qt_last; S.sd[0] += 1; S.sd[1] += cw; S.sd[2] += dt_; }
#endif
                q.head += cw;
                q.tail = nt;
                pops_total += cw;
                hits[13]++;
                declined_run = 0;
                streak = cw == nx ? streak + cw : 0;
                if (cw < burst_c && (chain || cw < nx) && avail < burst_avail) { burst = next_burst; if (next_burst < burst_max) next_burst *= 2; }
                if (cw == nx) window = (window * ECNE_WGROW < ECNE_RPL * ECNE_WG) ? window * ECNE_WGROW : ECNE_RPL * ECNE_WG;
                else if (cw < nx / 4) { uint32_t wn = 4 * cw; window = wn < ECNE_WMIN ? ECNE_WMIN : wn; }
                else next_burst = 16;
                QTICK(6);
                continue;
            }
            // (a long row at the head: the general path below pops it alone)
        }
        const uint32_t rpl = (n + ECNE_WG - 1) / ECNE_WG;          // rows per lane this round
        const uint32_t r0 = (uint32_t)tid * rpl;                    // my first rank
        uint32_t row[ECNE_RPL], shape[ECNE_RPL], xv[ECNE_RPL];
        uint32_t live = 0, noop = 0, noop_b = 0;                    // bit s = slot s
        // (a frontier that goes to all workgroups anyway: only the row at the head matters here -- is it a long row that has to be
        //  popped alone? -- the round loads its rows itself)
        const bool want_multi = TEAM && (eager || (J.nwg > 1 && avail >= multi_min(J) && ((v2 ? streak >= streak_min : window >= multi_window_min(J)) || declined_wide)));
#pragma unroll
        for (uint32_t sl = 0; sl < ECNE_RPL; ++sl) {
            row[sl] = 0; shape[sl] = 0; xv[sl] = 0;
            // (head_big: the executor before this one stopped in front of a live long row -- only that row matters, it is popped alone below)
            if (sl < rpl && r0 + sl < n && ((!want_multi && !head_big) || (tid == 0 && sl == 0))) {
                row[sl] = J.queue[(q.head + r0 + sl) & J.qmask];
                const RowInfo ri = J.rinfo[row[sl]];
                shape[sl] = ri.shape;
                xv[sl] = ri.x;
                if (!J.solved[row[sl]]) live |= 1u << sl;
            }
        }
        // a live long row at the head is popped alone when it cannot ride along in a round (R2..R6 shapes) -- or when the
        // window is narrow anyway: a workgroup round for a handful of rows costs ten times the long row's own pop
        if (tid == 0) { S.cut = n; S.fallback = ((shape[0] & SH_BIG) && (live & 1u) && (head_big || !big_plain(shape[0]) || n <= (v2wg ? (uint32_t)ECNE_WG : 64u))) ? 1u : 0u; }
        const bool was_head_big = head_big;
        head_big = false;
#pragma unroll
        for (uint32_t sl = 0; sl < ECNE_RPL; ++sl)   // a long row that can ride along sends the round down the general path
            if (sl < rpl && r0 + sl < n && (shape[sl] & SH_BIG) && (live & (1u << sl)) && big_plain(shape[sl])) S.hasbig = 1;
        __syncthreads();
        QTICK(0);
        if (was_head_big && !S.fallback) { if (tid == 0) S.hasbig = 0; __syncthreads(); continue; }      // (not a live long row after all: the round again, with its window loaded)
        if (!S.fallback && want_multi) {
            declined_wide = false;
            // a wide frontier: a chain of rounds on a team of workgroups (multi_chain) -- all of them for a frontier that fills their
            // lanes, the first K for a narrower one (at least 8, twice what the window needs: it may double along the chain)
            uint32_t K = J.nwg;
            if (J.nwg > 16 && !(J.drain & 8u)) {
                const uint32_t per = ECNE_WG * (drain_ok(J) ? 1u : 2u);
                const uint32_t need = (avail + per - 1) / per;       // (by what is queued, not by the window: windows drain as a whole and double)
                uint32_t kt = (2 * need + 7u) & ~7u;
                if (kt < 8) kt = 8;
                if (4 * kt <= 3 * J.nwg) K = kt;
            }
            const uint32_t cap_n = K * ECNE_WG * (drain_ok(J) ? 1u : 2u);
            uint32_t nm = avail < cap_n ? avail : cap_n;
            if (nm > mwindow) nm = mwindow;
            if (tid == 0) {
                const uint32_t sg = ld_agent(&J.ctr->sub_gen) >> 1;
                bar_local().sgen = sg;
                unsigned int* const cmd = J.ctr->q_cmd[bar_local().gen & 1u];      // the block of the job barrier about to be arrived at (engine_types.hpp)
                cmd[1] = q.head; cmd[2] = q.tail; cmd[3] = nm;
                cmd[4] = window; cmd[5] = mwindow;
                cmd[6] = S.depoch;      // the master's solo drain rounds moved its mark epoch on: everybody continues from there
                cmd[7] = K; cmd[8] = sg;
                __hip_atomic_store(&cmd[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            used_team = true;
            if (job_barrier(J, s_err)) { helpers_released = true; break; }
            ChainState st;
            st.head = q.head; st.tail = q.tail; st.window = window; st.mwindow = mwindow; st.streak = streak; st.rounds = 0; st.rows = 0;
            st.sd[0] = st.sd[1] = st.sd[2] = 0;
            bool failed = false;
            if constexpr (TEAM) failed = multi_chain(J, K, S, 0, nm, ECNE_CHAIN_MAX, st, C, my_pops, my_nnz, s_err) != 0;
            q.head = st.head; q.tail = st.tail; window = st.window; mwindow = st.mwindow; streak = st.streak;
            pops_total += st.rows;
            hits[13] += st.rounds;
            hits[14] += (unsigned long long)st.rounds << 16;      // diagnostics: multi rounds in the high half
            hits[15] += st.rows << 8;                             // and the rows they committed
            if (tid == 0) { S.sd[13] += st.sd[0]; S.sd[14] += st.sd[1]; S.sd[15] += st.sd[2]; }   // schedule diagnostics: multi rounds by rows committed
            if (failed) {
                if (K < J.nwg) job_barrier(J, s_err);      // the workgroups outside the team wait for a command: meet them (they leave on the error snapshot)
                helpers_released = true;
                break;
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
            // (tried: wave 0 tests for an empty pop of a decomposition row with one walk first -- secp256k1 7.19 -> 7.43 ms: the walk and
            //  its barrier cost what the general executor's own pass over an all-unique row does)
            // (a long decomposition all of whose terms are unique: nothing happens -- long_r4_done, fastrow.hip.hpp)
            const RowInfo bri_ = J.rinfo[brow];
            // (round 5: ... or whose pivot and lowest bit are not unique and whose pivot's bounds are cut already -- long_r4_idle)
            bool idle = false;
            if (long_r4(bri_.shape) && !J.solved[brow]) {
                idle = long_r4_idle(J, bri_.shape, J.flags[bri_.kpos], J.flags[bri_.kneg], bri_.kpos, bri_.kneg, bri_.lenC);
            }
            const bool wgdone = J.solved[brow] || idle || (J.rec != nullptr && long_r4_done(J, bri_.shape, bri_.kpos, bri_.kneg, bri_.lenC, J.rec[16ull * brow + 1]))
                                || exec_big_row_wg(J, S, brow, J.bigev, &S.nbig);
            if (wgdone) {
                // done by the whole workgroup (or an already solved row: the pop is all that happens)
            } else if (w == 0) {
                const uint32_t rr = brow;
                QState qq;
                qq.head = q.head + 1; qq.tail = q.tail; qq.evout = J.bigev; qq.nev = 0; qq.emit = 1;
                unsigned long long st = 0, nu = 0, ht[16];
                for (int i = 0; i < 16; ++i) ht[i] = 0;
                if (TEAM || !exec_long_r4(J, qq, rr, bri_, ht, st, nu)) exec_row(J, qq, rr, ht, st, nu);
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
#ifdef ECNE_ROUNDLOG
            if (tid == 0) { const unsigned long long d_ = wall_clock64() - rl_t0; printf("RL alone avail %u n %u c %u dt %llu\n", avail, S.nbig, 1u, d_); }
#endif
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
            uint32_t mycut = 0xFFFFFFFFu;   // (one LDS atomic per wavefront, below)
            if ((uint32_t)tid < n && (live & 1u) && !(shape[0] & SH_BIG)) {
                const uint32_t rank = (uint32_t)tid;
                bool blocked = false;
                auto test = [&](uint32_t v, uint32_t rd, uint32_t wr) {
                    if ((rd | wr) & 1) {
                        const uint32_t m = hlook(S, v, 0);
                        if (m < rank) blocked = true; else if (m > rank && m < mycut) mycut = m;
                    }
                    if ((rd | wr) & 2) {
                        const uint32_t m = hlook(S, v, 1);
                        if (m < rank) blocked = true; else if (m > rank && m < mycut) mycut = m;
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
                if (blocked && rank < mycut) mycut = rank;
            }
            { const uint32_t wm = wave_min(mycut); if (lane == 0 && wm != 0xFFFFFFFFu) atomicMin(&S.cut, wm); }
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
                row_mark_global(J, row[sl], shape[sl], xv[sl], rank);
            }
            __syncthreads();
            if (S.bl_any) { big_rows_mark(J, S); __syncthreads(); }
            QTICK(1);
            // ---- check: blocked if an earlier rank may write state I read, or reads/writes state I may write.
            // Marks are updated with device-scope atomics (performed at L2): read them past the L1.
            uint32_t mycut = 0xFFFFFFFFu;   // (one LDS atomic per wavefront, below)
#pragma unroll
            for (uint32_t sl = 0; sl < ECNE_RPL; ++sl) {
                if (sl >= rpl || r0 + sl >= n || !(live & (1u << sl)) || (shape[sl] & SH_BIG)) continue;
                const uint32_t rank = r0 + sl;
                bool blocked = false;
                if (noop & (1u << sl)) {
                    if (noop_b & (1u << sl)) blocked = row_noop_blocked_global(J, row[sl], rank);
                } else {
                    // wmark holds the LOWEST rank that may write that state. Lower than mine: I would read
                    // (or overwrite) what an earlier row writes -> I am blocked. Higher than mine: that row
                    // would overwrite what I read -> it (and everything after it) is cut off.
                    const uint32_t m = row_check_global(J, row[sl], shape[sl], xv[sl], rank);
                    if (m < mycut) mycut = m;
                }
                if (blocked && rank < mycut) mycut = rank;
            }
            { const uint32_t wm = wave_min(mycut); if (lane == 0 && wm != 0xFFFFFFFFu) atomicMin(&S.cut, wm); }
            if (S.bl_any) big_rows_check(J, S);
            __syncthreads();
            c = S.cut;   // >= 1: rank 0 is never blocked and not big
            if (S.bl_any) big_rows_unmark(J, S);
            // ---- unmark; tag the rows being popped with their rank (in_queue bookkeeping, see resolve_pushes)
#pragma unroll
            for (uint32_t sl = 0; sl < ECNE_RPL; ++sl) {
                if (sl >= rpl || r0 + sl >= n) continue;
                if ((live & (1u << sl)) && !(shape[sl] & SH_BIG) && !(noop & (1u << sl))) row_unmark_global(J, row[sl], shape[sl], xv[sl]);
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
                else if (!(shape[sl] & SH_BIG)) { C.rank = q.head + r0 + sl; exec_row_lane(J, row[sl], J.evbuf + (size_t)(r0 + sl) * ECNE_EVCAP, nev[sl], C); }
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
#ifdef ECNE_ROUNDLOG
        if (tid == 0) { const unsigned long long d_ = wall_clock64() - rl_t0; printf("RL wg avail %u n %u c %u dt %llu\n", avail, n, c, d_); }
#endif
        q.head += c;
        q.tail = new_tail;
        pops_total += c;
        hits[13]++;
        // adaptive: a short queue with a short independent prefix is a dependency chain -> sequential
        // burst, doubling while it stays that way
        if (c < burst_c && avail < burst_avail) { burst = next_burst; if (next_burst < burst_max) next_burst *= 2; }
        if (c == n) window = (window * ECNE_WGROW < ECNE_RPL * ECNE_WG) ? window * ECNE_WGROW : ECNE_RPL * ECNE_WG;
        else if (c < n / 4) { uint32_t wn = 4 * c; window = wn < ECNE_WMIN ? ECNE_WMIN : wn; }
        else next_burst = 16;
    }
    // ---- the helper workgroups wait at the command barrier: the caller tells them that the queue phase is over (team_leave) -- or does not
    if (team_io) *team_io = (used_team ? 1u : 0u) | (helpers_released ? 2u : 0u);
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
__device__ __noinline__ void queue_phase_helper(const Job& J, ChunkShared& S, uint32_t wgrank, int* s_err) {      // (leaves on q_cmd[0] == 0: ctr->team_cmd / team_outer say what is next)
    LaneCtr C;
    C.steps = C.nuniq = 0;
    for (int i = 0; i < 8; ++i) C.hits[i] = 0;
    uint32_t my_pops = 0, my_nnz = 0;
    if (threadIdx.x < 12) S.acc[threadIdx.x] = 0;   // long rows executed by this workgroup count here
    if (threadIdx.x == 0) big_reset(S);
    __syncthreads();
    for (;;) {
        if (job_barrier(J, s_err)) break;
        // the command block of the barrier just passed (its generation's parity; bar_local().gen was moved on by thread 0 in front of the
        // barrier's closing __syncthreads): the master may be writing the other block -- the next command -- already
        const unsigned int* const cmd = J.ctr->q_cmd[(bar_local().gen - 1u) & 1u];
        if (ld_agent(&cmd[0]) == 0) break;
        uint32_t head = ld_agent(&cmd[1]), tail = ld_agent(&cmd[2]), n = ld_agent(&cmd[3]);
        uint32_t window = ld_agent(&cmd[4]), mwindow = ld_agent(&cmd[5]);
        const uint32_t K = ld_agent(&cmd[7]);
        const uint32_t depoch = ld_agent(&cmd[6]), sgen = ld_agent(&cmd[8]);
        __syncthreads();      // (every thread has read bar_local().gen before thread 0 can move it again)
        if (wgrank >= K) { if (threadIdx.x == 0) bar_local().lazy = 1; continue; }          // not on this chain's team: back to the job barrier (polling lazily), for the next command
        if (threadIdx.x == 0) { S.depoch = depoch; bar_local().sgen = sgen; }
        __syncthreads();
        ChainState st;
        st.head = head; st.tail = tail; st.window = window; st.mwindow = mwindow; st.streak = 0; st.rounds = 0; st.rows = 0;
        st.sd[0] = st.sd[1] = st.sd[2] = 0;
        if (multi_chain(J, K, S, wgrank, n, ECNE_CHAIN_MAX, st, C, my_pops, my_nnz, s_err)) {      // the master's chain, derived here (see multi_chain_next)
            if (K < J.nwg) job_barrier(J, s_err);          // (an error inside a sub-team: see the master's side)
            break;
        }
    }
    // per-lane counters -> LDS -> ONE device atomic per counter and workgroup (22 000 lanes hitting twelve
    // words of device memory serialise at the L2 for ~90 us per outer iteration otherwise)
    Counters* ctr = J.ctr;
    if (C.steps) atomicAdd(&S.acc[0], (unsigned long long)C.steps);
    if (C.nuniq) atomicAdd(&S.acc[1], (unsigned long long)C.nuniq);
    for (int i = 0; i < 8; ++i)
        if (C.hits[i]) atomicAdd(&S.acc[2 + i], (unsigned long long)C.hits[i]);
    if (my_pops) atomicAdd(&S.acc[10], (unsigned long long)my_pops);
    if (my_nnz) atomicAdd(&S.acc[11], (unsigned long long)my_nnz);
    __syncthreads();
    if (threadIdx.x < 12 && S.acc[threadIdx.x]) atomicAdd(&ctr->q_acc[threadIdx.x], S.acc[threadIdx.x]);
}

}  // namespace ecne
