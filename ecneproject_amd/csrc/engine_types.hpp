// engine_types.hpp — data layout shared by the host layout step and the gfx950 kernels.
//
// HBM layout of one constraint system ("job"), all arrays flat, 1-based variable ids:
//   rp{A,B,C}[nC+1] u32      row pointers per part
//   col{A,B,C}[nnz] u32      variable ids; within a row part the entries are stored in the order
//                            the reference would iterate nonzeroKeys(part) (a Julia Set), so that
//                            every ordered walk of the reference is a left-to-right walk here
//   coef{A,B,C}[nnz][4] u64  canonical residues
//   csort[nnzC] u32          per row: entry positions sorted by |signed coefficient| (rule R7)
//   rinfo[nC] RowInfo        static shape of the row = which propagation rules it can ever feed
//   vals[...][4] u64         per-row constants: R2's two candidate values / R3's value, R4's bound
//   fo_ptr[nV+2], fo_rows    variable -> ascending rows containing it (variable_to_indices, :628-633)
//   rec[nC][16] u32          systems one workgroup solves: the row's lengths + up to 15 variable ids in ONE 64-byte line
//   foi[nV+2] 4 x u32        ... and per variable its fan-out inline (count + up to 3 rows, or count + offset into fo_rows)
// and the mutable solver state (SoA): flags u8 (bit0 unique, bit1 is_known, bit2 bounds==[0,1], bit3 bounds neither
// [0,1] nor the initial [0,p-1], bit4 carries a group tag i.e. abz != -1),
// abz i32, lb/ub 4xu64, nvalues u8 + values 2x4xu64, per-row inq/solved/flip bytes, FIFO ring.
#pragma once
#include <stdint.h>

namespace ecne {

#ifndef ECNE_SMALL_ROW
#define ECNE_SMALL_ROW 64   // rows with more entries than this are "long": handled by a whole workgroup
#endif
// k_classify_rows: rows with at most this many entries in C are classified by one lane each (the streaming pass); longer ones are listed
// by the layout (cls_list) and get a wavefront each (k_classify_wave)
#define ECNE_CLS_LANE 3
#ifndef ECNE_MAX_NWG
#define ECNE_MAX_NWG 248     // workgroups one system can get (q_part[][256] and the scratch sizes follow it)
#endif
#ifndef ECNE_ROWS_PER_WG
#define ECNE_ROWS_PER_WG 3072    // round 4 (level rounds on the master): 3072 -> 226 workgroups for ecdsa_like(26) 6.44 -> 6.40 ms, ecdsa_like(6) 3.65 -> 3.33; 2816: the same. With drain rounds (round 3, 4096): ecdsa_like(26) 85 workgroups 11.2 ms, 128: 11.9, 170: 10.5, 248: 10.7 (prefix rounds, round 2: 85: 22.2, 57: 22.5, 43: 23.0, 22: 26.7)
#endif
#ifndef ECNE_BIGK
#define ECNE_BIGK 8         // long rows one workgroup takes along in one round
#endif
#ifndef ECNE_CHAIN_ROWS
#define ECNE_CHAIN_ROWS 49152   // systems up to this many rows are solved by ONE workgroup with the chain executor (flags + in_queue tags in LDS)
#endif
#ifndef ECNE_ABSTRACT_DEVICE_ROWS
#define ECNE_ABSTRACT_DEVICE_ROWS 100000   // main files from this many rows on: abstraction's candidate scan runs on the GPU
#endif
#define ECNE_BIGTAB 2048    // big rows with an LDS slot for their push candidates (the rest use memory atomics directly)
#define ECNE_EVCAP 200      // REQUEUE events one small row can emit: 5 + 3 * ECNE_SMALL_ROW, rounded up
static_assert(ECNE_EVCAP >= 5 + 3 * ECNE_SMALL_ROW, "a small row can emit 5 + 3 * ECNE_SMALL_ROW REQUEUE events");
#define ECNE_CANDCAP 65536  // push candidates resolved in parallel per round; beyond: sequential fallback

// RowInfo.shape bits. "static" = depends only on coefficients/structure, computed once.
enum : uint32_t {
    SH_HAS_AB = 1u << 0,        // nzA or nzB non-empty: rules R3..R8 never run (:944-946)
    SH_C_EMPTY = 1u << 1,       // nzC empty
    SH_R2 = 1u << 2,            // C empty and (nzA u nzB) \ {1} is a single variable x (:875-942)
    SH_R2_BOUNDSERR = 1u << 3,  // C empty and no variable besides the constant: variable_states[-1] (:916)
    SH_R2_DIV0 = 1u << 4,       // x missing from A or from B: divexact by zero (:919-920)
    SH_R2_IS01 = 1u << 5,       // the two roots are {0,1}: bounds become [0,1] (:923-927)
    SH_R3 = 1u << 6,            // linear row with exactly one non-constant variable (:949-988)
    SH_R4_T = 1u << 7,          // C's values = {1,-2^0..-2^(l-2)} (:999, :1013)
    SH_R4_T2 = 1u << 8,         // C's values = {-1, 2^0..2^(l-2)}: flipped on first visit (:1000-1011)
    SH_R5 = 1u << 9,            // x == y (:1078-1146)
    SH_R6 = 1u << 10,           // 1 = x + y (:1148-1232)
    SH_R56_SWAP = 1u << 11,     // Set([k1,k2]) iterates k2 first (:1130, :1216)
    SH_P4 = 1u << 12,           // static part of the ABZ tagging test (:1427-1453)
    SH_P4_DIV0 = 1u << 13,      // A has no non-constant variable: divexact by zero (:1467)
    SH_CZERO = 1u << 14,        // C's map holds an explicit zero (incl. the one R3 inserts, :962)
    SH_R7_SORTED = 1u << 15,    // csort valid for this row
    SH_C_HAS1 = 1u << 16,       // the constant wire occurs in C with a non-zero coefficient
    SH_TOUCH1 = 1u << 17,       // the row can read/write the constant wire's bounds (R4/R5 shapes holding it)
    SH_BIG = 1u << 18,          // more than ECNE_SMALL_ROW entries: popped alone, wave-cooperatively
};

struct RowInfo {   // 32 bytes
    uint32_t shape;
    uint32_t x;      // R2 / R3 variable
    uint32_t kpos;   // linear rows: variable whose coefficient is  1 ; P4 rows: the B variable
    uint32_t kneg;   // linear rows: variable whose coefficient is -1 ; P4 rows: slope variable of A
    uint32_t k1, k2; // R5 / R6: the two variables in the reference's Dict order
    uint32_t validx; // first of two u256 slots in vals[], or 0xFFFFFFFF
    uint32_t lenC;   // l = |nzC|
};

struct Counters {   // one per job, device memory
    unsigned long long successful_steps, num_unique, pops, outer_iterations;
    unsigned long long rule_hits[16];
    unsigned long long unique_nontrivial, n_nontrivial, unique_targets, pop_nnz;
    int error;          // first ecne_status raised on the device (0 = none)
    unsigned long long err_key;   // lowest (queue position << 8 | -status) raised by a pop; all ones = none. Wins over `error`.
    unsigned int q_head, q_tail;
    unsigned int pad;
    unsigned long long phase_ticks[8];
    unsigned long long qticks[8];        // queue phase (master): head, mark, check+unmark, exec, flatten, resolve, alone+bursts, multi rounds
    unsigned long long mticks[8];        // multi-workgroup rounds: mark, check+cut, exec+scan, expand, count+scan, write
    unsigned long long sched[16];        // schedule diagnostics of the master (ecne_summary.sched)
    unsigned long long team_stat[4];     // ecne_summary.team
    // job-wide synchronisation words (zeroed before every launch)
    unsigned long long sync_steps;
    int error_snap;
    // the barrier words are polled by every waiting workgroup: keep them on a cache line of their own
    alignas(128) unsigned int bar_count;          // top level: XCD leaders arrive here
    alignas(128) unsigned int bar_gen;            // generation everybody waits on
    alignas(128) unsigned int xcd_count[8][32];   // per-XCD arrival counters, one cache line each
    unsigned int xcd_members[8];                  // workgroups of this job resident on each XCD
    unsigned int n_xcd_active, bar_ready;
    unsigned int heartbeat;                       // bumped by the master while it works alone (bounds the barrier wait, job_barrier)
    alignas(128) unsigned int sub_count;          // flat barrier of a sub-team (the first K workgroups of the job, rounds.hip.hpp multi_chain)
    alignas(128) unsigned int sub_gen;
    alignas(128) unsigned int pad_after_barrier;
    unsigned int p3_cand1, p3_nhot, p3_any, p3_fire;
    unsigned int p3_hot;   // some k >= 2 group could be complete in this pass (else nobody looks at the table)
    unsigned int p4_nfired, setup_tail;
    unsigned int p4_live;  // outer iteration in which some P4 candidate still had an untagged, non-unique b (k_solve, P4)
    // multi-workgroup queue rounds: command from the master, shared cut / totals, per-workgroup scan parts
    // q_cmd[s]: the master's command to the helpers that wait at the job barrier of GENERATION PARITY s (round 6). A helper reads the words AFTER that
    // barrier, and one that is not on the commanded team goes straight back to the next job barrier -- nothing made it read them before the master,
    // done with a short chain, wrote the NEXT command (found by the -DECNE_JITTER soak: a helper held up behind the release read the next command's
    // team size / its "queue phase over" and ran a chain nobody else ran, or left a barrier early: ECNE_ETIMEOUT). With one block per barrier parity
    // the words a helper reads behind barrier g are next written for barrier g + 2, which nobody reaches before every helper has arrived at g + 1.
    unsigned int q_cmd[2][12];      // mode (0 = queue phase over, 1 = run a chain of multi rounds), head, tail, n, window, mwindow, mark epoch, team size K, sub-team barrier generation
    unsigned int p4_tail;           // the queue tail P4's job-wide REQUEUE starts from (k_solve.hip.hpp)
    unsigned int q_cut, q_c_out, q_tail_out, q_fallback;
    unsigned int q_nhuge, q_huge[8][4];   // REQUEUE events with thousands of rows, expanded by the whole team after the expansion barrier: variable, rank, candidate base
    unsigned int d_cut[2], d_pend2[2], d_flag[2];   // drain rounds (drain.hip.hpp), by level parity: lowest demoted rank, rows left after the level, bit 0 "somebody is unstable" / bit 1 "somebody marked A"
    unsigned int q_part[2][256];
    unsigned int q_blk[2][ECNE_MAX_NWG * 8];   // per-wavefront totals of the block-order scan (team_block_scan)
    // (round 5) outer iterations the master of a team runs alone (k_solve.hip.hpp): whether the loop goes on (read by the helpers behind the
    // barrier at the loop top), what the helpers do when the master lets them out of the queue phase, and the state of P3's group table
    unsigned int sync_go;
    unsigned int team_cmd, team_outer;   // team_cmd: TEAM_FULL = P3 and P4 of iteration team_outer with everybody, TEAM_EXIT = the loop has ended
    unsigned int p3_tbl;                 // 0 = P3's group table is empty; 1 = it holds the counts of the last pass (kept for the incremental passes)
    unsigned long long q_acc[16];   // helpers' counter deltas: steps, nuniq, hits[0..7], pops, pop_nnz, rounds   // 100 MHz wall clock: 0 setup, 1 P1+P2+queue, 2 P3, 3 P4, 4 P5, 5 verdict; 6 = P3 rounds
};

// One file solved as several independent parts (ecne_engine.hip, SplitPlan): the parts -- single-workgroup jobs of one launch -- go
// through the outer loop (:706-1556) in lockstep, as the one loop of the whole file does: an iteration takes place for all of
// them when any of them made progress in the one before. One block per family, device memory, zeroed before every launch.
struct Family {
    alignas(128) unsigned int arrived;
    alignas(128) unsigned int gen;
    alignas(128) unsigned int progress[3];        // by outer iteration mod 3: somebody's successful_steps moved
    unsigned int abort;                           // a part left with an error (or waited too long): the others leave as well
    unsigned int var1_bad;                        // some part changed the state of the constant wire (the one variable the parts share)
    // the constant wire's state after setup (part 0), what every part's copy is compared with afterwards
    alignas(128) unsigned long long snap_lb[4], snap_ub[4], snap_values[8];
    int snap_abz;
    unsigned int snap_flags, snap_nvalues;
};

struct Job {
    // sizes
    uint32_t nC, nV, nSp, nKnown, nTarget, nP4, nP5, qmask, htmask, secp_solve, queue_mode, hotcap;
    uint32_t bar_timeout_ms; // how long the workgroups of the job wait at their barrier WITHOUT progress before they give up (K_ETIMEOUT)
    uint32_t warm_bytes;     // bytes of static arrays (from rpA on) a single-workgroup job streams once to warm its XCD's L2; 0 = off
    uint32_t lds_bytes;      // dynamic LDS of the launch: a single-workgroup job keeps as much of its mutable state there as fits (k_solve)
    uint32_t nwg, nBigCls;   // nBigCls: rows with more than 8 entries in C (classified one wavefront each)   // workgroups cooperating on this job (1 = the master alone)
    // static system
    const uint32_t *rpA, *rpB, *rpC;
    const uint32_t *colA, *colB, *colC;
    const uint64_t *coefA, *coefB, *coefC;
    uint32_t* csort;
    RowInfo* rinfo;
    uint64_t* vals;
    const uint32_t *fo_ptr, *fo_rows;
    // chain executor (chain.hip.hpp), single-workgroup jobs only (else null): rec[16 * row] = lenA | lenB << 8 | lenC << 16 |
    // 1 << 24 followed by the row's variable ids (A, B, C; stored order), word 0 = 0 for rows with more than 15 entries;
    // foi[4 * v] = {n, r0, r1, r2} for n <= 3 rows of variable_to_indices[v], else {n, offset into fo_rows, 0, 0}
    const uint32_t *rec, *foi;
    uint32_t lds_w2_off, lds_w2b_off;   // byte offsets of the fast wavefront / workgroup round's tables (wave2.hip.hpp) in the dynamic LDS, or 0xFFFFFFFF
    uint32_t lds_flags_off, lds_inq_off, lds_flip_off;   // byte offsets of flags / inq / flip3 in the dynamic LDS when resident there, else 0xFFFFFFFF (set by k_solve)
    const uint32_t *sp_in_ptr, *sp_in, *sp_out_ptr, *sp_out;
    const uint8_t* sp_kind;   // 1 = "BigMultModP", 2 = "BigLessThan", 0 = anything else (:751, :755)
    const uint32_t* k1_list;  // indices of the "BigMultModP" specials, ascending
    const uint32_t* k2_list;  // indices of the "BigLessThan" specials, ascending
    uint32_t nK1, nK2;
    const uint32_t *knowns, *targets;
    const uint8_t* nontrivial;
    const uint32_t* p4_list;
    const uint32_t* p4_b;      // per P4 row: the B variable
    const uint32_t* p4_s;      // per P4 row: slope variable of A; bit 31 set = none (divexact by zero, :1467)
    const uint32_t* cls_list;   // rows with lenC > ECNE_CLS_LANE (3), ascending: a wavefront each in k_classify_wave
    uint32_t* cls_defer;        // k_classify_rows: [0] = how many short rows its lanes deferred (a divisor other than +-1: one lane each in k_classify_wave, with the inversion), then their ids (classify.hip.hpp)
    const uint32_t *p5_rows, *p5_y;
    // mutable state
    uint8_t* flags;
    int32_t* abz;
    uint64_t *lb, *ub;
    uint8_t* nvalues;
    uint64_t* values;
    uint16_t* inq;   // 0 = not queued, 1 = queued, r+2 = being popped at rank r of the current chunk
    uint8_t *solved, *flip3;
    uint32_t* queue;
    // scratch
    uint32_t* varmin;
    const uint16_t* tbig;      // per row: 0, or 1 + index into bigrows[] (rows with > ECNE_SMALL_ROW entries, first ECNE_BIGTAB of them)
    const uint32_t* bigrows;
    uint32_t nBigRows;
    uint32_t* ht_list;         // group-table slots created in the current P3 sweep, one region per workgroup
    uint8_t* rdead;            // bit 0: row has no non-unique variable left (monotone), the sweeps skip it; bit 1: long row
    const uint32_t* long_list; // every row with more than ECNE_SMALL_ROW entries, ascending
    uint32_t nLong;
    uint8_t* p3k;
    uint32_t* p3stamp;         // per row: the outer iteration whose incremental P3 / P4 pass has looked at the row (a row popped twice is looked at once)
    uint64_t *p3h, *p3h2;
    uint64_t *ht_key, *ht_key2;
    uint32_t *ht_new, *ht_frozen;
    uint32_t* hot;
    uint8_t* fired;
    uint32_t* events;
    // per variable: lowest chunk rank that may WRITE its U-class state (unique, is_known bits:
    // monotone, final once both are set) and its B-class state (lb, ub, values, abz)
    uint32_t *wmarkU, *wmarkB;
    uint32_t* dmk[6];          // drain rounds (drain.hip.hpp): epoch-keyed mark planes X / A / C, U and B class each; multi-workgroup jobs only
    uint32_t lv_off;           // bit 0: no level rounds (level.hip.hpp) on this job: ECNE_LEVEL=0; bit 1: no crew rounds (crew.hip.hpp): ECNE_CREW=0; bit 2: long_r4_done off: ECNE_R4DONE=0 -- A/B runs
    uint32_t subteam;          // device-side copies only: 1 = this Job stands for the first `nwg` workgroups of the job, which meet at the sub-team barrier
    uint32_t drain;            // bit 0: rounds on all workgroups are drain rounds (0: prefix rounds, queue_round_multi); bit 1: test hook, every frontier is drained; bit 2: no solo drains
    uint32_t* best;            // per row: lowest candidate index that wants to push it
    uint32_t* prank;           // per row: its rank while it is being popped in a multi-workgroup round
    uint32_t* evbuf;           // per chunk rank: REQUEUE events emitted by the row popped there
    uint32_t* evcnt;           // per chunk rank of a round on a team: how many (dense; bit 31 = a long row's events, in the pool)
    uint32_t *fvar, *frank, *fbase;   // flat event list of one resolution round: variable, rank, candidate base
    uint32_t* bigev;           // events of a big row popped alone
    uint32_t* bigpool;         // events (+ candidate offsets) of long rows executed inside rounds: ECNE_MAX_NWG * ECNE_BIGK slots
    uint32_t bigstride;        // u32 words per pool slot (2 * (longest C part + 8))
    uint32_t* cand;            // per push candidate: target row | eligibility bit
    uint32_t candcap;          // capacity of cand[] (single-workgroup rounds use the first ECNE_CANDCAP)
    // rows / specials that name a variable id above num_variables (malformed input): tables for the reference's lazy BoundsError
    // (oob.hip.hpp), or nullptr. Such a system is solved by one workgroup with strictly sequential pops.
    const uint32_t* oob;
    // one of the independent parts of a file (see Family), or nullptr
    Family* family;
    uint32_t fam_rank, fam_size;
    Counters* ctr;
};

}  // namespace ecne
