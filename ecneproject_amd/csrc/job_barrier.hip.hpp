// job_barrier.hip.hpp — the XCD-hierarchical barrier of the workgroups of one job, job-wide scans.
// Part of the gfx950 device code of libecne_hip (see kernels.hip.hpp for the overview).
#pragma once
#include "schedule.hip.hpp"

namespace ecne {

// ------------------------------------------------------------------------------------ job barrier
// A job (one constraint system) is run by J.nwg co-resident workgroups: workgroup 0 (the "master")
// executes everything whose order matters (P1, P2, the queue, the decisions of P3, P5, all REQUEUEs);
// the others join for the row-parallel passes of the whole-system sweeps P3 / P4, the setup and the
// verdict count. They meet at this barrier: sense-reversing counter, agent-scope release before
// arriving (writes back this XCD's dirty L2 lines) and agent-scope acquire after leaving (drops
// stale L1/L2 lines) — per-XCD L2s are not coherent with each other on MI355X. The last arriver
// snapshots the job's error word, so every workgroup leaves with the SAME view of it and takes the
// same branch. Waits are bounded (K_ETIMEOUT).
__device__ __forceinline__ uint32_t my_xcc_id() {
    // HW_REG_XCC_ID (hwreg 20), bits [3:0]: which of the 8 XCDs this wave runs on
    return __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20) & 7u;
}

// What a workgroup remembers between job barriers (LDS): the generation it waits for next and, once the
// first barrier of the launch has established them, its XCD's member count and the number of XCDs in use
// -- so that a barrier costs one atomic per level and one polled word, no other memory round trips.
struct BarLocal { unsigned gen, members, nxcd, ready, sgen, lazy, jit; };

// -DECNE_JITTER (developer build, tools/gp_soak_jitter.sh): pseudo-random delays in front of every barrier arrival and behind every release,
// a random start delay per workgroup, the helpers held back past the master's first commands -- to shake out orderings the barrier
// protocols might depend on by accident. The sequence is a function of (seed, workgroup, call count): g_jitter_seed is set by the host
// from ECNE_JITTER_SEED before every launch. One call in eight sleeps long (up to ~60 us), the others up to ~7 us.
#ifdef ECNE_JITTER
static __device__ uint32_t g_jitter_seed;
__device__ __forceinline__ BarLocal& bar_local();
__device__ __forceinline__ void jitter(uint32_t salt) {
    if (threadIdx.x != 0) return;
    uint32_t x = (g_jitter_seed + 0x9E3779B9u * (blockIdx.x + 1u)) ^ (salt * 0x85EBCA6Bu) ^ (++bar_local().jit * 0xC2B2AE35u);
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    uint32_t n = (x & 7u) == 0 ? ((x >> 3) & 15u) : ((x >> 3) & 1u) * ((x >> 4) & 1u);      // units of s_sleep(127) ~ 3.4 us; most calls: none
    if ((x & 0x3FFu) == 1u) n = 64u + ((x >> 10) & 63u);                                        // one in a thousand: 0.2-0.4 ms
    for (uint32_t i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(127);
    if (n == 0) for (uint32_t i = 0; i < ((x >> 5) & 3u); ++i) __builtin_amdgcn_s_sleep(8);
}
#define ECNE_JIT(salt) jitter(salt)
#else
#define ECNE_JIT(salt) do { } while (0)
#endif
__device__ __forceinline__ BarLocal& bar_local() {
    __shared__ BarLocal b;
    return b;
}
// "the job is alive": thread 0 of a workgroup that works while others wait at the JOB barrier (the master between two commands, the
// master of a sub-team during its chain of rounds, the sweeps' master-only stretches) moves this word; a waiter's clock restarts
// whenever it has moved, so the barrier's time bound is on time WITHOUT progress, however long a legitimate sequential stretch is.
__device__ __forceinline__ void job_heartbeat(const Job& J) {
    if (threadIdx.x == 0 && (J.nwg > 1 || J.subteam)) __hip_atomic_fetch_add(&J.ctr->heartbeat, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void job_barrier_init() {   // thread 0, once per launch (the device words are zeroed by the host)
    BarLocal& b = bar_local();
    b.gen = 0; b.members = 0; b.nxcd = 0; b.ready = 0; b.sgen = 0; b.lazy = 0; b.jit = 0;
}

// Barrier of a SUB-TEAM: the first J.nwg workgroups of the job (J is a copy of the job with nwg = K and subteam = 1, made by
// multi_chain in rounds.hip.hpp; the others wait at the job's own barrier for the next command). Flat -- every workgroup releases
// and acquires, one counter, one generation word -- which costs what the hierarchical one costs up to ~48 workgroups. The local
// generation is set from the master's command at the start of every chain (different chains have different members).
__device__ int sub_barrier(const Job& J, int* s_err) {
    ECNE_JIT(1);
    __syncthreads();
    if (threadIdx.x == 0) {
        Counters* c = J.ctr;
        BarLocal& b = bar_local();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned g = b.sgen;
        const unsigned arrived = __hip_atomic_fetch_add(&c->sub_count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (arrived == J.nwg - 1) {
            const int e = __hip_atomic_load(&c->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&c->sub_count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&c->sub_gen, ((g + 1u) << 1) | (e != 0 ? 1u : 0u), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
        unsigned spins = 0, w;
        unsigned long long t_wait0 = wall_clock64();
        while (((w = __hip_atomic_load(&c->sub_gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 1) == g) {
            __builtin_amdgcn_s_sleep(4);
            if ((++spins & 1023u) == 0 && wall_clock64() - t_wait0 > 100000ull * (unsigned long long)J.bar_timeout_ms) {
#ifdef ECNE_JITTER
                printf("ECNE TIMEOUT sub_barrier: block %u team %u waits for gen %u, sub_gen %u sub_count %u bar_gen %u bar_count %u team_cmd %u heartbeat %u error %d\n", blockIdx.x, J.nwg, g,
                       ld_agent(&c->sub_gen), ld_agent(&c->sub_count), ld_agent(&c->bar_gen), ld_agent(&c->bar_count), ld_agent(&c->team_cmd), ld_agent(&c->heartbeat), (int)ld_agent((const uint32_t*)&c->error));
#endif
                raise(J, K_ETIMEOUT); w = 1; break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        ECNE_JIT(2);
        b.sgen = g + 1;
        *s_err = (w & 1u) ? __hip_atomic_load(&c->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
    }
    __syncthreads();
    return *s_err;
}

__device__ int job_barrier(const Job& J, int* s_err) {
    if (J.subteam && J.nwg > 1) return sub_barrier(J, s_err);
    if (J.nwg > 1) ECNE_JIT(3);
    __syncthreads();
    if (threadIdx.x == 0) {
        Counters* c = J.ctr;
        if (J.nwg == 1) {
            *s_err = __hip_atomic_load(&c->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            // XCD-hierarchical: workgroups of one XCD share its L2, so only the last of them to arrive
            // (the XCD leader) pays for the agent-scope release (L2 write-back) before arriving at
            // the top-level counter; everybody waits on one generation word and then drops its stale
            // L1 lines. The first barrier of a launch is flat and establishes the XCD membership.
            // The generation word carries the generation in its upper bits and "an error was raised" in
            // bit 0, so the waiters learn both from the one word they poll.
            BarLocal& b = bar_local();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // my stores have reached my XCD's L2
            const unsigned g = b.gen;
            const bool hier = b.ready != 0;
            bool arrive_top = true;
            const unsigned x = my_xcc_id();
            if (hier) {
                const unsigned a = __hip_atomic_fetch_add(&c->xcd_count[x][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (a == b.members - 1)
                    __hip_atomic_store(&c->xcd_count[x][0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else arrive_top = false;
            } else {
                __hip_atomic_fetch_add(&c->xcd_members[x], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (arrive_top) {
                ECNE_JIT(4);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const unsigned expect = hier ? b.nxcd : J.nwg;
                const unsigned arrived = __hip_atomic_fetch_add(&c->bar_count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (arrived == expect - 1) {
                    if (!hier) {
                        unsigned na = 0;
                        for (int i = 0; i < 8; ++i)
                            na += __hip_atomic_load(&c->xcd_members[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
                        __hip_atomic_store(&c->n_xcd_active, na, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    const int e = __hip_atomic_load(&c->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&c->bar_count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&c->bar_gen, ((g + 1u) << 1) | (e != 0 ? 1u : 0u), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            // the wait is bounded by wall-clock time (J.bar_timeout_ms of the 100 MHz counter; the host scales it with the system:
            // 0.2 s + 2 us per row): the launch was checked to fit the device as a whole (ecne_engine.hip) -- they all run unless something else
            // occupies the device (include/ecne.h: one solver process per device)
            // (helpers legitimately wait for as long as the master works alone -- a deep chain can take many milliseconds --
            //  so the clock restarts whenever the master's heartbeat word has moved: the bound is on time WITHOUT progress)
            unsigned spins = 0, w;
            unsigned long long t_wait0 = wall_clock64();
            unsigned hb = __hip_atomic_load(&c->heartbeat, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (((w = __hip_atomic_load(&c->bar_gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 1) == g) {
                // (a workgroup outside the team of the running chain of rounds backs off: hundreds of pollers take memory bandwidth and
                //  latency from the workgroups that work. Not the helpers that wait for the master's next command between chains: waking
                //  them late cost 0.2 ms per ecdsa-scale solve)
                if (!b.lazy || spins < 48) __builtin_amdgcn_s_sleep(8); else if (spins < 192) __builtin_amdgcn_s_sleep(32); else __builtin_amdgcn_s_sleep(127);
                if ((++spins & 1023u) == 0) {
                    const unsigned long long now = wall_clock64();
                    const unsigned hb2 = __hip_atomic_load(&c->heartbeat, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (hb2 != hb) { hb = hb2; t_wait0 = now; }
                    else if (now - t_wait0 > 100000ull * (unsigned long long)J.bar_timeout_ms) {
#ifdef ECNE_JITTER
                        printf("ECNE TIMEOUT job_barrier: block %u of %u xcd %u hier %d members %u nxcd %u lazy %u waits for gen %u, bar_gen %u bar_count %u xcd_count %u sub_gen %u sub_count %u team_cmd %u heartbeat %u error %d\n",
                               blockIdx.x, J.nwg, x, (int)hier, b.members, b.nxcd, b.lazy, g, ld_agent(&c->bar_gen), ld_agent(&c->bar_count), ld_agent(&c->xcd_count[x][0]), ld_agent(&c->sub_gen), ld_agent(&c->sub_count),
                               ld_agent(&c->team_cmd), hb2, (int)ld_agent((const uint32_t*)&c->error));
#endif
                        raise(J, K_ETIMEOUT); w = 1; break;
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            ECNE_JIT(5);
            b.gen = g + 1;
            b.lazy = 0;
            if (!hier) {   // the first barrier of the launch just completed: remember the XCD layout
                b.members = __hip_atomic_load(&c->xcd_members[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                b.nxcd = __hip_atomic_load(&c->n_xcd_active, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                b.ready = 1;
            }
            *s_err = (w & 1u) ? __hip_atomic_load(&c->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
        }
    }
    __syncthreads();
    return *s_err;
}

// ---- the parts of one file (Family, engine_types.hpp): one thread of each part's workgroup, once per outer iteration. Returns 1 when
// the loop goes on (some part made progress in the iteration before), 0 when it ends for everybody, 2 when a part left with an
// error or the wait ran out (the host then solves the file as one system).
__device__ __noinline__ uint32_t family_sync(const Job& J, bool progress, uint32_t outer) {
    Family* const F = J.family;
    const uint32_t slot = outer % 3u;
    if (progress) atomicOr(&F->progress[slot], 1u);
    if (outer == 0 && J.fam_rank == 0) {        // the constant wire as setup leaves it
        for (int i = 0; i < 4; ++i) { F->snap_lb[i] = J.lb[4 + i]; F->snap_ub[i] = J.ub[4 + i]; }
        for (int i = 0; i < 8; ++i) F->snap_values[i] = J.values[8 + i];
        F->snap_abz = J.abz[1]; F->snap_flags = J.flags[1]; F->snap_nvalues = J.nvalues[1];
    }
    __threadfence();
    ECNE_JIT(6);
    if (ld_agent(&F->abort)) return 2u;
    const uint32_t g = ld_agent(&F->gen);
    if (atomicAdd(&F->arrived, 1u) == J.fam_size - 1u) {
        __hip_atomic_store(&F->arrived, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&F->progress[(outer + 2u) % 3u], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (next written two barriers from now)
        __threadfence();
        atomicAdd(&F->gen, 1u);
    } else {
        const unsigned long long t0 = wall_clock64(), bound = 100000ull * 20ull * (unsigned long long)(J.bar_timeout_ms ? J.bar_timeout_ms : 200u);
        while (ld_agent(&F->gen) == g) {
            if (ld_agent(&F->abort)) return 2u;
            if (wall_clock64() - t0 > bound) { __hip_atomic_store(&F->abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return 2u; }
            __builtin_amdgcn_s_sleep(8);
        }
    }
    __threadfence();
    ECNE_JIT(7);
    return ld_agent(&F->progress[slot]) ? 1u : 0u;
}

}  // namespace ecne
