// Developer microbenchmark (not part of the product): what one barrier between the workgroups of a job costs on MI355X.
//   A  the engine's barrier: XCD-hierarchical, agent-scope release (L2 write-back) by each XCD's last arriver, agent-scope
//      acquire (L1 + L2 invalidate) by everybody                                      -- job_barrier(), job_barrier.hip.hpp
//   B  flat: every workgroup releases, arrives at one counter, acquires
//   C  only the workgroups that sit on workgroup 0's XCD take part: they share one L2, so stores only have to be
//      acknowledged (s_waitcnt) and readers drop their L1 (buffer_inv sc0); no L2 write-back / invalidate
// Each round every workgroup stamps a word and, after the barrier, checks its left neighbour's stamp (stale reads counted).
// hipcc --offload-arch=gfx950 -O3 -o /tmp/bar tools/micro/bar.hip && /tmp/bar [nwg]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

struct Bar { unsigned count, gen, xcd_count[8][16], xcd_members[8], n_active, team_count, team_gen, team_n, team_ids[128]; };

__device__ __forceinline__ unsigned xcc() { return __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20) & 7u; }

__device__ void barrier_flat(Bar* b, unsigned nwg, unsigned& gen) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        const unsigned a = __hip_atomic_fetch_add(&b->count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (a == nwg - 1) {
            __hip_atomic_store(&b->count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&b->gen, gen + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
        while (__hip_atomic_load(&b->gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gen) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    ++gen;
    __syncthreads();
}

__device__ void barrier_hier(Bar* b, unsigned members, unsigned nxcd, unsigned& gen) {
    __syncthreads();
    if (threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned x = xcc();
        const unsigned a = __hip_atomic_fetch_add(&b->xcd_count[x][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (a == members - 1) {
            __hip_atomic_store(&b->xcd_count[x][0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned t = __hip_atomic_fetch_add(&b->count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t == nxcd - 1) {
                __hip_atomic_store(&b->count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&b->gen, gen + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        while (__hip_atomic_load(&b->gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gen) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    ++gen;
    __syncthreads();
}

// same-XCD team: stores acknowledged by the shared L2, readers drop their L1
__device__ void barrier_team(Bar* b, unsigned n, unsigned& gen) {
    __syncthreads();
    if (threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned a = __hip_atomic_fetch_add(&b->team_count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (a == n - 1) {
            __hip_atomic_store(&b->team_count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&b->team_gen, gen + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        while (__hip_atomic_load(&b->team_gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gen) __builtin_amdgcn_s_sleep(1);
    }
    ++gen;
    __syncthreads();
    asm volatile("buffer_inv sc1" ::: "memory");
}

template <int MODE>
__global__ void __launch_bounds__(512) k(Bar* b, unsigned* stamps, unsigned* payload, int rounds, unsigned long long* res) {
    const unsigned nwg = gridDim.x, wg = blockIdx.x;
    __shared__ unsigned s_members, s_nxcd, s_team, s_rank, s_left;
    unsigned gen = 0;
    // first barrier (flat): XCD membership
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(&b->xcd_members[xcc()], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (xcc() == 0) { const unsigned r = __hip_atomic_fetch_add(&b->team_n, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); b->team_ids[r] = wg; s_rank = r; }
    }
    barrier_flat(b, nwg, gen);
    if (threadIdx.x == 0) {
        s_members = b->xcd_members[xcc()];
        unsigned na = 0;
        for (int i = 0; i < 8; ++i) na += b->xcd_members[i] != 0;
        s_nxcd = na;
        s_team = __hip_atomic_load(&b->team_n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const bool in_team = xcc() == 0;
    if (MODE == 2 && !in_team) return;
    if (MODE == 2 && threadIdx.x == 0) s_left = b->team_ids[(s_rank + s_team - 1) % s_team];
    if (MODE != 2 && threadIdx.x == 0) s_left = (wg + nwg - 1) % nwg;
    __syncthreads();
    const unsigned left = s_left;
    unsigned tgen = 0, stale = 0;
    const unsigned long long t0 = wall_clock64();
    for (int r = 1; r <= rounds; ++r) {
        payload[(size_t)wg * 512 + threadIdx.x] = (unsigned)r;         // every thread dirties a word
        if (threadIdx.x == 0) stamps[wg * 16] = (unsigned)r;
        if (MODE == 0) barrier_hier(b, s_members, s_nxcd, gen);
        if (MODE == 1) barrier_flat(b, nwg, gen);
        if (MODE == 2) barrier_team(b, s_team, tgen);
        if (payload[(size_t)left * 512 + threadIdx.x] != (unsigned)r) ++stale;   // plain load
        if (stamps[left * 16] != (unsigned)r) ++stale;
        // second barrier: nobody overwrites before everybody has read
        if (MODE == 0) barrier_hier(b, s_members, s_nxcd, gen);
        if (MODE == 1) barrier_flat(b, nwg, gen);
        if (MODE == 2) barrier_team(b, s_team, tgen);
    }
    const unsigned long long t1 = wall_clock64();
    if (stale) atomicAdd(&res[1], (unsigned long long)stale);
    if (wg == 0 && threadIdx.x == 0) { res[0] = t1 - t0; res[2] = MODE == 2 ? s_team : nwg; }
}

int main(int argc, char** argv) {
    const int nwg = argc > 1 ? atoi(argv[1]) : 43, rounds = 2000;
    Bar* b; hipMalloc(&b, sizeof(Bar));
    unsigned *stamps, *payload; hipMalloc(&stamps, 4096 * 64); hipMalloc(&payload, 512ull * 4 * 1024);
    unsigned long long* res; hipMalloc(&res, 64);
    const char* names[] = {"A hierarchical (engine)", "B flat agent-scope", "C same-XCD team"};
    for (int m = 0; m < 3; ++m) {
        for (int rep = 0; rep < 2; ++rep) {
            hipMemset(b, 0, sizeof(Bar)); hipMemset(res, 0, 64); hipMemset(stamps, 0, 4096 * 64); hipMemset(payload, 0, 512ull * 4 * 1024);
            if (m == 0) k<0><<<nwg, 512>>>(b, stamps, payload, rounds, res);
            if (m == 1) k<1><<<nwg, 512>>>(b, stamps, payload, rounds, res);
            if (m == 2) k<2><<<nwg, 512>>>(b, stamps, payload, rounds, res);
            hipError_t e = hipDeviceSynchronize();
            if (e != hipSuccess) { printf("error %s\n", hipGetErrorString(e)); return 1; }
        }
        unsigned long long r[3]; hipMemcpy(r, res, 24, hipMemcpyDeviceToHost);
        printf("%-28s %3llu workgroups: %.2f us per barrier (store + barrier + check + barrier = 2 barriers per round), stale reads %llu\n",
               names[m], r[2], r[0] * 0.01 / (2.0 * rounds), r[1]);
    }
    return 0;
}
