// ecne_engine.hip — libecne_hip: C ABI (include/ecne.h) + host orchestration of the gfx950 kernels.
//
// What runs where
//   host (this file, host_model.hpp): .r1cs parsing, trusted-function abstraction, and the one-time
//        layout of a constraint system into flat HBM arrays (row entries stored in the order the
//        reference would visit them, variable->rows fan-out, static candidate lists).
//   device (kernels.hip.hpp): k_classify_rows (row shapes + field constants) and k_solve, the
//        whole SolveConstraintsSymbolic fixed point (reference src/R1CSConstraintSolver.jl:583-1646).
// There is no CPU implementation of the propagation rules in this library: without a HIP device
// every solve entry point returns ECNE_ENODEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <atomic>
#include <mutex>
#include <shared_mutex>
#include <new>
#include <array>
#include <map>
#include <set>
#include <queue>
#include <string>
#include <vector>

#include "../../include/ecne.h"
#include "engine_types.hpp"
#include "fp256.hpp"
#include "frontend.hpp"
#include "host_model.hpp"
#include "jlorder.hpp"
#include "kernels.hip.hpp"

using namespace ecne;

struct ecne_r1cs {
    std::shared_ptr<R1CSFile> file = std::make_shared<R1CSFile>();   // shared with the systems made from it
    R1CSFile& f = *file;
    // the same rows resident on a device (device front-end, frontend.hpp): what ecne_r1cs_load produced when the file went through
    // fe::parse_on_device; the host rows of `file` are then fetched only when somebody asks for them (ensure_host_rows)
    std::shared_ptr<fe::DevRows> drows;
};

// Which front-end a file / system goes through: 0 = host (host_model.hpp), 1 = device whenever a HIP device is there,
// 2 = auto (device for files of ECNE_FRONTEND_DEVICE_ROWS rows and more). ECNE_FRONTEND=host|device|auto, ecne_set_frontend().
#ifndef ECNE_FRONTEND_DEVICE_ROWS
#define ECNE_FRONTEND_DEVICE_ROWS 100000
#endif
static std::atomic<int>& frontend_setting() {
    static std::atomic<int> m{[] {
        const char* e = std::getenv("ECNE_FRONTEND");
        if (!e) return 2;
        return !std::strcmp(e, "host") ? 0 : !std::strcmp(e, "device") ? 1 : 2;
    }()};
    return m;
}
static bool frontend_wants_device(uint64_t n_rows) {
    const int m = frontend_setting().load(std::memory_order_relaxed);
    if (m == 0) return false;
    if (m == 2 && n_rows < ECNE_FRONTEND_DEVICE_ROWS) return false;
    return ecne_device_count() > 0;
}
// A device front-end step that failed for want of device memory or capacity (the parse needs 5-6x the file size in temporaries,
// the layout a sort scratch) is not an error of the INPUT: the host front-end takes over, as it does for FE_FALLBACK. What is
// propagated: the reference's own exceptions (format, KeyError) -- a genuinely faulted device makes the solve fail later anyway.
static bool fe_falls_back(int rc) {
    if (rc == fe::FE_FALLBACK) return true;
    if (rc == K_ENODEVICE || rc == K_ECAPACITY) { (void)hipGetLastError(); return true; }
    return false;
}
// every entry point that switches the HIP device leaves the caller's current device as it found it (include/ecne.h)
struct RestoreDevice {
    int prev = -1;
    RestoreDevice() { if (hipGetDevice(&prev) != hipSuccess) prev = -1; }
    ~RestoreDevice() { if (prev >= 0) (void)hipSetDevice(prev); }
    RestoreDevice(const RestoreDevice&) = delete;
    RestoreDevice& operator=(const RestoreDevice&) = delete;
};
static int current_device() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess) d = 0;
    return d;
}
// ECNE_FE_DEBUG=1: wall-clock of the steps of an entry point on stderr (developer aid)
struct AbiTick {
    bool on;
    std::chrono::steady_clock::time_point t;
    AbiTick() : on(std::getenv("ECNE_FE_DEBUG") != nullptr), t(std::chrono::steady_clock::now()) {}
    void operator()(const char* label) {
        if (!on) return;
        const auto n = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[abi] %-40s %8.3f ms\n", label, std::chrono::duration<double, std::milli>(n - t).count());
        t = n;
    }
};
// timing of the calling thread's last trip through the device front-end (ecne_frontend_stats)
struct FrontendStats { fe::ParseStats parse; fe::AbstractDevStats abs; double layout_ms = 0; int parse_dev = 0, abs_dev = 0, layout_dev = 0; };
static thread_local FrontendStats g_fe_stats;

// host rows of a file that was parsed on the device only (the file object is shared with the systems made from it: one lock for
// every lazy fetch of host rows, here and in sys_host_rows)
static std::mutex g_lazy_rows_mu;
static int ensure_host_rows(const ecne_r1cs* h) {
    R1CSFile& f = const_cast<ecne_r1cs*>(h)->f;
    std::lock_guard<std::mutex> g(g_lazy_rows_mu);
    if (f.host_rows) return K_OK;
    // always from the rows the solve runs on (the device-resident ones): a file replaced on disk since the load must not give the host
    // side -- abstraction patterns, the tables of ids above num_variables, split plans -- other rows than the device has
    if (h->drows) {
        const int rc = fe::download_rows(*h->drows, f.rows);
        if (rc != K_OK) return rc;
        f.host_rows = true;
        return K_OK;
    }
    return K_EINVAL;
}

// Host image of the flat arrays (built once per system, uploaded once per device)
struct Layout {
    uint32_t nC = 0, nV = 0;
    // (the big ones are sized once and filled by the host worker threads: no zero fill, huge-page advised)
    std::vector<uint32_t, RawAlloc<uint32_t>> rp[3], col[3];
    std::vector<uint64_t, RawAlloc<uint64_t>> coef[3];
    std::vector<RowInfo, RawAlloc<RowInfo>> rinfo;
    uint32_t n_vals = 0;
    std::vector<uint32_t> fo_ptr;
    std::vector<uint32_t, RawAlloc<uint32_t>> fo_rows;
    std::vector<uint32_t> sp_in_ptr, sp_in, sp_out_ptr, sp_out;
    std::vector<uint8_t> sp_kind;
    std::vector<uint32_t> knowns, targets;
    std::vector<uint8_t> nontrivial;
    std::vector<uint32_t> p4_list, p5_rows, p5_y, cls_list;
    uint64_t nnz[3] = {0, 0, 0};
    uint64_t stream_bytes = 0;   // bytes k_classify_rows reads + writes (roofline numerator)
    // a system laid out by the device front-end keeps its big arrays (rp / col / coef / rinfo / fan-out / lists) on the device
    // only; the counts the arena is carved from and the small lists above are filled either way
    bool host_arrays = false;    // rp, col, coef, rinfo hold data on the host
    uint32_t n_p4 = 0, n_p5 = 0, n_cls = 0, n_long = 0, n_bigrows = 0, fo_total = 0, maxrowC = 0;
    // secp_solve's dsu setup (:634-678) raises BoundsError on this system (`l[1]` / `l[2]` of a two-entry C whose non-zero keys are
    // none or the constant alone, :650-656; or a root outside the dsu when ids above num_variables occur)
    bool dsu_err = false;
    // ids above num_variables in a row (non-zero coefficient) or a special: tables for the reference's lazy BoundsError (oob.hip.hpp)
    std::vector<uint32_t> oob_blob;
};

struct DeviceImage {
    int device = -1;
    std::unique_ptr<fe::LayoutDev> lay;   // static arrays built by the device front-end (else they sit in `arena`, uploaded from the host Layout)
    void* arena = nullptr;
    size_t arena_bytes = 0;
    Job job;          // host copy with device pointers
    bool classified = false;
    double classify_ms = 0;
};

// ---- One file, several independent parts. A file whose rows fall into groups that share no variable but the constant wire (N copies
// of a circuit written into one file, sub-circuits that never meet) is N independent fixed points: the reference's one FIFO,
// restricted to the rows of a group, IS that group's own FIFO (a pop only pushes rows of its own group), the batch phases P3 / P4 /
// P5 act inside a group, and the outer loop runs while ANY group makes progress. The plan renumbers every group (or bin of groups)
// into a system of its own -- variables and rows ascending, orders that depend on ids or on a row's neighbour taken from the file's
// own ids (orig_var / orig_row) -- and the parts are solved as single-workgroup jobs of one launch, their outer loops in lockstep
// (Family, engine_types.hpp), each with its state LDS-resident where the file as one system was a team on device-memory state
// (45 x EdDSAMiMCSponge in one file: 178 ms as one system). Their states are scattered back into the file's own arrays on the device
// (k_scatter_part), so results, digests and bad rows read as after any solve. Anything unusual -- an error in a part, a part that
// changed the constant wire's state -- and the file is solved again as one system. ECNE_SPLIT: 0 never, 1 (default) from the second
// solve of a system on, when the first took long enough to pay for the plan, 2 at the first solve.
struct ecne_system;
struct SplitPlan {
    bool ok = false;
    int device = -1;
    std::vector<ecne_system*> kids;      // owned
    std::vector<uint32_t*> d_map;        // per part: its variable -> the file's (device, n_vars + 1 words)
    Family* d_family = nullptr;
    double plan_ms = 0;
    uint32_t n_groups = 0;
    ~SplitPlan();
};
static uint64_t next_system_uid() { static std::atomic<uint64_t> n{1}; return n.fetch_add(1, std::memory_order_relaxed); }
struct ecne_system {
    // dictionary order, current rows: the parsed file's own arrays until the first abstraction (no copy;
    // `base` keeps them alive after ecne_r1cs_free), `reduced` afterwards. A system that came through the device front-end
    // holds the rows on the device (`drows`) and fetches the host copy only when somebody needs it (sys_host_rows).
    std::shared_ptr<R1CSFile> base;
    std::shared_ptr<fe::DevRows> drows;
    Rows reduced;
    const Rows* cur = nullptr;
    const Rows& rows() const { return *cur; }      // after sys_host_rows()
    uint64_t n_rows() const { return cur ? cur->n() : drows ? drows->n : 0; }
    std::vector<Special> specials;
    std::vector<int64_t> knowns, targets;
    int64_t n_vars = 0, n_rows_main = 0;
    uint32_t screen[4] = {0, 0, 0, 0};      // build_split's device screen (split_screen)
    bool screen_done = false;
    double screen_ms = 0;
    int secp_solve_override = -1;      // ecne_system_set_secp_solve: this system's kwarg secp_solve inside a batch (-1: the launch's opts decide)
    bool laid_out = false;
    uint64_t generation = 0;   // bumped by every solve
    const uint64_t uid = next_system_uid();   // process-unique: a result made from a freed system never matches a new one at the same address
    Layout L;
    DeviceImage dev;
    std::vector<int64_t> order_buf;   // ecne_system_report_order
    // ---- one file as several independent parts (SplitPlan below). A part is a system of its own whose variables and rows are
    // renumbered, both ascending: orig_var / orig_row give the file's ids back wherever the reference's order depends on them (the
    // hash order of Set / Dict iteration, build_layout) or on a row's neighbour (P5's row pairs).
    std::vector<uint32_t> orig_var, orig_row;
    std::unique_ptr<SplitPlan> split;
    bool split_tried = false;
    double last_kernel_ms = 0;
    ~ecne_system() {
        if (dev.arena) {
            int prev = -1;
            (void)hipGetDevice(&prev);
            (void)hipSetDevice(dev.device);
            (void)hipFree(dev.arena);
            if (prev >= 0) (void)hipSetDevice(prev);
        }
    }
};
SplitPlan::~SplitPlan() {
    int prev = -1;
    (void)hipGetDevice(&prev);
    if (device >= 0) (void)hipSetDevice(device);
    for (uint32_t* m : d_map) if (m) (void)hipFree(m);
    if (d_family) (void)hipFree(d_family);
    for (ecne_system* k : kids) delete k;
    if (prev >= 0) (void)hipSetDevice(prev);
}
// host copy of the system's current rows (lazily: a device-front-end system downloads them on first use)
static int sys_host_rows(ecne_system& S) {
    if (S.cur) return K_OK;
    std::lock_guard<std::mutex> g(g_lazy_rows_mu);
    if (S.base) {
        if (!S.base->host_rows) {
            if (!S.drows) return K_EINVAL;
            const int rc = fe::download_rows(*S.drows, S.base->rows);
            if (rc != K_OK) return rc;
            S.base->host_rows = true;
        }
        S.cur = &S.base->rows;
        return K_OK;
    }
    if (!S.drows) return K_EINVAL;
    const int rc = fe::download_rows(*S.drows, S.reduced);
    if (rc != K_OK) return rc;
    S.cur = &S.reduced;
    return K_OK;
}

static void system_changed_rows(ecne_system* sys);
struct ecne_result {
    ecne_summary sum;
    // per-variable state is downloaded from HBM on first request (ecne_result_states /
    // ecne_result_bad_rows), and only while `sys` has not been solved again (generation check)
    ecne_system* sys = nullptr;
    uint64_t generation = 0, sys_uid = 0;
    bool have_states = false;
    std::vector<uint8_t> flags, nvalues;
    std::vector<uint64_t> lb, ub, values;
    std::vector<int32_t> abz;
    std::vector<int64_t> bad_rows;
};

// Live systems: a result fetches its per-variable state lazily from its system's device image, so it has to know
// whether that system still exists (ecne_system_free may come first).
static std::mutex g_live_mu;
static std::set<const ecne_system*>& live_systems() { static std::set<const ecne_system*> s; return s; }
static bool system_is_live(const ecne_system* s) { std::lock_guard<std::mutex> g(g_live_mu); return live_systems().count(s) != 0; }

// nothing throws across the C ABI
template <class F>
static int guarded(F&& f) {
    try { return f(); }
    catch (const std::bad_alloc&) { return ECNE_ECAPACITY; }
    catch (...) { return ECNE_EINVAL; }
}

// One multi-workgroup solve at a time per device inside this process: its workgroups meet at a hand-rolled barrier and
// have to be resident together (two such launches on one device could each hold CUs the other one waits for).
// (single-workgroup launches share the device with each other; a launch that holds a multi-workgroup job, or a batch, has it alone)
static std::shared_mutex& device_launch_mutex(int device) { static std::shared_mutex m[64]; return m[(unsigned)device & 63u]; }

// What the host reads of a job's Counters after a launch: everything in front of the synchronisation words. One block per job copies
// that part into one contiguous buffer, so that a batch costs ONE device-to-host copy (67 blocking copies were 0.8 ms per pass of the
// circomlib suite, 4 % of it).
#define ECNE_RESULT_BYTES (offsetof(Counters, sync_steps))
static_assert(ECNE_RESULT_BYTES % 4 == 0 && ECNE_RESULT_BYTES <= 1024, "the result part of Counters is gathered 4 bytes per thread by 256 threads");
// ... and before a launch every job's Counters are zeroed by one block each (504 hipMemsetAsync calls were 2 ms of host time per pass of
// `bench.py --workload many`)
__global__ __launch_bounds__(256) void k_zero_counters(const Job* jobs) {
    uint32_t* dst = reinterpret_cast<uint32_t*>(jobs[blockIdx.x].ctr);
    for (uint32_t i = threadIdx.x; i < sizeof(Counters) / 4; i += 256) dst[i] = 0;
}
static_assert(sizeof(Counters) % 4 == 0, "Counters are zeroed four bytes per thread");
__global__ __launch_bounds__(256) void k_gather_results(const Job* jobs, unsigned char* out) {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(jobs[blockIdx.x].ctr);
    uint32_t* dst = reinterpret_cast<uint32_t*>(out + (size_t)blockIdx.x * ECNE_RESULT_BYTES);
    if (threadIdx.x < ECNE_RESULT_BYTES / 4) dst[threadIdx.x] = src[threadIdx.x];
}

// rows whose descriptor carries any bit of `mask` (build_split's screen: rows that can write the constant wire's state)
__global__ __launch_bounds__(256) void k_count_shape(Job J, uint32_t mask, uint32_t* out) {
    uint32_t n = 0;
    for (uint32_t r = blockIdx.x * 256u + threadIdx.x; r < J.nC; r += gridDim.x * 256u) n += (J.rinfo[r].shape & mask) ? 1u : 0u;
    for (int d = 32; d >= 1; d >>= 1) n += __shfl_xor(n, d, 64);
    if ((threadIdx.x & 63) == 0 && n) atomicAdd(out, n);
}

// ---- build_split's screen on the device: the groups of rows that share a variable other than the constant wire, from the resident
// fan-out lists (variable -> rows, :628-633: the only way a pop reaches another row) -- union-find over the ROWS by atomic hooking
// (the larger root under the smaller, path halving), one thread per variable for lists of up to 16 rows, one workgroup per longer list
// (a multiplexer's select wire is in 84 000 rows: consecutive pairs, lanes across them); P5's row pairs (:1492-1536) as build_split
// unites them. Then every row's root, the rows per root, the number of roots and the largest group. The host plan that follows a
// positive answer unites through zero coefficients too (a dictionary key the device arrays do not hold): its groups can only be coarser.
__device__ __forceinline__ uint32_t cc_find(uint32_t* p, uint32_t x) {
    for (;;) {
        const uint32_t px = __hip_atomic_load(&p[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (px == x) return x;
        const uint32_t ppx = __hip_atomic_load(&p[px], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (ppx != px) __hip_atomic_store(&p[x], ppx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (halving: any ancestor is a valid parent)
        x = px;
    }
}
__device__ __forceinline__ void cc_unite(uint32_t* p, uint32_t a, uint32_t b) {
    for (;;) {
        a = cc_find(p, a); b = cc_find(p, b);
        if (a == b) return;
        if (a > b) { const uint32_t t = a; a = b; b = t; }
        uint32_t expect = b;
        if (__hip_atomic_compare_exchange_strong(&p[b], &expect, a, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
    }
}
__global__ __launch_bounds__(256) void k_cc_init(uint32_t* parent, uint32_t* cnt, uint32_t n) {
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) { parent[i] = i; cnt[i] = 0; }
}
#define ECNE_CC_SHORT 16u
#define ECNE_CC_LONGCAP 65536u
// out: [0] roots, [1] largest group, [2] long lists, [3] long lists that did not fit the list (the answer is then void)
__global__ __launch_bounds__(256) void k_cc_hook(Job J, uint32_t* parent, uint32_t* long_vars, uint32_t* out) {
    const uint32_t nV = J.nV;
    for (uint32_t v = 2 + blockIdx.x * 256u + threadIdx.x; v <= nV; v += gridDim.x * 256u) {
        const uint32_t f0 = J.fo_ptr[v], f1 = J.fo_ptr[v + 1];
        if (f1 - f0 < 2) continue;
        if (f1 - f0 > ECNE_CC_SHORT) {
            const uint32_t k = atomicAdd(&out[2], 1u);
            if (k < ECNE_CC_LONGCAP) long_vars[k] = v; else atomicAdd(&out[3], 1u);
            continue;
        }
        const uint32_t r0 = J.fo_rows[f0];
        for (uint32_t e = f0 + 1; e < f1; ++e) cc_unite(parent, r0, J.fo_rows[e]);
    }
    // P5's pairs: rows i, i + 1 with two non-zero terms in C, then none in C and one in B (the static half of :1493-1506)
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i + 1 < J.nC; i += gridDim.x * 256u)
        if (J.rpC[i + 1] - J.rpC[i] == 2 && J.rpC[i + 2] - J.rpC[i + 1] == 0 && J.rpB[i + 2] - J.rpB[i + 1] == 1) cc_unite(parent, i, i + 1);
}
__global__ __launch_bounds__(256) void k_cc_hook_long(Job J, uint32_t* parent, const uint32_t* long_vars, const uint32_t* out) {
    const uint32_t n = out[2] < ECNE_CC_LONGCAP ? out[2] : ECNE_CC_LONGCAP;
    for (uint32_t k = blockIdx.x; k < n; k += gridDim.x) {
        const uint32_t v = long_vars[k];
        const uint32_t f0 = J.fo_ptr[v], f1 = J.fo_ptr[v + 1];
        for (uint32_t e = f0 + 1 + threadIdx.x; e < f1; e += 256u) cc_unite(parent, J.fo_rows[e - 1], J.fo_rows[e]);
    }
}
__global__ __launch_bounds__(256) void k_cc_count(uint32_t* parent, uint32_t* cnt, uint32_t n, uint32_t* out) {
    // (one atomic per distinct root of a wavefront: a file that is ONE group would otherwise send a million additions to one word)
    uint32_t roots = 0, best = 0;
    for (uint32_t i0 = blockIdx.x * 256u; i0 < n; i0 += gridDim.x * 256u) {      // (uniform trip count: ballots inside)
        const uint32_t i = i0 + threadIdx.x;
        bool todo = i < n;
        const uint32_t r = todo ? cc_find(parent, i) : 0u;
        if (todo && r == i) ++roots;
        for (uint64_t m = __ballot(todo); m; m = __ballot(todo)) {
            const uint32_t lr = (uint32_t)__shfl((int)r, __ffsll((long long)m) - 1, 64);
            const uint64_t same = __ballot(todo && r == lr);
            if ((threadIdx.x & 63u) == (uint32_t)(__ffsll((long long)m) - 1)) {
                const uint32_t c = atomicAdd(&cnt[lr], (uint32_t)__popcll(same)) + (uint32_t)__popcll(same);
                best = c > best ? c : best;
            }
            if (r == lr) todo = false;
        }
    }
    for (int d = 32; d >= 1; d >>= 1) { roots += __shfl_xor(roots, d, 64); const uint32_t o = __shfl_xor(best, d, 64); best = o > best ? o : best; }
    if ((threadIdx.x & 63u) == 0) { if (roots) atomicAdd(&out[0], roots); if (best) atomicMax(&out[1], best); }
}

// ---- a split file (SplitPlan): one part's state into the file's own arrays; the part's copy of the constant wire against what setup
// left (any difference: the file is solved again as one system), part 0's copy is the one that is kept
__global__ __launch_bounds__(256) void k_scatter_part(Job K, Job P, const uint32_t* map, uint32_t nv) {
    for (uint32_t v = 1 + blockIdx.x * 256u + threadIdx.x; v <= nv; v += gridDim.x * 256u) {
        if (v == 1) {
            const Family* const F = K.family;
            bool same = K.flags[1] == (uint8_t)F->snap_flags && K.nvalues[1] == (uint8_t)F->snap_nvalues && K.abz[1] == F->snap_abz;
            for (int k = 0; k < 4; ++k) same = same && K.lb[4 + k] == F->snap_lb[k] && K.ub[4 + k] == F->snap_ub[k];
            for (int k = 0; k < 8; ++k) same = same && K.values[8 + k] == F->snap_values[k];
            if (!same) atomicOr(&K.family->var1_bad, 1u);
            if (K.fam_rank != 0) continue;
        }
        const uint32_t pv = map[v];
        P.flags[pv] = K.flags[v];
        P.nvalues[pv] = K.nvalues[v];
        const int32_t a = K.abz[v];
        P.abz[pv] = a > 0 ? (int32_t)map[a] : a;          // (a group tag is a variable id, :960)
        for (int k = 0; k < 4; ++k) { P.lb[4ull * pv + k] = K.lb[4ull * v + k]; P.ub[4ull * pv + k] = K.ub[4ull * v + k]; }
        for (int k = 0; k < 8; ++k) P.values[8ull * pv + k] = K.values[8ull * v + k];
    }
}
// ... and the verdict counts (:1558-1597) of the file from its own lists, as k_solve's epilogue takes them
__global__ __launch_bounds__(256) void k_part_counts(Job P) {
    __shared__ unsigned int acc[3];
    if (threadIdx.x < 3) acc[threadIdx.x] = 0;
    __syncthreads();
    uint32_t un = 0, nn = 0, ut = 0;
    const uint32_t g = blockIdx.x * 256u + threadIdx.x, st = gridDim.x * 256u;
    for (uint32_t v = 1 + g; v <= P.nV; v += st)
        if (P.nontrivial[v]) { nn++; if (P.flags[v] & 1) un++; }
    for (uint32_t i = g; i < P.nTarget; i += st)
        if (P.flags[P.targets[i]] & 1) ut++;
    if (un) atomicAdd(&acc[0], un);
    if (nn) atomicAdd(&acc[1], nn);
    if (ut) atomicAdd(&acc[2], ut);
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(&P.ctr->unique_nontrivial, (unsigned long long)acc[0]);
        atomicAdd(&P.ctr->n_nontrivial, (unsigned long long)acc[1]);
        atomicAdd(&P.ctr->unique_targets, (unsigned long long)acc[2]);
    }
}

// ecne_result_digest: per variable a splitmix64 chain over its state, summed over the variables (commutative: any thread order)
__device__ __forceinline__ unsigned long long dg_mix(unsigned long long x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__global__ __launch_bounds__(256) void k_state_digest(Job J, uint32_t n_vars, unsigned long long* out) {
    unsigned long long s0 = 0, s1 = 0;
    for (uint32_t v = 1 + blockIdx.x * 256u + threadIdx.x; v <= n_vars; v += gridDim.x * 256u) {
        unsigned long long h = dg_mix(v);
        const uint32_t nvl = J.nvalues[v];
        h = dg_mix(h ^ (unsigned long long)(J.flags[v] & 3u));
        h = dg_mix(h ^ (unsigned long long)(uint32_t)J.abz[v]);
        h = dg_mix(h ^ (unsigned long long)nvl);
        for (int k = 0; k < 4; ++k) h = dg_mix(h ^ J.lb[4ull * v + k]);
        for (int k = 0; k < 4; ++k) h = dg_mix(h ^ J.ub[4ull * v + k]);
        for (uint32_t k = 0; k < 4u * (nvl < 2u ? nvl : 2u); ++k) h = dg_mix(h ^ J.values[8ull * v + k]);
        s0 += h;
        s1 += dg_mix(h ^ 0xA5A5A5A5A5A5A5A5ull);
    }
    for (int d = 32; d >= 1; d >>= 1) { s0 += __shfl_xor(s0, d, 64); s1 += __shfl_xor(s1, d, 64); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&out[0], s0); atomicAdd(&out[1], s1); }
}

// launch scratch of the calling thread (job descriptors, workgroup table, events): kept between solves
struct LaunchScratch {
    int device = -1;
    Job* d_jobs = nullptr; size_t jobs_cap = 0;
    WgDesc* d_descs = nullptr; size_t descs_cap = 0;
    unsigned char* d_res = nullptr; size_t res_cap = 0;      // the leading (result) part of every job's Counters, gathered by k_gather_results
    hipEvent_t e0 = nullptr, e1 = nullptr;
    // (a launch that holds a multi-workgroup job next to single-workgroup ones: the latter on a stream of their own, through the kernel without team code)
    hipStream_t side = nullptr; hipEvent_t e_up = nullptr, e_side = nullptr;
    int side_stream() {
        if (side) return K_OK;
        if (hipStreamCreateWithFlags(&side, hipStreamNonBlocking) != hipSuccess) { side = nullptr; (void)hipGetLastError(); return K_ENODEVICE; }
        if (hipEventCreateWithFlags(&e_up, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&e_side, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); return K_ENODEVICE; }
        return K_OK;
    }
    void release() {
        if (device < 0) return;
        RestoreDevice restore;
        (void)hipSetDevice(device);
        if (d_res) (void)hipFree(d_res);
        if (d_jobs) (void)hipFree(d_jobs);
        if (d_descs) (void)hipFree(d_descs);
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
        if (e_up) (void)hipEventDestroy(e_up);
        if (e_side) (void)hipEventDestroy(e_side);
        if (side) (void)hipStreamDestroy(side);
        *this = LaunchScratch();
    }
    int prepare(int dev, size_t n_jobs, size_t n_descs) {
        if (device != dev) release();
        device = dev;
        if (n_jobs > jobs_cap) {
            if (d_jobs) (void)hipFree(d_jobs);
            d_jobs = nullptr; jobs_cap = 0;
            if (hipMalloc((void**)&d_jobs, sizeof(Job) * n_jobs) != hipSuccess) return K_ENODEVICE;
            jobs_cap = n_jobs;
        }
        if (n_jobs > res_cap) {
            if (d_res) (void)hipFree(d_res);
            d_res = nullptr; res_cap = 0;
            if (hipMalloc((void**)&d_res, ECNE_RESULT_BYTES * n_jobs) != hipSuccess) return K_ENODEVICE;
            res_cap = n_jobs;
        }
        if (n_descs > descs_cap) {
            if (d_descs) (void)hipFree(d_descs);
            d_descs = nullptr; descs_cap = 0;
            if (hipMalloc((void**)&d_descs, sizeof(WgDesc) * n_descs) != hipSuccess) return K_ENODEVICE;
            descs_cap = n_descs;
        }
        if (!e0 && hipEventCreate(&e0) != hipSuccess) return K_ENODEVICE;
        if (!e1 && hipEventCreate(&e1) != hipSuccess) return K_ENODEVICE;
        return K_OK;
    }
    ~LaunchScratch() { /* (thread exit: the HIP runtime may already be gone; the few KB are left to process teardown) */ }
};
static LaunchScratch& launch_scratch() { static thread_local LaunchScratch s; return s; }

// ------------------------------------------------------------------------------------ layout
static void build_small_lists(ecne_system& S, uint32_t nVall, std::vector<uint32_t>* marks);
static void row_variables_in_set_order(const Rows& R, size_t i, jl::SlotTable& set, std::vector<int64_t>& out);

// IntDisjointSet as secp_solve's setup uses it (:634-678): find_root raises BoundsError outside 1..length, push! appends an element
struct HostDsu {
    std::vector<uint32_t> parent;      // 1-based
    bool err = false;
    explicit HostDsu(size_t n) : parent(n + 1) { for (size_t i = 0; i <= n; ++i) parent[i] = (uint32_t)i; }
    size_t size() const { return parent.size() - 1; }
    uint32_t root(int64_t x) {
        if (x < 1 || (size_t)x > size()) { err = true; return 0; }
        uint32_t r = (uint32_t)x;
        while (parent[r] != r) r = parent[r];
        for (uint32_t c = (uint32_t)x; parent[c] != r;) { const uint32_t nx = parent[c]; parent[c] = r; c = nx; }
        return r;
    }
    uint32_t find(int64_t x, bool& bad) const {
        if (x < 1 || (size_t)x > size()) { bad = true; return 0; }
        uint32_t r = (uint32_t)x;
        while (parent[r] != r) r = parent[r];
        return r;
    }
    void unite(int64_t a, int64_t b) {
        const uint32_t ra = root(a);
        if (err) return;
        const uint32_t rb = root(b);
        if (err) return;
        if (ra != rb) parent[rb] = ra;
    }
    uint32_t push() { parent.push_back((uint32_t)parent.size()); return (uint32_t)size(); }
};

// Tables for the reference's lazy BoundsError (oob.hip.hpp; format: the OOB_* words there). Host-laid systems only (the device
// layout hands a system with such ids back to the host): per row that names one with a non-zero coefficient, the variables in
// front of the first such id in the order each rule walks the row -- R1: B, A (nonzeroKeys order, :828-846), then C; R2 and P3:
// getVariables order (:36-56); P4: A in nonzeroKeys order (:1431), then B's key (:1443); P5: A of row i (:1503) -- and, for secp_solve, the dsu of the setup (:634-678) with ids as they are.
static void build_oob_tables(ecne_system& S, const std::vector<uint8_t>& a_equal_next) {
    Layout& L = S.L;
    const Rows& R = S.rows();
    const uint32_t nC = L.nC;
    const uint32_t nVref = (uint32_t)std::min<int64_t>(S.n_vars, 0x7FFFFFFF);
    auto is_oob = [&](uint32_t v) { return v > nVref; };
    std::vector<uint32_t> rowids, rowrec_off, recs, p5;
    uint32_t n_p5 = 0;
    jl::SlotTable set;
    std::vector<int64_t> gv;
    for (uint32_t i = 0; i < nC; ++i) {
        bool any = false;
        for (int p = 0; p < 3 && !any; ++p)
            for (uint32_t e = L.rp[p][i]; e < L.rp[p][i + 1]; ++e) if (is_oob(L.col[p][e])) { any = true; break; }
        if (!any) continue;
        const uint32_t nA = L.rp[0][i + 1] - L.rp[0][i], nB = L.rp[1][i + 1] - L.rp[1][i], nCc = L.rp[2][i + 1] - L.rp[2][i];
        uint32_t flags = 0;
        std::vector<uint32_t> strict, cpart, r2, p3, p4;
        if (nA == 0 && nB == 0) flags |= 1u;                 // OOBF_LINEAR
        if (nCc == 0) flags |= 4u;                           // OOBF_C_EMPTY
        if (nCc == 0) {
            // P4's visit (:1427-1448): `unique_a` walks A in nonzeroKeys order while unique (:1431-1436), then -- at most one key in B --
            // B's key is read (:1442-1448)
            bool a_hit = false;
            for (uint32_t e = L.rp[0][i]; e < L.rp[0][i + 1]; ++e) {
                if (is_oob(L.col[0][e])) { a_hit = true; break; }
                p4.push_back(L.col[0][e]);
            }
            if (a_hit) flags |= 8u;                          // OOBF_P4_A_HAS
            if (nB == 1 && is_oob(L.col[1][L.rp[1][i]])) flags |= 16u;      // OOBF_P4_B
        }
        bool hit = false;
        for (int p = 1; p >= 0 && !hit; --p)                 // B first, then A
            for (uint32_t e = L.rp[p][i]; e < L.rp[p][i + 1]; ++e) {
                if (is_oob(L.col[p][e])) { hit = true; break; }
                strict.push_back(L.col[p][e]);
            }
        if (hit) flags |= 2u;                                // OOBF_STRICT_HAS
        else
            for (uint32_t e = L.rp[2][i]; e < L.rp[2][i + 1]; ++e) {
                if (is_oob(L.col[2][e])) break;
                cpart.push_back(L.col[2][e]);
            }
        gv.clear();
        row_variables_in_set_order(R, i, set, gv);
        for (int64_t v : gv) {
            if (is_oob((uint32_t)v)) break;
            if (nCc == 0) r2.push_back((uint32_t)v);
            bool inA = false, inB = false;
            for (uint32_t e = L.rp[0][i]; e < L.rp[0][i + 1]; ++e) inA |= L.col[0][e] == (uint32_t)v;
            for (uint32_t e = L.rp[1][i]; e < L.rp[1][i + 1]; ++e) inB |= L.col[1][e] == (uint32_t)v;
            if (inA && inB) p3.push_back((uint32_t)v);
        }
        rowids.push_back(i);
        rowrec_off.push_back((uint32_t)recs.size());
        recs.push_back(flags);
        recs.push_back((uint32_t)strict.size()); recs.push_back((uint32_t)cpart.size()); recs.push_back((uint32_t)r2.size()); recs.push_back((uint32_t)p3.size());
        recs.push_back((uint32_t)p4.size());
        recs.insert(recs.end(), strict.begin(), strict.end());
        recs.insert(recs.end(), cpart.begin(), cpart.end());
        recs.insert(recs.end(), r2.begin(), r2.end());
        recs.insert(recs.end(), p3.begin(), p3.end());
        recs.insert(recs.end(), p4.begin(), p4.end());
    }
    // P5 (:1492-1550) at row i: the three counts, then A of row i while unique, then (maps equal, y, C's keys) variable_states[y]
    for (uint32_t i = 0; i + 1 < nC; ++i) {
        const uint32_t nc_next = L.rp[2][i + 2] - L.rp[2][i + 1], nb_next = L.rp[1][i + 2] - L.rp[1][i + 1], nc_this = L.rp[2][i + 1] - L.rp[2][i];
        if (nc_next != 0 || nb_next != 1 || nc_this != 2) continue;
        std::vector<uint32_t> pre;
        bool hit = false;
        for (uint32_t e = L.rp[0][i]; e < L.rp[0][i + 1]; ++e) {
            if (is_oob(L.col[0][e])) { hit = true; break; }
            pre.push_back(L.col[0][e]);
        }
        if (!hit) {
            if (!a_equal_next[i]) continue;
            const uint32_t y = L.col[1][L.rp[1][i + 1]];
            if (y == 1 || !is_oob(y)) continue;
            bool bad = false;
            for (uint32_t k = L.rp[2][i]; k < L.rp[2][i + 1]; ++k) if (L.col[2][k] != 1 && L.col[2][k] != y) bad = true;
            if (bad) continue;
        }
        p5.push_back(i); p5.push_back(hit ? 0u : 1u); p5.push_back((uint32_t)pre.size());
        p5.insert(p5.end(), pre.begin(), pre.end());
        ++n_p5;
    }
    // secp_solve: the dsu of the setup, with the ids as they are
    std::vector<uint32_t> k1, k2;
    for (uint32_t i = 0; i < S.specials.size(); ++i) { if (S.specials[i].name == "BigMultModP") k1.push_back(i); else if (S.specials[i].name == "BigLessThan") k2.push_back(i); }
    HostDsu dsu((size_t)nVref);
    {
        const fp::u256 ONE = fp::make(1), PM1 = fp::pminus1();
        std::map<std::array<uint64_t, 4>, uint32_t> const_vals;
        for (uint32_t e = 0; e < nC && !dsu.err; ++e) {
            if (L.rp[0][e + 1] != L.rp[0][e] || L.rp[1][e + 1] != L.rp[1][e]) continue;
            const uint64_t d0 = R.ptr[2][e];
            if (R.ptr[2][e + 1] - d0 != 2) continue;
            const uint32_t c0 = L.rp[2][e], nl = L.rp[2][e + 1] - c0;      // l = nonzeroKeys(c) in Set order
            const fp::u256 &u = R.coef[2][d0], &v = R.coef[2][d0 + 1];
            if ((fp::eq(u, ONE) && fp::eq(v, PM1)) || (fp::eq(u, PM1) && fp::eq(v, ONE))) { dsu.unite(L.col[2][c0], L.col[2][c0 + 1]); continue; }
            if (nl == 0) { dsu.err = true; break; }
            bool constant_val = false;
            for (uint32_t t = 0; t < nl; ++t) constant_val |= L.col[2][c0 + t] == 1;
            uint32_t non_one = L.col[2][c0], pos = 0;
            if (non_one == 1) { if (nl < 2) { dsu.err = true; break; } non_one = L.col[2][c0 + 1]; pos = 1; }
            if (!constant_val) continue;
            fp::u256 c1 = fp::make(0), cx = fp::make(L.coef[2][4ull * (c0 + pos)], L.coef[2][4ull * (c0 + pos) + 1], L.coef[2][4ull * (c0 + pos) + 2], L.coef[2][4ull * (c0 + pos) + 3]);
            for (uint32_t t = 0; t < nl; ++t)
                if (L.col[2][c0 + t] == 1) c1 = fp::make(L.coef[2][4ull * (c0 + t)], L.coef[2][4ull * (c0 + t) + 1], L.coef[2][4ull * (c0 + t) + 2], L.coef[2][4ull * (c0 + t) + 3]);
            const fp::u256 value = fp::mul(c1, fp::inv(fp::neg(cx)));      // divexact(c[1], -c[non_one]) (:662)
            const std::array<uint64_t, 4> key = {value.w[0], value.w[1], value.w[2], value.w[3]};
            auto it = const_vals.find(key);
            if (it == const_vals.end()) it = const_vals.emplace(key, dsu.push()).first;
            dsu.unite(non_one, it->second);
        }
    }
    L.dsu_err = dsu.err;
    std::vector<uint32_t> codes(std::max<size_t>(k1.size() * k2.size(), 1), 0);
    if (!dsu.err)
        for (size_t a = 0; a < k1.size(); ++a)
            for (size_t b = 0; b < k2.size(); ++b) {
                const Special &ci = S.specials[k1[a]], &cj = S.specials[k2[b]];
                if (ci.inputs.size() < 9 || cj.inputs.size() < 6) continue;      // (the device raises on the sizes first)
                uint32_t code = 0;
                bool same = true;
                for (int k = 1; k <= 6 && !code; ++k) {
                    bool bad = false;
                    const uint32_t ra = dsu.find(ci.inputs[(size_t)k + 2], bad), rb = bad ? 0 : dsu.find(cj.inputs[(size_t)k - 1], bad);
                    if (bad) code |= 1u;
                    else if (ra != rb) same = false;
                }
                if (!code && same) {
                    if (cj.outputs.empty()) code |= 1u;                                   // constraint_j[3][1]
                    else if (cj.outputs[0] > S.n_vars) code |= 2u;
                    else {
                        bool o = false;
                        for (int64_t v : ci.outputs) o |= v > S.n_vars;
                        for (int idx : {1, 2, 3, 7, 8, 9}) o |= ci.inputs[(size_t)idx - 1] > S.n_vars;
                        if (o) code |= 4u;
                    }
                }
                codes[a * k2.size() + b] = code;
            }
    std::vector<uint32_t>& B = L.oob_blob;
    B.assign(8, 0);
    B[0] = (uint32_t)rowids.size(); B[1] = n_p5; B[2] = nVref; B[3] = (uint32_t)dsu.size();
    B[4] = (uint32_t)B.size(); B.insert(B.end(), rowids.begin(), rowids.end());
    B[5] = (uint32_t)B.size();
    const uint32_t rec_base = (uint32_t)(B.size() + rowrec_off.size());
    for (uint32_t o : rowrec_off) B.push_back(rec_base + o);
    B.insert(B.end(), recs.begin(), recs.end());
    B[6] = (uint32_t)B.size(); B.insert(B.end(), p5.begin(), p5.end());
    B[7] = (uint32_t)B.size(); B.insert(B.end(), codes.begin(), codes.end());
}

static void build_layout(ecne_system& S) {
    Layout& L = S.L;
    const Rows& R = S.rows();
    const size_t nC = R.n();
    L = Layout();
    L.nC = (uint32_t)nC;
    L.nV = (uint32_t)S.n_vars;
    const size_t BLK = 4096, nblk = (nC + BLK - 1) / BLK;
    const unsigned W = for_chunks_workers(std::max<size_t>(nblk, 1));
    // a special or a row may mention a variable above nWires+1 only in malformed input; size state
    // arrays for the largest id seen so that indexing stays in bounds. Same pass: non-zero terms per
    // block and part, i.e. where every block of rows starts in the flat arrays.
    std::vector<uint32_t> wmax(W, L.nV), wmax_nz(W, 0);
    std::vector<uint8_t> wdsu(W, 0);
    std::vector<uint64_t> blk_pos[3];
    for (int p = 0; p < 3; ++p) blk_pos[p].assign(nblk + 1, 0);
    for_chunks(nblk, [&](size_t blk, unsigned w) {
        const size_t i0 = blk * BLK, i1 = std::min(nC, i0 + BLK);
        uint32_t mx = wmax[w], mxnz = wmax_nz[w];
        for (int p = 0; p < 3; ++p) {
            uint64_t nz = 0;
            for (uint64_t k = R.ptr[p][i0]; k < R.ptr[p][i1]; ++k) {
                mx = std::max(mx, R.var[p][k]);
                if (!fp::is_zero(R.coef[p][k])) { ++nz; mxnz = std::max(mxnz, R.var[p][k]); }
            }
            blk_pos[p][blk + 1] = nz;
        }
        wmax[w] = mx;
        wmax_nz[w] = mxnz;
    });
    uint32_t maxv = L.nV;
    for (uint32_t m : wmax) maxv = std::max(maxv, m);
    for (auto& sp : S.specials) {
        for (int64_t v : sp.inputs) maxv = std::max<uint32_t>(maxv, (uint32_t)v);
        for (int64_t v : sp.outputs) maxv = std::max<uint32_t>(maxv, (uint32_t)v);
    }
    const uint32_t nVall = maxv;   // arrays hold ids 0..nVall
    L.nV = nVall;
    // ids above num_variables that a rule can ever read (a non-zero coefficient, a special's lists): the reference raises
    // BoundsError at the first such read (variable_states has num_variables entries, :681) -- build_oob_tables below
    bool oob = false;
    for (uint32_t m : wmax_nz) oob |= (int64_t)m > S.n_vars;
    for (auto& sp : S.specials) {
        for (int64_t v : sp.inputs) oob |= v > S.n_vars;
        for (int64_t v : sp.outputs) oob |= v > S.n_vars;
    }
    for (int p = 0; p < 3; ++p) {
        for (size_t blk = 0; blk < nblk; ++blk) blk_pos[p][blk + 1] += blk_pos[p][blk];
        L.nnz[p] = blk_pos[p][nblk];
        L.rp[p].resize(nC + 1);
        L.rp[p][0] = 0;
        L.col[p].resize(L.nnz[p]);
        L.coef[p].resize(4 * L.nnz[p]);
    }
    L.rinfo.resize(nC);
    L.nontrivial.assign((size_t)nVall + 1, 0);
    std::vector<uint8_t> a_equal_next(nC, 0);
    std::vector<std::vector<uint32_t>> blk_p4(nblk), blk_cls(nblk);
    std::vector<uint32_t> blk_vals(nblk + 1, 0);   // rows of the block that own a pair of value slots

    const fp::u256 ONE = fp::make(1), PM1 = fp::pminus1();
    struct E { uint32_t v; fp::u256 c; };
    // (a part of a split file: hash orders from the file's own ids)
    const uint32_t* const ov = S.orig_var.empty() ? nullptr : S.orig_var.data();
    auto OV = [ov](uint32_t v) -> int64_t { return (int64_t)(ov ? ov[v] : v); };
    // per-row pass, one block of rows per task (every row writes its own slice of the flat arrays)
    for_chunks(nblk, [&](size_t blk, unsigned) {
    jl::SlotTable set;
    std::vector<E> nz[3], tmp;
    uint64_t pos[3] = {blk_pos[0][blk], blk_pos[1][blk], blk_pos[2][blk]};
    uint32_t vals_here = 0;
    const size_t i0 = blk * BLK, i1 = std::min(nC, i0 + BLK);
    for (size_t i = i0; i < i1; ++i) {
        RowInfo ri;
        std::memset(&ri, 0, sizeof ri);
        ri.validx = 0xFFFFFFFFu;
        uint32_t zc[3] = {0, 0, 0};
        bool c_has_key1 = false;
        for (int p = 0; p < 3; ++p) {
            // nonzeroKeys(part): a Set filled in dictionary order, iterated in slot order (:26-34)
            nz[p].clear();
            tmp.clear();
            for (uint64_t k = R.ptr[p][i]; k < R.ptr[p][i + 1]; ++k) {
                if (p == 2 && R.var[p][k] == 1) c_has_key1 = true;
                if (fp::is_zero(R.coef[p][k])) { zc[p]++; continue; }
                tmp.push_back({R.var[p][k], R.coef[p][k]});
            }
            if (tmp.size() <= 1) {   // nothing to order
                nz[p] = tmp;
            } else {
                set.reset();
                for (size_t t = 0; t < tmp.size(); ++t) { bool ins; set.upsert(OV(tmp[t].v), (int64_t)t, ins); }
                set.for_each([&](int64_t, int64_t pay) { nz[p].push_back(tmp[(size_t)pay]); });
            }
            for (auto& e : nz[p]) {
                L.col[p][pos[p]] = e.v;
                for (int w = 0; w < 4; ++w) L.coef[p][4 * pos[p] + w] = e.c.w[w];
                ++pos[p];
                __atomic_store_n(&L.nontrivial[e.v], (uint8_t)1, __ATOMIC_RELAXED);   // same value from every worker
            }
            L.rp[p][i + 1] = (uint32_t)pos[p];
        }
        const size_t nA = nz[0].size(), nB = nz[1].size(), nCc = nz[2].size();
        ri.lenC = (uint32_t)nCc;
        // secp_solve's dsu setup (:634-678): a two-entry C without A and B whose non-zero keys are none (`l[1]`, :650) or the
        // constant wire alone (`l[2]`, :652) is a BoundsError
        if (nA == 0 && nB == 0 && R.ptr[2][i + 1] - R.ptr[2][i] == 2 && (nCc == 0 || (nCc == 1 && nz[2][0].v == 1)))
            __atomic_store_n(&wdsu[0], (uint8_t)1, __ATOMIC_RELAXED);
        for (auto& e : nz[2]) if (e.v == 1) ri.shape |= SH_C_HAS1;
        if (nA + nB + nCc > ECNE_SMALL_ROW) ri.shape |= SH_BIG;
        if (nA || nB) ri.shape |= SH_HAS_AB;
        if (nCc == 0) {
            ri.shape |= SH_C_EMPTY;
            // S = (nzA u nzB) \ {1}
            uint32_t x = 0, distinct = 0;
            for (int p = 0; p < 2; ++p)
                for (auto& e : nz[p])
                    if (e.v != 1) {
                        if (distinct == 0) { x = e.v; distinct = 1; }
                        else if (e.v != x) distinct = 2;
                    }
            if (distinct == 0) ri.shape |= SH_R2_BOUNDSERR;
            else if (distinct == 1) {
                ri.shape |= SH_R2;
                ri.x = x;
                bool inA = false, inB = false;
                for (auto& e : nz[0]) inA |= e.v == x;
                for (auto& e : nz[1]) inB |= e.v == x;
                if (!inA || !inB) ri.shape |= SH_R2_DIV0;
            }
            // P4 static test (:1427-1466)
            if (nB == 1 && nA <= 2) {
                ri.shape |= SH_P4;
                ri.kpos = nz[1][0].v;
                uint32_t slope_index = 0;
                bool have = false;
                for (auto& e : nz[0])
                    if (e.v != 1) { slope_index = e.v; have = true; }   // last one wins (:1462-1465)
                ri.kneg = slope_index;
                if (!have) ri.shape |= SH_P4_DIV0;
                blk_p4[blk].push_back((uint32_t)i);
            }
        }
        if (!(ri.shape & SH_HAS_AB) && nCc > 0) {
            // R3 (:949-960)
            uint32_t nonone = 0, x = 0;
            for (auto& e : nz[2]) if (e.v != 1) { nonone++; x = e.v; }
            if (nonone == 1) { ri.shape |= SH_R3; ri.x = x; }
            uint32_t czero_eff = zc[2] + (((ri.shape & SH_R3) && !c_has_key1) ? 1u : 0u);
            if (czero_eff) ri.shape |= SH_CZERO;
            // R5 / R6 in dictionary order (:1082-1092, :1154-1173)
            const uint64_t d0 = R.ptr[2][i], d1 = R.ptr[2][i + 1];
            if (!czero_eff && d1 - d0 == 2) {
                const fp::u256 &u = R.coef[2][d0], &v = R.coef[2][d0 + 1];
                if ((fp::eq(u, ONE) && fp::eq(v, PM1)) || (fp::eq(u, PM1) && fp::eq(v, ONE))) {
                    ri.shape |= SH_R5;
                    ri.k1 = R.var[2][d0];
                    ri.k2 = R.var[2][d0 + 1];
                }
            }
            if (!czero_eff && d1 - d0 == 3) {
                int ones = 0, mones = 0;
                bool one_on_const = true;
                uint32_t k1 = 0, k2 = 0;
                for (uint64_t k = d0; k < d1; ++k) {
                    if (fp::eq(R.coef[2][k], ONE)) { ones++; if (R.var[2][k] != 1) one_on_const = false; }
                    else if (fp::eq(R.coef[2][k], PM1)) { if (mones == 0) k1 = R.var[2][k]; else k2 = R.var[2][k]; mones++; }
                }
                if (ones == 1 && mones == 2 && one_on_const) { ri.shape |= SH_R6; ri.k1 = k1; ri.k2 = k2; }
            }
            if (ri.shape & (SH_R5 | SH_R6)) {   // order of `for j in Set([k1, k2])`
                set.reset();
                bool ins;
                set.upsert(OV(ri.k1), 0, ins);
                set.upsert(OV(ri.k2), 1, ins);
                int64_t first = -1;
                set.for_each([&](int64_t key, int64_t) { if (first < 0) first = key; });
                if (first == OV(ri.k2) && ri.k1 != ri.k2) ri.shape |= SH_R56_SWAP;
            }
        }
        if ((ri.shape & SH_R2) || (!(ri.shape & SH_HAS_AB) && nCc > 0)) {
            ri.validx = 2 * vals_here;   // relative to the block; rebased below
            ++vals_here;
        }
        L.rinfo[i] = ri;
        if (nCc > ECNE_CLS_LANE) blk_cls[blk].push_back((uint32_t)i);
        // A-map equality with the next row, zeros included (:1512)
        if (i + 1 < nC) {
            const uint64_t x0 = R.ptr[0][i], x1 = R.ptr[0][i + 1], y0 = R.ptr[0][i + 1], y1 = R.ptr[0][i + 2];
            bool eq = (x1 - x0) == (y1 - y0);
            for (uint64_t k = x0; eq && k < x1; ++k) {
                bool found = false;
                for (uint64_t m = y0; m < y1; ++m)
                    if (R.var[0][m] == R.var[0][k]) { found = fp::eq(R.coef[0][m], R.coef[0][k]); break; }
                eq = found;
            }
            a_equal_next[i] = eq;
        }
    }
    blk_vals[blk + 1] = vals_here;
    });
    // value slots are numbered in row order; the per-block lists are joined in block order
    for (size_t blk = 0; blk < nblk; ++blk) blk_vals[blk + 1] += blk_vals[blk];
    L.n_vals = 2 * blk_vals[nblk];
    for_chunks(nblk, [&](size_t blk, unsigned) {
        const uint32_t base = 2 * blk_vals[blk];
        const size_t i0 = blk * BLK, i1 = std::min(nC, i0 + BLK);
        if (base)
            for (size_t i = i0; i < i1; ++i)
                if (L.rinfo[i].validx != 0xFFFFFFFFu) L.rinfo[i].validx += base;
    });
    for (size_t blk = 0; blk < nblk; ++blk) {
        L.p4_list.insert(L.p4_list.end(), blk_p4[blk].begin(), blk_p4[blk].end());
        L.cls_list.insert(L.cls_list.end(), blk_cls[blk].begin(), blk_cls[blk].end());
    }
    // P5 static candidates (:1492-1536)
    {
        std::vector<std::vector<uint32_t>> blk_rows(nblk), blk_y(nblk);
        for_chunks(nblk, [&](size_t blk, unsigned) {
            const size_t i0 = blk * BLK, i1 = std::min(nC, i0 + BLK);
            for (size_t i = i0; i < i1 && i + 1 < nC; ++i) {
                const uint32_t nc_next = L.rp[2][i + 2] - L.rp[2][i + 1];
                const uint32_t nb_next = L.rp[1][i + 2] - L.rp[1][i + 1];
                const uint32_t nc_this = L.rp[2][i + 1] - L.rp[2][i];
                if (nc_next != 0 || nb_next != 1 || nc_this != 2 || !a_equal_next[i]) continue;
                if (!S.orig_row.empty() && S.orig_row[i + 1] != S.orig_row[i] + 1) continue;      // (a part: neighbours in the file only)
                const uint32_t y = L.col[1][L.rp[1][i + 1]];
                if (y == 1) continue;
                bool bad = false;
                for (uint32_t k = L.rp[2][i]; k < L.rp[2][i + 1]; ++k)
                    if (L.col[2][k] != 1 && L.col[2][k] != y) bad = true;
                if (bad) continue;
                blk_rows[blk].push_back((uint32_t)i);
                blk_y[blk].push_back(y);
            }
        });
        for (size_t blk = 0; blk < nblk; ++blk) {
            L.p5_rows.insert(L.p5_rows.end(), blk_rows[blk].begin(), blk_rows[blk].end());
            L.p5_y.insert(L.p5_y.end(), blk_y[blk].begin(), blk_y[blk].end());
        }
    }
    // variable_to_indices (:628-633): ascending rows per variable. Every task owns a range of variable ids
    // and walks all rows for it (count, then fill), so each list comes out in row order whatever the
    // number of workers; the walk is a linear read of the column arrays.
    {
        const size_t nvar = (size_t)nVall + 1;
        const size_t nrange = std::min<size_t>(W, nvar);
        const size_t per = (nvar + nrange - 1) / nrange;
        std::vector<uint32_t> deg(nvar + 1, 0), last(nvar, 0xFFFFFFFFu);
        auto walk = [&](size_t rg, bool fill, std::vector<uint32_t>& cursor) {
            const uint32_t v0 = (uint32_t)(rg * per), v1 = (uint32_t)std::min(nvar, (rg + 1) * per);
            for (size_t i = 0; i < nC; ++i)
                for (int p = 0; p < 3; ++p)
                    for (uint32_t k = L.rp[p][i]; k < L.rp[p][i + 1]; ++k) {
                        const uint32_t v = L.col[p][k];
                        if (v < v0 || v >= v1) continue;
                        if (!fill) {
                            if (last[v] != (uint32_t)i) { last[v] = (uint32_t)i; deg[v + 1]++; }
                        } else if (last[v] != (uint32_t)i) {
                            last[v] = (uint32_t)i;
                            L.fo_rows[cursor[v]++] = (uint32_t)i;
                        }
                    }
        };
        std::vector<uint32_t> none;
        for_chunks(nrange, [&](size_t rg, unsigned) { walk(rg, false, none); });
        L.fo_ptr.assign(nvar + 1, 0);
        for (size_t v = 0; v < nvar; ++v) L.fo_ptr[v + 1] = L.fo_ptr[v] + deg[v + 1];
        L.fo_rows.resize(L.fo_ptr[nvar]);
        std::vector<uint32_t> cursor(L.fo_ptr.begin(), L.fo_ptr.end() - 1);
        std::fill(last.begin(), last.end(), 0xFFFFFFFFu);
        for_chunks(nrange, [&](size_t rg, unsigned) { walk(rg, true, cursor); });
    }
    // specials, I/O lists, nontrivial set (:600-618)
    build_small_lists(S, nVall, nullptr);
    uint64_t nnz = L.nnz[0] + L.nnz[1] + L.nnz[2];
    L.stream_bytes = (uint64_t)nC * (12 + 32 + 32) + nnz * 36 + L.nnz[2] * 4 + (uint64_t)L.n_vals * 32;
    L.host_arrays = true;
    L.n_p4 = (uint32_t)L.p4_list.size(); L.n_p5 = (uint32_t)L.p5_rows.size(); L.n_cls = (uint32_t)L.cls_list.size();
    L.fo_total = (uint32_t)L.fo_rows.size();
    L.dsu_err = wdsu[0] != 0;
    if (oob) build_oob_tables(S, a_equal_next);
    S.laid_out = true;
}

// ------------------------------------------------------------------------------------ device image
#define HIP_TRY(x)                                   \
    do {                                             \
        hipError_t e_ = (x);                         \
        if (e_ != hipSuccess) return K_ENODEVICE;      \
    } while (0)

struct Carver {
    size_t off = 0;
    size_t take(size_t bytes) {
        size_t o = off;
        off += (bytes + 255) & ~(size_t)255;
        return o;
    }
};

static uint32_t pow2_at_least(uint64_t x) {
    uint64_t p = 16;
    while (p < x) p <<= 1;
    return (uint32_t)p;
}

// specials, I/O lists and their nontrivial marks (:600-618): small, host side for both front-ends
static void build_small_lists(ecne_system& S, uint32_t nVall, std::vector<uint32_t>* marks) {
    Layout& L = S.L;
    L.sp_in_ptr.assign(1, 0);
    L.sp_out_ptr.assign(1, 0);
    L.sp_in.clear(); L.sp_out.clear(); L.sp_kind.clear(); L.knowns.clear(); L.targets.clear();
    for (auto& sp : S.specials) {
        for (int64_t v : sp.inputs) { L.sp_in.push_back((uint32_t)v); if (marks) marks->push_back((uint32_t)v); else L.nontrivial[(size_t)v] = 1; }
        for (int64_t v : sp.outputs) { L.sp_out.push_back((uint32_t)v); if (marks) marks->push_back((uint32_t)v); else L.nontrivial[(size_t)v] = 1; }
        L.sp_in_ptr.push_back((uint32_t)L.sp_in.size());
        L.sp_out_ptr.push_back((uint32_t)L.sp_out.size());
        L.sp_kind.push_back(sp.name == "BigMultModP" ? 1 : sp.name == "BigLessThan" ? 2 : 0);
    }
    for (int64_t v : S.knowns) if (v >= 1 && (uint64_t)v <= nVall) L.knowns.push_back((uint32_t)v);
    for (int64_t v : S.targets)
        if (v >= 1 && (uint64_t)v <= nVall) { L.targets.push_back((uint32_t)v); if (marks) marks->push_back((uint32_t)v); else L.nontrivial[(size_t)v] = 1; }
}

static int upload_system(ecne_system& S, int device) {
    if (S.dev.arena && S.dev.device == device) return K_OK;
    if (S.dev.arena) { (void)hipSetDevice(S.dev.device); (void)hipFree(S.dev.arena); S.dev = DeviceImage(); }
    HIP_TRY(hipSetDevice(device));
    // Static arrays: built on the device when the rows are there (device front-end), else laid out on the host and uploaded
    bool dev_layout = false;
    std::vector<uint32_t> marks;
    if (S.drows && S.drows->device == device && !(S.laid_out && S.L.host_arrays)) {
        uint32_t min_nv = 0;
        for (auto& sp : S.specials) {
            for (int64_t v : sp.inputs) min_nv = std::max<uint32_t>(min_nv, (uint32_t)v);
            for (int64_t v : sp.outputs) min_nv = std::max<uint32_t>(min_nv, (uint32_t)v);
        }
        int rc = fe::layout_on_device(*S.drows, (uint32_t)S.n_vars, min_nv, S.dev.lay);
        if (rc == K_OK && S.dev.lay->cnt.nVall > (uint64_t)S.n_vars) { rc = fe::FE_FALLBACK; }      // ids above num_variables: the host lays the system out (build_oob_tables)
        if (rc == K_OK) {
            dev_layout = true;
            const fe::LayoutCounts& C = S.dev.lay->cnt;
            Layout& L = S.L;
            L = Layout();
            L.nC = C.nC; L.nV = C.nVall;
            for (int p = 0; p < 3; ++p) L.nnz[p] = C.nnz[p];
            L.n_vals = C.n_vals; L.n_p4 = C.nP4; L.n_p5 = C.nP5; L.n_cls = C.nCls; L.n_long = C.nLong; L.n_bigrows = C.nBigRows;
            L.fo_total = C.fo_total; L.maxrowC = C.maxrowC;
            L.dsu_err = C.dsu_err != 0;
            L.host_arrays = false;
            build_small_lists(S, C.nVall, &marks);
            const uint64_t nnz = L.nnz[0] + L.nnz[1] + L.nnz[2];
            L.stream_bytes = (uint64_t)L.nC * (12 + 32 + 32) + nnz * 36 + L.nnz[2] * 4 + (uint64_t)L.n_vals * 32;
            S.laid_out = true;
            g_fe_stats.layout_ms = S.dev.lay->ms;
            g_fe_stats.layout_dev = 1;
        } else if (!fe_falls_back(rc)) return rc;
        else S.dev.lay.reset();
    }
    if (!dev_layout) {
        { const int rc = sys_host_rows(S); if (rc != K_OK) return rc; }
        if (!S.laid_out || !S.L.host_arrays) build_layout(S);
        g_fe_stats.layout_dev = 0;
    }
    const Layout& L = S.L;
    const uint32_t nC = L.nC, nV = L.nV, nSp = (uint32_t)L.sp_kind.size();
    const uint32_t qcap = pow2_at_least((uint64_t)nC + 2);
    const uint32_t htcap = pow2_at_least((uint64_t)nC * 2 + 16);
    const uint32_t hotcap = 4096;
    const uint32_t nev = std::max<uint32_t>(L.n_p4, 64);
    const size_t fo_ptr_n = (size_t)nV + 2;
    Carver c;
    // host layout: the static arrays live in the arena; device layout: they live in S.dev.lay and the arena holds the rest
    auto take_h = [&](size_t bytes) { return dev_layout ? (size_t)0 : c.take(bytes); };
    size_t o_rp[3], o_col[3], o_coef[3];
    for (int p = 0; p < 3; ++p) {
        o_rp[p] = take_h(4ull * (nC + 1));
        o_col[p] = take_h(4ull * std::max<uint64_t>(L.nnz[p], 1));
        o_coef[p] = take_h(32ull * std::max<uint64_t>(L.nnz[p], 1));
    }
    size_t o_csort = c.take(4ull * std::max<uint64_t>(L.nnz[2], 1));
    size_t o_rinfo = take_h(sizeof(RowInfo) * std::max<size_t>(nC, 1));
    size_t o_vals = c.take(32ull * std::max<uint32_t>(L.n_vals, 1));
    size_t o_foptr = take_h(4ull * fo_ptr_n);
    size_t o_forows = take_h(4ull * std::max<size_t>(L.fo_total, 1));
    size_t o_spinptr = c.take(4ull * L.sp_in_ptr.size()), o_spin = c.take(4ull * std::max<size_t>(L.sp_in.size(), 1));
    size_t o_spoutptr = c.take(4ull * L.sp_out_ptr.size()), o_spout = c.take(4ull * std::max<size_t>(L.sp_out.size(), 1));
    size_t o_spkind = c.take(std::max<size_t>(nSp, 1));
    std::vector<uint32_t> k1_list, k2_list;
    for (uint32_t i = 0; i < nSp; ++i) { if (L.sp_kind[i] == 1) k1_list.push_back(i); else if (L.sp_kind[i] == 2) k2_list.push_back(i); }
    size_t o_k1 = c.take(4ull * std::max<size_t>(k1_list.size(), 1)), o_k2 = c.take(4ull * std::max<size_t>(k2_list.size(), 1));
    size_t o_knowns = c.take(4ull * std::max<size_t>(L.knowns.size(), 1)), o_targets = c.take(4ull * std::max<size_t>(L.targets.size(), 1));
    size_t o_nontriv = take_h((size_t)nV + 1);
    size_t o_p4 = take_h(4ull * std::max<size_t>(L.n_p4, 1));
    std::vector<uint32_t> p4_b, p4_s;
    std::vector<uint16_t> tbig;
    std::vector<uint32_t> bigrows, long_list;
    std::vector<uint32_t, RawAlloc<uint32_t>> rec, foi;
    uint32_t maxrowC = L.maxrowC;
    if (!dev_layout) {
        p4_b.resize(L.p4_list.size()); p4_s.resize(L.p4_list.size());
        for (size_t i = 0; i < L.p4_list.size(); ++i) {
            const RowInfo& ri = L.rinfo[L.p4_list[i]];
            p4_b[i] = ri.kpos;
            p4_s[i] = ri.kneg | ((ri.shape & SH_P4_DIV0) ? 0x80000000u : 0u);
        }
        tbig.assign(std::max<size_t>(nC, 1), 0);
        for (uint32_t r = 0; r < nC && bigrows.size() < ECNE_BIGTAB; ++r)
            if (L.rinfo[r].shape & SH_BIG) { bigrows.push_back(r); tbig[r] = (uint16_t)bigrows.size(); }
        for (uint32_t r = 0; r < nC; ++r) if (L.rinfo[r].shape & SH_BIG) long_list.push_back(r);
    }
    const size_t n_bigrows = dev_layout ? L.n_bigrows : bigrows.size(), n_long = dev_layout ? L.n_long : long_list.size();
    size_t o_p4b = take_h(4ull * std::max<size_t>(L.n_p4, 1)), o_p4s = take_h(4ull * std::max<size_t>(L.n_p4, 1));
    size_t o_cls = take_h(4ull * std::max<size_t>(L.n_cls, 1));
    size_t o_tbig = take_h(2ull * std::max<size_t>(nC, 1)), o_bigrows = take_h(4ull * std::max<size_t>(n_bigrows, 1));
    size_t o_long = take_h(4ull * std::max<size_t>(n_long, 1));
    size_t o_p5r = take_h(4ull * std::max<size_t>(L.n_p5, 1)), o_p5y = take_h(4ull * std::max<size_t>(L.n_p5, 1));
    // row records and inline fan-out lists (chain executor, fast wavefront rounds): the row in one line, the fan-out inline
    const bool chain = true;
    if (!dev_layout) {
        rec.resize(16ull * std::max<size_t>(nC, 1));
        const size_t RB = 8192, nrb = ((size_t)nC + RB - 1) / RB;
        if (nC == 0) std::fill(rec.begin(), rec.end(), 0u);
        for_chunks(nrb, [&](size_t blk, unsigned) {
            for (uint32_t r = (uint32_t)(blk * RB); r < std::min<size_t>(nC, (blk + 1) * RB); ++r) {
                const uint32_t la = L.rp[0][r + 1] - L.rp[0][r], lb = L.rp[1][r + 1] - L.rp[1][r], lc = L.rp[2][r + 1] - L.rp[2][r];
                uint32_t* w = &rec[16ull * r];
                std::memset(w, 0, 64);
                if (la + lb + lc > 15) { w[1] = 0xFFFFFFFFu; continue; }      // no record; words 1, 2: the row's watched pair (fastrow.hip.hpp), none yet
                w[0] = la | lb << 8 | lc << 16 | 1u << 24;
                uint32_t k = 1;
                for (int p = 0; p < 3; ++p)
                    for (uint32_t e = L.rp[p][r]; e < L.rp[p][r + 1]; ++e) w[k++] = L.col[p][e];
            }
        });
        const size_t nvar = L.fo_ptr.size() - 1;
        foi.resize(4ull * L.fo_ptr.size());
        std::memset(&foi[4 * nvar], 0, 16);
        const size_t nvb = (nvar + RB - 1) / RB;
        for_chunks(nvb, [&](size_t blk, unsigned) {
            for (size_t v = blk * RB; v < std::min(nvar, (blk + 1) * RB); ++v) {
                const uint32_t f0 = L.fo_ptr[v], n = L.fo_ptr[v + 1] - f0;
                uint32_t* w = &foi[4 * v];
                w[0] = n; w[1] = w[2] = w[3] = 0;
                if (n <= 3) for (uint32_t i = 0; i < n; ++i) w[1 + i] = L.fo_rows[f0 + i];
                else w[1] = f0;
            }
        });
        maxrowC = 0;
        for (uint32_t r = 0; r < nC; ++r) maxrowC = std::max(maxrowC, L.rp[2][r + 1] - L.rp[2][r]);
    }
    size_t o_rec = take_h(64ull * std::max<size_t>(nC, 1)), o_foi = take_h(16ull * fo_ptr_n);
    const size_t static_end = c.off;   // [o_rp[0], static_end): everything the solve only reads
    size_t o_flags = c.take((size_t)nV + 1), o_abz = c.take(4ull * (nV + 1));
    size_t o_lb = c.take(32ull * (nV + 1)), o_ub = c.take(32ull * (nV + 1));
    size_t o_nvalues = c.take((size_t)nV + 1), o_values = c.take(64ull * (nV + 1));
    size_t o_inq = c.take(2 * std::max<size_t>(nC, 1)), o_solved = c.take((size_t)nC + 1), o_flip3 = c.take(std::max<size_t>(nC, 1));
    size_t o_queue = c.take(4ull * qcap);
    size_t o_varmin = c.take(4ull * (nV + 1));
    size_t o_rdead = c.take(std::max<size_t>(nC, 1) + 4);   // read four rows at a time
    size_t o_p3stamp = c.take(4ull * std::max<size_t>(nC, 1));
    size_t o_clsdefer = c.take(4ull * ((size_t)nC + 1));
    size_t o_p3k = c.take(std::max<size_t>(nC, 1)), o_p3h = c.take(8ull * std::max<size_t>(nC, 1)), o_p3h2 = c.take(8ull * std::max<size_t>(nC, 1));
    size_t o_htkey = c.take(8ull * htcap), o_htkey2 = c.take(8ull * htcap), o_htnew = c.take(4ull * htcap), o_htfrozen = c.take(4ull * htcap);
    size_t o_htlist = c.take(4ull * ((size_t)nC + (size_t)ECNE_MAX_NWG * 2049 + 64));
    size_t o_hot = c.take(4ull * hotcap), o_fired = c.take((size_t)nC + nSp + 1), o_events = c.take(4ull * nev);
    size_t o_dmk[6];
    for (int p = 0; p < 6; ++p) o_dmk[p] = c.take(4ull * (nV + 1));       // mark planes of the drain rounds (drain.hip.hpp)
    size_t o_wmark = c.take(4ull * (nV + 1)), o_wmarkB = c.take(4ull * (nV + 1)), o_best = c.take(4ull * std::max<size_t>(nC, 1)), o_prank = c.take(4ull * std::max<size_t>(nC, 1));
    // one event slot list per rank of a round: single-workgroup rounds examine <= 4 * 512 queue entries,
    // multi-workgroup rounds <= min(rows, ECNE_MAX_NWG workgroups * 512 lanes * 2)
    const size_t max_ranks = std::max<size_t>((size_t)4 * ECNE_WG, std::min<size_t>((size_t)nC + 1, (size_t)ECNE_MAX_NWG * ECNE_WG * 2));
    size_t o_evcnt = c.take(4ull * max_ranks);
    size_t o_evbuf = c.take(4ull * max_ranks * ECNE_EVCAP), o_cand = c.take(4ull * std::max<size_t>(ECNE_CANDCAP, 8ull * nC));
    const size_t flatcap = std::max<size_t>((size_t)4 * ECNE_WG * ECNE_EVCAP, 16384) + (size_t)ECNE_BIGK * (maxrowC + 8);
    size_t o_fvar = c.take(4ull * flatcap), o_frank = c.take(4ull * flatcap), o_fbase = c.take(4ull * (flatcap + 1));
    size_t o_bigev = c.take(4ull * ((size_t)maxrowC * 3 + 64));
    const uint32_t bigstride = 2 * (maxrowC + 8);
    size_t o_bigpool = c.take(4ull * ECNE_MAX_NWG * ECNE_BIGK * bigstride);
    size_t o_oob = L.oob_blob.empty() ? 0 : c.take(4ull * L.oob_blob.size());
    size_t o_ctr = c.take(sizeof(Counters));
    char* base = nullptr;
    HIP_TRY(hipMalloc((void**)&base, c.off));
    // (the image is published -- S.dev.arena set -- only once it is complete: a failed step below frees the arena, so the next
    //  call starts over instead of launching on a half-filled image)
    struct ArenaGuard { char* p; ecne_system& S; bool done = false; ~ArenaGuard() { if (!done) { (void)hipFree(p); S.dev = DeviceImage(); } } } arena_guard{base, S};
    if (getenv("ECNE_POISON")) HIP_TRY(hipMemset(base, 0xA5, c.off));      // test hook: whatever the solve reads before writing it shows up
    // The image of a SMALL system (a circomlib file, a part of a split file) is some forty arrays of a few kilobytes: one synchronous
    // hipMemcpy each cost more than the bytes (~10 us a call: 240 parts = 100 ms of calls). Pieces are collected and neighbouring ones
    // (the carve puts the static arrays side by side, 256-byte aligned) go up as ONE copy from a staging buffer; pieces of a megabyte
    // and more are copied as they are.
    struct Piece { size_t off; const void* src; size_t bytes; };
    std::vector<Piece> pieces;
    auto up = [&](size_t off, const void* src, size_t bytes) -> hipError_t {
        if (!bytes) return hipSuccess;
        if (bytes >= (1u << 20)) return hipMemcpy(base + off, src, bytes, hipMemcpyHostToDevice);
        pieces.push_back({off, src, bytes});
        return hipSuccess;
    };
    auto up_flush = [&]() -> hipError_t {
        std::sort(pieces.begin(), pieces.end(), [](const Piece& a, const Piece& b) { return a.off < b.off; });
        std::vector<unsigned char> stage;
        for (size_t i = 0; i < pieces.size();) {
            size_t j = i + 1, end = pieces[i].off + pieces[i].bytes;
            while (j < pieces.size() && pieces[j].off >= end && pieces[j].off - end <= 512 && pieces[j].off + pieces[j].bytes - pieces[i].off <= (8u << 20)) { end = pieces[j].off + pieces[j].bytes; ++j; }
            if (j == i + 1) {
                const hipError_t e = hipMemcpy(base + pieces[i].off, pieces[i].src, pieces[i].bytes, hipMemcpyHostToDevice);
                if (e != hipSuccess) return e;
            } else {
                stage.assign(end - pieces[i].off, 0);      // (the alignment gaps between two arrays belong to nobody)
                for (size_t k = i; k < j; ++k) std::memcpy(stage.data() + (pieces[k].off - pieces[i].off), pieces[k].src, pieces[k].bytes);
                const hipError_t e = hipMemcpy(base + pieces[i].off, stage.data(), stage.size(), hipMemcpyHostToDevice);
                if (e != hipSuccess) return e;
            }
            i = j;
        }
        pieces.clear();
        return hipSuccess;
    };
    if (!dev_layout) {
        for (int p = 0; p < 3; ++p) {
            HIP_TRY(up(o_rp[p], L.rp[p].data(), 4ull * L.rp[p].size()));
            HIP_TRY(up(o_col[p], L.col[p].data(), 4ull * L.col[p].size()));
            HIP_TRY(up(o_coef[p], L.coef[p].data(), 8ull * L.coef[p].size()));
        }
        HIP_TRY(up(o_rinfo, L.rinfo.data(), sizeof(RowInfo) * L.rinfo.size()));
        HIP_TRY(up(o_foptr, L.fo_ptr.data(), 4ull * L.fo_ptr.size()));
        HIP_TRY(up(o_forows, L.fo_rows.data(), 4ull * L.fo_rows.size()));
        HIP_TRY(up(o_nontriv, L.nontrivial.data(), L.nontrivial.size()));
        HIP_TRY(up(o_p4, L.p4_list.data(), 4ull * L.p4_list.size()));
        HIP_TRY(up(o_p4b, p4_b.data(), 4ull * p4_b.size()));
        HIP_TRY(up(o_p4s, p4_s.data(), 4ull * p4_s.size()));
        HIP_TRY(up(o_cls, L.cls_list.data(), 4ull * L.cls_list.size()));
        HIP_TRY(up(o_tbig, tbig.data(), 2ull * tbig.size()));
        HIP_TRY(up(o_bigrows, bigrows.data(), 4ull * bigrows.size()));
        HIP_TRY(up(o_long, long_list.data(), 4ull * long_list.size()));
        HIP_TRY(up(o_p5r, L.p5_rows.data(), 4ull * L.p5_rows.size()));
        HIP_TRY(up(o_p5y, L.p5_y.data(), 4ull * L.p5_y.size()));
        HIP_TRY(up(o_rec, rec.data(), 4ull * rec.size()));
        HIP_TRY(up(o_foi, foi.data(), 4ull * foi.size()));
    } else {
        const int rc = fe::mark_bytes(device, S.dev.lay->dst.nontrivial, marks);
        if (rc != K_OK) return rc;
    }
    HIP_TRY(up(o_spinptr, L.sp_in_ptr.data(), 4ull * L.sp_in_ptr.size()));
    HIP_TRY(up(o_spin, L.sp_in.data(), 4ull * L.sp_in.size()));
    HIP_TRY(up(o_spoutptr, L.sp_out_ptr.data(), 4ull * L.sp_out_ptr.size()));
    HIP_TRY(up(o_spout, L.sp_out.data(), 4ull * L.sp_out.size()));
    HIP_TRY(up(o_spkind, L.sp_kind.data(), L.sp_kind.size()));
    HIP_TRY(up(o_k1, k1_list.data(), 4ull * k1_list.size()));
    HIP_TRY(up(o_k2, k2_list.data(), 4ull * k2_list.size()));
    HIP_TRY(up(o_knowns, L.knowns.data(), 4ull * L.knowns.size()));
    HIP_TRY(up(o_targets, L.targets.data(), 4ull * L.targets.size()));
    if (!L.oob_blob.empty()) HIP_TRY(up(o_oob, L.oob_blob.data(), 4ull * L.oob_blob.size()));
    HIP_TRY(up_flush());
    Job& J = S.dev.job;
    std::memset(&J, 0, sizeof J);
    J.oob = L.oob_blob.empty() ? nullptr : (const uint32_t*)(base + o_oob);
    J.nC = nC; J.nV = nV; J.nSp = nSp;
    J.nKnown = (uint32_t)L.knowns.size(); J.nTarget = (uint32_t)L.targets.size();
    J.nP4 = L.n_p4; J.nP5 = L.n_p5;
    J.qmask = qcap - 1; J.htmask = htcap - 1; J.hotcap = hotcap;
    if (!dev_layout) {
        J.rpA = (const uint32_t*)(base + o_rp[0]); J.rpB = (const uint32_t*)(base + o_rp[1]); J.rpC = (const uint32_t*)(base + o_rp[2]);
        J.colA = (const uint32_t*)(base + o_col[0]); J.colB = (const uint32_t*)(base + o_col[1]); J.colC = (const uint32_t*)(base + o_col[2]);
        J.coefA = (const uint64_t*)(base + o_coef[0]); J.coefB = (const uint64_t*)(base + o_coef[1]); J.coefC = (const uint64_t*)(base + o_coef[2]);
        J.rinfo = (RowInfo*)(base + o_rinfo);
        J.fo_ptr = (const uint32_t*)(base + o_foptr); J.fo_rows = (const uint32_t*)(base + o_forows);
        J.nontrivial = (const uint8_t*)(base + o_nontriv);
        J.p4_list = (const uint32_t*)(base + o_p4);
        J.p4_b = (const uint32_t*)(base + o_p4b); J.p4_s = (const uint32_t*)(base + o_p4s);
        J.cls_list = (const uint32_t*)(base + o_cls);
        J.p5_rows = (const uint32_t*)(base + o_p5r); J.p5_y = (const uint32_t*)(base + o_p5y);
        J.long_list = (const uint32_t*)(base + o_long);
        J.tbig = (const uint16_t*)(base + o_tbig); J.bigrows = (const uint32_t*)(base + o_bigrows);
        J.rec = chain ? (const uint32_t*)(base + o_rec) : nullptr;
        J.foi = chain ? (const uint32_t*)(base + o_foi) : nullptr;
    } else {
        const fe::LayoutDst& D = S.dev.lay->dst;
        J.rpA = D.rp[0]; J.rpB = D.rp[1]; J.rpC = D.rp[2];
        J.colA = D.col[0]; J.colB = D.col[1]; J.colC = D.col[2];
        J.coefA = D.coef[0]; J.coefB = D.coef[1]; J.coefC = D.coef[2];
        J.rinfo = D.rinfo;
        J.fo_ptr = D.fo_ptr; J.fo_rows = D.fo_rows;
        J.nontrivial = D.nontrivial;
        J.p4_list = D.p4_list; J.p4_b = D.p4_b; J.p4_s = D.p4_s;
        J.cls_list = D.cls_list;
        J.p5_rows = D.p5_rows; J.p5_y = D.p5_y;
        J.long_list = D.long_list;
        J.tbig = D.tbig; J.bigrows = D.bigrows;
        J.rec = chain ? D.rec : nullptr;
        J.foi = chain ? D.foi : nullptr;
    }
    J.csort = (uint32_t*)(base + o_csort);
    J.vals = (uint64_t*)(base + o_vals);
    J.sp_in_ptr = (const uint32_t*)(base + o_spinptr); J.sp_in = (const uint32_t*)(base + o_spin);
    J.sp_out_ptr = (const uint32_t*)(base + o_spoutptr); J.sp_out = (const uint32_t*)(base + o_spout);
    J.sp_kind = (const uint8_t*)(base + o_spkind);
    J.k1_list = (const uint32_t*)(base + o_k1); J.k2_list = (const uint32_t*)(base + o_k2);
    J.nK1 = (uint32_t)k1_list.size(); J.nK2 = (uint32_t)k2_list.size();
    J.knowns = (const uint32_t*)(base + o_knowns); J.targets = (const uint32_t*)(base + o_targets);
    J.nBigCls = L.n_cls;
    J.flags = (uint8_t*)(base + o_flags); J.abz = (int32_t*)(base + o_abz);
    J.lb = (uint64_t*)(base + o_lb); J.ub = (uint64_t*)(base + o_ub);
    J.nvalues = (uint8_t*)(base + o_nvalues); J.values = (uint64_t*)(base + o_values);
    J.inq = (uint16_t*)(base + o_inq); J.solved = (uint8_t*)(base + o_solved); J.flip3 = (uint8_t*)(base + o_flip3);
    J.queue = (uint32_t*)(base + o_queue);
    J.varmin = (uint32_t*)(base + o_varmin);
    J.rdead = (uint8_t*)(base + o_rdead);
    J.nLong = (uint32_t)n_long;
    J.nBigRows = (uint32_t)n_bigrows;
    J.p3stamp = (uint32_t*)(base + o_p3stamp);
    J.cls_defer = (uint32_t*)(base + o_clsdefer);
    J.p3k = (uint8_t*)(base + o_p3k); J.p3h = (uint64_t*)(base + o_p3h); J.p3h2 = (uint64_t*)(base + o_p3h2);
    J.ht_key = (uint64_t*)(base + o_htkey); J.ht_key2 = (uint64_t*)(base + o_htkey2);
    J.ht_new = (uint32_t*)(base + o_htnew); J.ht_frozen = (uint32_t*)(base + o_htfrozen);
    J.ht_list = (uint32_t*)(base + o_htlist);
    J.hot = (uint32_t*)(base + o_hot); J.fired = (uint8_t*)(base + o_fired); J.events = (uint32_t*)(base + o_events);
    for (int p = 0; p < 6; ++p) J.dmk[p] = (uint32_t*)(base + o_dmk[p]);
    J.drain = 0;
    J.subteam = 0;
    J.wmarkU = (uint32_t*)(base + o_wmark); J.wmarkB = (uint32_t*)(base + o_wmarkB); J.best = (uint32_t*)(base + o_best); J.prank = (uint32_t*)(base + o_prank);
    J.evbuf = (uint32_t*)(base + o_evbuf); J.evcnt = (uint32_t*)(base + o_evcnt); J.cand = (uint32_t*)(base + o_cand);
    J.candcap = (uint32_t)std::max<size_t>(ECNE_CANDCAP, 8ull * nC);
    J.fvar = (uint32_t*)(base + o_fvar); J.frank = (uint32_t*)(base + o_frank); J.fbase = (uint32_t*)(base + o_fbase);
    J.bigev = (uint32_t*)(base + o_bigev);
    J.bigpool = (uint32_t*)(base + o_bigpool); J.bigstride = bigstride;
    J.ctr = (Counters*)(base + o_ctr);
    J.lds_flags_off = J.lds_inq_off = J.lds_flip_off = J.lds_w2_off = J.lds_w2b_off = 0xFFFFFFFFu;
    // (the L2 warm-up streams one contiguous range: the arena's static part; a device-laid system's static arrays are elsewhere)
    J.warm_bytes = (!dev_layout && (static_end - o_rp[0]) <= (7u << 19)) ? (uint32_t)(static_end - o_rp[0]) : 0u;   // fits one XCD's 4 MB L2 beside the state
    S.dev.classified = false;
    S.dev.arena = base;
    S.dev.arena_bytes = c.off;
    S.dev.device = device;
    arena_guard.done = true;
    return K_OK;
}

// k_classify_rows on one system (idempotent); records the HIP-event time
static int classify_system(ecne_system& S, hipStream_t stream, Job* d_job_slot) {
    if (S.dev.classified) return K_OK;
    HIP_TRY(hipMemcpyAsync(d_job_slot, &S.dev.job, sizeof(Job), hipMemcpyHostToDevice, stream));
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    uint32_t nblk0 = (S.L.nC + 255) / 256;
    uint32_t blk_cap = 256 * 16;               // >> 256 workgroups: fills all 8 XCDs, grid-strides the rest
    if (const char* e = getenv("ECNE_CLS_BLOCKS")) blk_cap = (uint32_t)atoi(e);   // experiment hook
    if (nblk0 > blk_cap) nblk0 = blk_cap;
    if (nblk0 == 0) nblk0 = 1;
    uint32_t nblk1 = (S.L.n_cls + 3) / 4;
    if (nblk1 > 256 * 16) nblk1 = 256 * 16;
    HIP_TRY(hipMemsetAsync(S.dev.job.cls_defer, 0, 4, stream));
    HIP_TRY(hipEventRecord(e0, stream));
    // two launches: the streaming pass (one lane per row with at most three entries in C), then one wavefront per row for the rows the
    // layout listed (cls_list: more entries) and the ones the streaming lanes deferred (a divisor that needs a real inversion: usually none).
    // (Tried: the listed rows on a stream of their own next to the streaming pass -- 0.143 -> 0.140 ms at 2.8 M rows, 0.053 -> 0.054 at 0.7 M:
    //  not worth a second stream, two events and a lock.)
    // (round 6: the deferred short rows get a LANE each in the second launch -- with the inversion --, dealt over at least one workgroup per CU)
    hipLaunchKernelGGL(k_classify_rows, dim3(nblk0), dim3(256), 0, stream, (const Job*)d_job_slot, 0u);
    hipLaunchKernelGGL(k_classify_wave, dim3(std::max<uint32_t>(nblk1, 1024u)), dim3(256), 0, stream, (const Job*)d_job_slot, 0u);
    HIP_TRY(hipEventRecord(e1, stream));
    HIP_TRY(hipEventSynchronize(e1));
    HIP_TRY(hipGetLastError());
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    S.dev.classify_ms = ms;
    S.dev.classified = true;
    return K_OK;
}

// download the per-variable state of a finished solve (lazy; see ecne_result)
static int fetch_states(ecne_result* r) {
    if (r->have_states) return K_OK;
    if (!r->sys || !system_is_live(r->sys) || r->sys->uid != r->sys_uid || r->sys->generation != r->generation || !r->sys->dev.arena) return K_EINVAL;
    ecne_system& S = *r->sys;
    const Layout& L = S.L;
    const Job& J = S.dev.job;
    RestoreDevice restore;
    HIP_TRY(hipSetDevice(S.dev.device));
    const size_t nv = (size_t)S.n_vars;
    std::vector<uint8_t> flags(L.nV + 1), nvals(L.nV + 1);
    std::vector<uint64_t> lb(4ull * (L.nV + 1)), ub(4ull * (L.nV + 1)), vals(8ull * (L.nV + 1));
    std::vector<int32_t> abz(L.nV + 1);
    HIP_TRY(hipMemcpy(flags.data(), J.flags, flags.size(), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(nvals.data(), J.nvalues, nvals.size(), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(lb.data(), J.lb, lb.size() * 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(ub.data(), J.ub, ub.size() * 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(vals.data(), J.values, vals.size() * 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(abz.data(), J.abz, abz.size() * 4, hipMemcpyDeviceToHost));
    // re-base to "variable v at index v-1", n_vars entries
    r->flags.assign(nv, 0); r->nvalues.assign(nv, 0); r->abz.assign(nv, -1);
    r->lb.assign(4 * nv, 0); r->ub.assign(4 * nv, 0); r->values.assign(8 * nv, 0);
    for (size_t v = 1; v <= nv && v <= L.nV; ++v) {
        r->flags[v - 1] = flags[v] & 3;
        r->nvalues[v - 1] = nvals[v];
        r->abz[v - 1] = abz[v];
        std::memcpy(&r->lb[4 * (v - 1)], &lb[4 * v], 32);
        std::memcpy(&r->ub[4 * (v - 1)], &ub[4 * v], 32);
        if (nvals[v] >= 1) std::memcpy(&r->values[8 * (v - 1)], &vals[8 * v], 32);
        if (nvals[v] >= 2) std::memcpy(&r->values[8 * (v - 1) + 4], &vals[8 * v + 4], 32);
    }
    // "Bad Constraints": rows with a variable that is not uniquely determined (:1609-1618) -- flagged and compacted on the device
    {
        const int rc = fe::bad_rows(S.dev.device, J, r->bad_rows);
        if (rc != K_OK) return rc;
    }
    r->have_states = true;
    return K_OK;
}

// ------------------------------------------------------------------------------------ C ABI
extern "C" {

static int ecne_r1cs_load_impl(const char* path, ecne_r1cs** out) {
    if (!path || !out) return ECNE_EINVAL;
    *out = nullptr;
    AbiTick tick;
    std::unique_ptr<ecne_r1cs> r(new ecne_r1cs());
    FileView fv(path);
    size_t cons_off = 0;
    int st = read_r1cs_header(fv, r->f, cons_off);
    if (st != K_OK) return st;
    tick("load: open + map + header");
    r->f.path = path;
    g_fe_stats.parse = fe::ParseStats();
    g_fe_stats.parse_dev = 0;
    bool done = false;
    if (frontend_wants_device(r->f.n_cons)) {
        // the constraint section goes to the device as it is; rows come out in the reference's dictionary order there
        st = fe::parse_on_device(fv.data, fv.size, cons_off, r->f.n_cons, current_device(), r->drows, g_fe_stats.parse);
        if (st == K_OK) {
            for (int p = 0; p < 3; ++p) r->f.nnz[p] = r->drows->nnz[p];
            g_fe_stats.parse_dev = 1;
            done = true;
        } else if (!fe_falls_back(st)) return st;
        else r->drows.reset();
        tick("load: parse_on_device");
    }
    if (!done) {
        st = read_r1cs_rows(fv, cons_off, path, r->f);
        if (st != K_OK) return st;
    }
    *out = r.release();
    return ECNE_OK;
}
int ecne_r1cs_info(const ecne_r1cs* f, ecne_info* o) {
    if (!f || !o) return ECNE_EINVAL;
    o->field_size = f->f.field_size; o->n_wires = f->f.n_wires; o->n_pub_out = f->f.n_pub_out;
    o->n_pub_in = f->f.n_pub_in; o->n_prv_in = f->f.n_prv_in; o->n_constraints = f->f.n_cons;
    o->n_labels = f->f.n_labels;
    for (int p = 0; p < 3; ++p) o->nnz[p] = f->f.nnz[p];
    o->n_vars = f->f.n_vars;
    return ECNE_OK;
}
static int ecne_r1cs_csr_impl(const ecne_r1cs* f, int part, const uint64_t** rowptr, const uint32_t** col, const uint64_t** coeff) {
    if (!f || part < 0 || part > 2) return ECNE_EINVAL;
    if (!f->f.csr_built) {   // (a handle is used by one thread at a time, include/ecne.h)
        const int st = build_file_csr(const_cast<ecne_r1cs*>(f)->f);
        if (st != K_OK) return st;
    }
    if (rowptr) *rowptr = f->f.csr_ptr[part].data();
    if (col) *col = f->f.csr_col[part].data();
    if (coeff) *coeff = f->f.csr_coef[part].data();
    return ECNE_OK;
}
int ecne_r1cs_io(const ecne_r1cs* f, const int64_t** known, size_t* nk, const int64_t** targets, size_t* nt) {
    if (!f) return ECNE_EINVAL;
    if (known) *known = f->f.knowns.data();
    if (nk) *nk = f->f.knowns.size();
    if (targets) *targets = f->f.outputs.data();
    if (nt) *nt = f->f.outputs.size();
    return ECNE_OK;
}
void ecne_r1cs_free(ecne_r1cs* f) { delete f; }

static int ecne_system_from_r1cs_impl(const ecne_r1cs* m, ecne_system** out) {
    if (!m || !out) return ECNE_EINVAL;
    ecne_system* s = new ecne_system();
    s->base = m->file;
    s->drows = m->drows;
    s->cur = m->f.host_rows ? &s->base->rows : nullptr;
    s->knowns = m->f.knowns;
    s->targets = m->f.outputs;
    s->n_vars = m->f.n_vars;
    s->n_rows_main = (int64_t)m->f.n_cons;
    { std::lock_guard<std::mutex> g(g_live_mu); live_systems().insert(s); }
    *out = s;
    return ECNE_OK;
}
// statistics of the calling thread's last ecne_abstract (ecne_abstract_stats)
struct AbstractStats { double fp_ms = 0, scan_ms = 0, bytes = 0, upload_ms = 0; int used_device = 0; size_t n_cand = 0; };
static thread_local AbstractStats g_last_abstract;

static int ecne_abstract_impl(ecne_system* sys, const ecne_r1cs* trusted, const char* name) {
    if (!sys || !trusted || !name) return ECNE_EINVAL;
    AbiTick tick;
    { const int rc = ensure_host_rows(trusted); if (rc != K_OK) return rc; }      // the pattern side is prepared on the host (small)
    tick("abstract: host rows of the trusted file");
    g_last_abstract = AbstractStats();
    g_fe_stats.abs = fe::AbstractDevStats();
    g_fe_stats.abs_dev = 0;
    // rows on a device (device front-end): fingerprints, window scan, exact verification and compaction all happen there
    if (sys->drows && frontend_setting().load(std::memory_order_relaxed) != 0) {
        std::shared_ptr<fe::DevRows> red;
        std::vector<Special> fresh = sys->specials;
        const int rc = fe::abstract_on_device(name, sys->drows, trusted->f, fresh, red, g_fe_stats.abs);
        tick("abstract: abstract_on_device");
        if (rc == K_OK) {
            system_changed_rows(sys);   // layout and device image are stale, earlier results unreadable
            sys->specials = std::move(fresh);
            if (red != sys->drows) {
                sys->drows = red;
                sys->reduced = Rows();
                sys->cur = nullptr;
                sys->base.reset();
            }
            g_fe_stats.abs_dev = 1;
            g_last_abstract.used_device = 1;
            g_last_abstract.fp_ms = g_fe_stats.abs.fp_ms; g_last_abstract.scan_ms = g_fe_stats.abs.scan_ms;
            g_last_abstract.bytes = (double)g_fe_stats.abs.bytes; g_last_abstract.n_cand = g_fe_stats.abs.n_cand;
            tick("abstract: handle update");
            return K_OK;
        }
        if (!fe_falls_back(rc)) return rc;
    }
    { const int rc = sys_host_rows(*sys); if (rc != K_OK) return rc; }
    Rows red;
    // The candidate scan runs on the GPU for files worth the upload (ECNE_ABSTRACT_DEVICE=0 / 1 forces host / device);
    // verification and replacement are the same host code either way, so the result does not depend on the choice.
    std::vector<size_t> dcand;
    bool have_dcand = false;
    {
        const char* e = getenv("ECNE_ABSTRACT_DEVICE");
        const bool want = e ? atoi(e) != 0 : sys->rows().n() >= ECNE_ABSTRACT_DEVICE_ROWS;
        if (want && ecne_device_count() > 0) {
            fe::AbstractDevStats ds;
            double up_ms = 0;
            have_dcand = fe::candidates_for_host_rows(sys->rows(), trusted->f.rows, current_device(), dcand, ds, up_ms) == K_OK;
            if (have_dcand) {
                g_last_abstract.used_device = 1; g_last_abstract.fp_ms = ds.fp_ms; g_last_abstract.scan_ms = ds.scan_ms;
                g_last_abstract.bytes = (double)ds.bytes; g_last_abstract.upload_ms = up_ms; g_last_abstract.n_cand = dcand.size();
            }
        }
    }
    const int rc = abstract_one(name, sys->rows(), trusted->f, sys->specials, red, have_dcand ? &dcand : nullptr);
    if (rc != K_OK) return rc;
    system_changed_rows(sys);   // layout and device image are stale, earlier results unreadable
    sys->reduced = std::move(red);
    sys->cur = &sys->reduced;
    sys->base.reset();
    sys->drows.reset();         // (the device copy, if any, holds the rows before this abstraction)
    return K_OK;
}
static int ensure_layout(ecne_system& S);
static int ecne_system_info_get_impl(const ecne_system* sys, ecne_system_info* o) {
    if (!sys || !o) return ECNE_EINVAL;
    ecne_system* s = const_cast<ecne_system*>(sys);
    { const int rc = ensure_layout(*s); if (rc != K_OK) return rc; }
    o->n_rows = (int64_t)s->L.nC;
    o->n_rows_main = sys->n_rows_main;
    o->n_vars = sys->n_vars;
    o->n_specials = (int64_t)sys->specials.size();
    o->n_known = (int64_t)sys->knowns.size();
    o->n_targets = (int64_t)sys->targets.size();
    for (int p = 0; p < 3; ++p) o->nnz[p] = s->L.nnz[p];
    return ECNE_OK;
}
int ecne_system_special(const ecne_system* sys, int64_t idx, const char** name, const int64_t** in, size_t* nin,
                        const int64_t** outv, size_t* nout) {
    if (!sys || idx < 0 || (size_t)idx >= sys->specials.size()) return ECNE_EINVAL;
    const Special& sp = sys->specials[(size_t)idx];
    if (name) *name = sp.name.c_str();
    if (in) *in = sp.inputs.data();
    if (nin) *nin = sp.inputs.size();
    if (outv) *outv = sp.outputs.data();
    if (nout) *nout = sp.outputs.size();
    return ECNE_OK;
}
// the system laid out (flat arrays + counts) by whichever front-end it belongs to, without solving it
static int ensure_layout(ecne_system& S) {
    if (S.laid_out) return K_OK;
    if (S.drows && frontend_setting().load(std::memory_order_relaxed) != 0 && ecne_device_count() > S.drows->device) {
        const int prev = current_device();
        const int rc = upload_system(S, S.drows->device);
        (void)hipSetDevice(prev);
        return rc;
    }
    const int rc = sys_host_rows(S);
    if (rc != K_OK) return rc;
    build_layout(S);
    return K_OK;
}
static int ecne_system_rows_impl(ecne_system* sys, int part, const uint32_t** rowptr, const uint32_t** col, const uint64_t** coeff) {
    if (!sys || part < 0 || part > 2) return ECNE_EINVAL;
    { const int rc = ensure_layout(*sys); if (rc != K_OK) return rc; }
    Layout& L = sys->L;
    if (!L.host_arrays && L.rp[0].size() != (size_t)L.nC + 1) {
        // laid out on the device: fetch the three CSR parts (once)
        if (!sys->dev.lay) return ECNE_EINVAL;
        const fe::LayoutDst& D = sys->dev.lay->dst;
        RestoreDevice restore;
        HIP_TRY(hipSetDevice(sys->dev.lay->device));
        // (fetched into temporaries: the handle's arrays count as fetched by their size, so they only change once every copy has succeeded)
        std::vector<uint32_t, RawAlloc<uint32_t>> rp[3], col[3];
        std::vector<uint64_t, RawAlloc<uint64_t>> coef[3];
        for (int p = 0; p < 3; ++p) {
            rp[p].resize((size_t)L.nC + 1);
            col[p].resize(L.nnz[p]);
            coef[p].resize(4 * L.nnz[p]);
            HIP_TRY(hipMemcpy(rp[p].data(), D.rp[p], 4ull * ((size_t)L.nC + 1), hipMemcpyDeviceToHost));
            if (L.nnz[p]) {
                HIP_TRY(hipMemcpy(col[p].data(), D.col[p], 4ull * L.nnz[p], hipMemcpyDeviceToHost));
                HIP_TRY(hipMemcpy(coef[p].data(), D.coef[p], 32ull * L.nnz[p], hipMemcpyDeviceToHost));
            }
        }
        for (int p = 0; p < 3; ++p) { L.rp[p].swap(rp[p]); L.col[p].swap(col[p]); L.coef[p].swap(coef[p]); }
    }
    if (rowptr) *rowptr = L.rp[part].data();
    if (col) *col = L.col[part].data();
    if (coeff) *coeff = L.coef[part].data();
    return ECNE_OK;
}
void ecne_system_free(ecne_system* sys) {
    if (!sys) return;
    { std::lock_guard<std::mutex> g(g_live_mu); live_systems().erase(sys); }
    delete sys;
}

int ecne_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

#define K_ESPLIT (-1000)      // internal: the split launch cannot be made (a part needs more than one workgroup, the parts do not fit the device)
// sl: sys[0 .. n-2] are the parts of sys[n-1] (SplitPlan): the parts are solved, in lockstep, and scattered into the file's arrays;
// the file itself gets no workgroup. fam_flags[0 / 1]: a part left early / the constant wire's state moved.
// the side launch of the calling thread's last solve_batch_core (a team's kernel and k_solve at the same time on two streams); a batch whose
// team timed out at its barrier under it is solved once more without (ecne_solve_batch)
static thread_local bool tl_side_used = false, tl_side_forbid = false;
static int solve_batch_core(ecne_system** sys, size_t n, const ecne_opts* opts, ecne_result** out, SplitPlan* sl, uint32_t* fam_flags) {
    tl_side_used = false;
    if (!sys || !out || n == 0) return ECNE_EINVAL;
    for (size_t i = 0; i < n; ++i) out[i] = nullptr;
    ecne_opts o;
    std::memset(&o, 0, sizeof o);
    if (opts) o = *opts;
    if (ecne_device_count() <= o.device) return ECNE_ENODEVICE;   // never falls back to a CPU path
    RestoreDevice restore;      // the caller's current device is left as it was
    HIP_TRY(hipSetDevice(o.device));
    hipStream_t stream = (hipStream_t)o.stream;
    for (size_t i = 0; i < n; ++i) {
        if (!sys[i]) return ECNE_EINVAL;
        int st = upload_system(*sys[i], o.device);
        if (st != K_OK) return st;
    }
    int n_cu = 0;
    if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, o.device) != hipSuccess || n_cu < 1) return ECNE_ENODEVICE;
    const uint32_t cap = (uint32_t)std::max(1, n_cu - 8);   // margin: never rely on the last CU being free
    // job descriptors, workgroup table and the two events live in the calling thread's launch scratch (no allocation per solve)
    LaunchScratch& scratch = launch_scratch();
    { const int st = scratch.prepare(o.device, n, cap); if (st != K_OK) return st; }
    Job* const d_jobs = scratch.d_jobs;
    std::vector<Job> hj(n);
    int rc = ECNE_OK;
    do {
        std::vector<uint8_t> seq_only(n, 0);
        for (size_t i = 0; i < n; ++i) {
            int st = classify_system(*sys[i], stream, d_jobs);
            if (st != K_OK) { rc = st; break; }
            hj[i] = sys[i]->dev.job;
            hj[i].secp_solve = (sys[i]->secp_solve_override >= 0 ? sys[i]->secp_solve_override != 0 : o.secp_solve != 0) ? 1u : 0u;
            {   // barrier wait without progress: 0.2 s + 2 us per row (a long sequential stretch on a huge system is not a hang), or
                // ECNE_BARRIER_TIMEOUT_MS
                uint64_t tmo = 200 + (uint64_t)hj[i].nC / 500;
                if (const char* e = getenv("ECNE_BARRIER_TIMEOUT_MS")) tmo = (uint64_t)std::max(1L, atol(e));
                hj[i].bar_timeout_ms = (uint32_t)std::min<uint64_t>(tmo, 3600000);
            }
            // (ids above num_variables: strictly sequential pops, oob.hip.hpp. The same for a caller's known_variables WITHOUT the constant
            //  wire (:682-693 then leave variable 1 like any other unknown): every parallel schedule relies on its `unique` / `is_known` never
            //  changing and pads short rows with it)
            const bool wire_free = std::find(sys[i]->knowns.begin(), sys[i]->knowns.end(), (int64_t)1) == sys[i]->knowns.end();
            seq_only[i] = hj[i].oob != nullptr || wire_free;
            hj[i].queue_mode = seq_only[i] ? 1u : (uint32_t)o.queue_mode;
            {   // rounds on all workgroups: drain rounds (drain.hip.hpp) unless queue_mode 3 / ECNE_DRAIN=0 ask for the prefix rounds
                static const int drain_env = []() { const char* e = getenv("ECNE_DRAIN"); return e ? atoi(e) : 1; }();      // 2: test hook (every frontier is drained)
                static const bool solo_off = []() { const char* e = getenv("ECNE_SOLO"); return e && atoi(e) == 0; }();
                hj[i].drain = o.queue_mode == 3 ? 0u : o.queue_mode == 4 ? 3u : drain_env <= 0 ? 0u : drain_env >= 2 ? 3u : 1u;      // bit 0: drain rounds, bit 1: every frontier, bit 2: no solo drains
                if (hj[i].drain && solo_off) hj[i].drain |= 4u;
                static const bool sub_off = []() { const char* e = getenv("ECNE_SUBTEAM"); return e && atoi(e) == 0; }();      // bit 3: rounds always on all workgroups
                if (sub_off) hj[i].drain |= 8u;
                static const bool lv_off = []() { const char* e = getenv("ECNE_LEVEL"); return e && atoi(e) == 0; }();      // level rounds (level.hip.hpp) off: A/B runs
                static const bool crew_off = []() { const char* e = getenv("ECNE_CREW"); return e && atoi(e) == 0; }();      // crew rounds (crew.hip.hpp) off: A/B runs
                static const bool r4d_off = []() { const char* e = getenv("ECNE_R4DONE"); return e && atoi(e) == 0; }();      // long_r4_done (fastrow.hip.hpp) off: A/B runs
                hj[i].lv_off = (lv_off ? 1u : 0u) | (crew_off ? 2u : 0u) | (r4d_off ? 4u : 0u) | (wire_free ? 8u : 0u);      // bit 3: known_variables without the constant wire (R2 as the reference states it, rules_wave.hip.hpp)
            }
            hj[i].family = nullptr; hj[i].fam_rank = 0; hj[i].fam_size = 0;
            if (sl && i + 1 < n) { hj[i].family = sl->d_family; hj[i].fam_rank = (uint32_t)i; hj[i].fam_size = (uint32_t)(n - 1); }
        }
        if (rc != ECNE_OK) break;
        if (sl && hipMemsetAsync(sl->d_family, 0, sizeof(Family), stream) != hipSuccess) { rc = ECNE_ENODEVICE; break; }
        // Workgroups per job: large systems get helpers for the row-parallel sweep passes. All
        // workgroups of one launch must be co-resident (they meet at a hand-rolled barrier), and
        // k_solve occupies a whole CU per workgroup (512 threads x 256 VGPRs), so a launch never
        // holds more workgroups than the device has CUs; a batch that needs more is split.
        // dynamic LDS: whatever the CU has beyond k_solve's static tables (a workgroup owns its CU anyway);
        // single-workgroup jobs keep their hot state there (k_solve, "LDS residency")
        uint32_t dyn_lds = 0;
        {
            // (two kernels, k_solve for single-workgroup jobs and k_solve_team: the same dynamic LDS for both)
            hipFuncAttributes fa, fb;
            int lds_max = 0;
            if (hipFuncGetAttributes(&fa, (const void*)k_solve) == hipSuccess && hipFuncGetAttributes(&fb, (const void*)k_solve_team) == hipSuccess &&
                hipDeviceGetAttribute(&lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock, o.device) == hipSuccess &&
                (size_t)lds_max > std::max(fa.sharedSizeBytes, fb.sharedSizeBytes) + 1024) {
                dyn_lds = ((uint32_t)lds_max - (uint32_t)std::max(fa.sharedSizeBytes, fb.sharedSizeBytes) - 256u) & ~255u;
                if (const char* e = getenv("ECNE_LDS_BYTES")) dyn_lds = std::min<uint32_t>(dyn_lds, (uint32_t)atoi(e));   // test hook
                if (dyn_lds && (hipFuncSetAttribute((const void*)k_solve, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn_lds) != hipSuccess ||
                                hipFuncSetAttribute((const void*)k_solve_team, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn_lds) != hipSuccess)) dyn_lds = 0;
            }
            (void)hipGetLastError();
        }
        uint32_t single_wg_rows = ECNE_ROWS_PER_WG;
        if (const char* e = getenv("ECNE_SINGLE_WG_ROWS")) single_wg_rows = (uint32_t)atoi(e);   // experiment hook
        for (size_t i = 0; i < n; ++i) {
            uint32_t want = (hj[i].nC + ECNE_ROWS_PER_WG - 1u) / ECNE_ROWS_PER_WG;
            // a system whose flags and in_queue tags fit the LDS is solved by ONE workgroup (chain executor, chain.hip.hpp)
            const bool lds_fits = hj[i].rec && hj[i].nC <= ECNE_CHAIN_ROWS && ((size_t)hj[i].nV + 1 + 16) + (2ull * hj[i].nC + 16) + 8192 <= dyn_lds;
            if (hj[i].nC <= single_wg_rows || (lds_fits && !getenv("ECNE_SINGLE_WG_ROWS"))) want = 1;
            if (o.debug > 0) want = (uint32_t)o.debug;          // test hook: force the helper count
            if (seq_only[i]) want = 1;
            hj[i].nwg = std::max<uint32_t>(1u, std::min<uint32_t>(want, std::min<uint32_t>(cap, (uint32_t)ECNE_MAX_NWG)));
            hj[i].lds_bytes = dyn_lds;
            if (sl && i + 1 < n && hj[i].nwg != 1) rc = K_ESPLIT;      // (a part is a single-workgroup job)
            if (sl && i + 1 == n) hj[i].nwg = 0;                       // the file itself: its arrays receive the parts' states
        }
        if (sl && n - 1 > cap) rc = K_ESPLIT;
        if (rc != ECNE_OK) break;
        if (hipMemcpyAsync(d_jobs, hj.data(), sizeof(Job) * n, hipMemcpyHostToDevice, stream) != hipSuccess) { rc = ECNE_ENODEVICE; break; }
        hipLaunchKernelGGL(k_zero_counters, dim3((unsigned)n), dim3(256), 0, stream, (const Job*)d_jobs);
        // every launch's workgroup descriptors in one upload (a batch of more jobs than the device has CUs is several launches back to back)
        size_t total_wg = 0;
        for (size_t i = 0; i < n; ++i) total_wg += hj[i].nwg;
        { const int st = scratch.prepare(o.device, n, std::max<size_t>(cap, total_wg)); if (st != K_OK) { rc = st; break; } }
        const hipEvent_t e0 = scratch.e0, e1 = scratch.e1;
        {
            // a launch that holds a multi-workgroup job must have the device to itself (its workgroups meet at a barrier)
            bool any_multi = sl != nullptr;      // (parts meet at their family barrier: resident together, the device to themselves)
            for (size_t i = 0; i < n; ++i) any_multi |= hj[i].nwg > 1;
            std::unique_lock<std::shared_mutex> exclusive_lock(device_launch_mutex(o.device), std::defer_lock);
            std::shared_lock<std::shared_mutex> shared_lock(device_launch_mutex(o.device), std::defer_lock);
            if (any_multi || n > 1) exclusive_lock.lock(); else shared_lock.lock();
            // Co-residency of a multi-workgroup job: the launch is refused unless the device can hold the whole grid at once
            // (occupancy query: workgroups of k_solve per CU with this much dynamic LDS x CUs). ECNE_COOPERATIVE=1 additionally makes
            // the launch through hipLaunchCooperativeKernel (the runtime checks the same thing and uses its cooperative queue) -- not
            // the default: rocprofv3 of ROCm 7.2 crashes at process exit when the application used the cooperative queue.
            int coop_attr = 0;
            const bool coop_ok = getenv("ECNE_COOPERATIVE") && atoi(getenv("ECNE_COOPERATIVE")) != 0 &&
                                 hipDeviceGetAttribute(&coop_attr, hipDeviceAttributeCooperativeLaunch, o.device) == hipSuccess && coop_attr != 0;
            int wg_per_cu = 0;
            const void* const kernel = any_multi ? (const void*)k_solve_team : (const void*)k_solve;      // single-workgroup jobs: the kernel without team code
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&wg_per_cu, kernel, ECNE_WG, dyn_lds) != hipSuccess) { wg_per_cu = 1; (void)hipGetLastError(); }
            const size_t resident_cap = (size_t)std::max(wg_per_cu, 0) * (size_t)n_cu;
            // (the side launch runs k_solve_team and k_solve at the same time: both kernels' occupancy counts -- the same today, 512 threads x 256
            //  VGPRs and the same dynamic LDS, one workgroup per CU)
            int wg_per_cu_side = wg_per_cu;
            if (any_multi && hipOccupancyMaxActiveBlocksPerMultiprocessor(&wg_per_cu_side, (const void*)k_solve, ECNE_WG, dyn_lds) != hipSuccess) { wg_per_cu_side = 1; (void)hipGetLastError(); }
            const size_t resident_cap_both = (size_t)std::max(std::min(wg_per_cu, wg_per_cu_side), 0) * (size_t)n_cu;
            bool refused = false;
            (void)hipEventRecord(e0, stream);
            std::vector<WgDesc> all_descs;
            std::vector<size_t> launch_at;      // first descriptor of every launch, and the end
            // A multi-workgroup job next to single-workgroup ones that all fit the device together (the verification DAG of BASELINE config 5:
            // ecdsa_like on a team, secp256k1 and its two trusted functions on a workgroup each): the single-workgroup jobs go through k_solve
            // -- the kernel without team code, 4-9 % faster on them and a third on secp256k1 (7.2 against 9.7 ms) -- on a stream of their own,
            // at the same time as the team's launch; both kernels' workgroups are resident together (their sum is within `cap`).
            size_t n_side = 0, total_all = 0;
            for (size_t i = 0; i < n; ++i) { total_all += hj[i].nwg; if (hj[i].nwg == 1) ++n_side; }
            static const bool side_off = []() { const char* e = getenv("ECNE_SIDE_LAUNCH"); return e && atoi(e) == 0; }();
            const bool side_launch = !sl && any_multi && n_side > 0 && n_side < n && total_all <= cap && total_all <= resident_cap_both && !coop_ok && !side_off && !tl_side_forbid &&
                                     scratch.side_stream() == K_OK;
            tl_side_used = side_launch;
            if (side_launch) {
                for (size_t i = 0; i < n; ++i) if (hj[i].nwg > 1) for (uint32_t r = 0; r < hj[i].nwg; ++r) all_descs.push_back({(uint32_t)i, r});
                launch_at.push_back(0);
                launch_at.push_back(all_descs.size());
                for (size_t i = 0; i < n; ++i) if (hj[i].nwg == 1) all_descs.push_back({(uint32_t)i, 0u});
            } else
            {
                // (round 5) a batch of single-workgroup jobs only is ONE launch whatever its size: nothing in it waits for another workgroup,
                // the dispatcher starts the next job on whichever CU becomes free (the caller hands the jobs over longest first: jobs.Runner) --
                // 504 mid-depth circuits were three launches of <= 248, each as long as its longest job (1.40 ms of kernel time in all)
                const size_t launch_cap = any_multi ? cap : (size_t)-1;
                size_t i = 0;
                while (i < n) {
                    launch_at.push_back(all_descs.size());
                    size_t in_launch = 0;
                    while (i < n && in_launch + hj[i].nwg <= launch_cap) {
                        for (uint32_t r = 0; r < hj[i].nwg; ++r) all_descs.push_back({(uint32_t)i, r});
                        in_launch += hj[i].nwg;
                        ++i;
                    }
                    if (in_launch == 0 && i < n && hj[i].nwg != 0) break;      // (cannot happen: nwg <= cap)
                }
                launch_at.push_back(all_descs.size());
            }
            bool fail = false;
#ifdef ECNE_JITTER
            {   // (developer build: the seed of this launch's pseudo-random delays, job_barrier.hip.hpp)
                static std::atomic<uint32_t> n_launch{0};
                const char* je = getenv("ECNE_JITTER_SEED");
                const uint32_t seed = (je ? (uint32_t)strtoul(je, nullptr, 0) : 1u) * 2654435761u + n_launch.fetch_add(1) * 40503u;
                if (hipMemcpyToSymbolAsync(HIP_SYMBOL(g_jitter_seed), &seed, 4, 0, hipMemcpyHostToDevice, stream) != hipSuccess) fail = true;
            }
#endif
            if (!all_descs.empty() && hipMemcpyAsync(scratch.d_descs, all_descs.data(), sizeof(WgDesc) * all_descs.size(), hipMemcpyHostToDevice, stream) != hipSuccess) fail = true;
            if (side_launch && !fail && (hipEventRecord(scratch.e_up, stream) != hipSuccess || hipStreamWaitEvent(scratch.side, scratch.e_up, 0) != hipSuccess)) fail = true;
            for (size_t li = 0; li + 1 < launch_at.size() && !fail; ++li) {
                const size_t nd = launch_at[li + 1] - launch_at[li];
                if (nd == 0) continue;
                WgDesc* const d_descs = scratch.d_descs + launch_at[li];
                // the workgroups of a multi-workgroup job meet at a barrier of their own: launched cooperatively, the runtime either
                // makes the whole grid resident together or refuses the launch (no 0.2 s wait for workgroups that never start)
                bool launched = false;
                if (any_multi && nd > resident_cap) { refused = true; fail = true; break; }
                if (any_multi && coop_ok) {
                    const Job* a0 = d_jobs;
                    const WgDesc* a1 = d_descs;
                    void* kargs[2] = {(void*)&a0, (void*)&a1};
                    const hipError_t ce = hipLaunchCooperativeKernel(kernel, dim3((unsigned)nd), dim3(ECNE_WG), kargs, dyn_lds, stream);
                    if (ce == hipSuccess) launched = true;
                    else if (ce == hipErrorCooperativeLaunchTooLarge) { (void)hipGetLastError(); refused = true; fail = true; break; }
                    else (void)hipGetLastError();      // (not available on this stack: the plain launch below, with the barrier's own time bound)
                }
                if (!launched) {
                    if (any_multi) hipLaunchKernelGGL(k_solve_team, dim3((unsigned)nd), dim3(ECNE_WG), dyn_lds, stream, (const Job*)d_jobs, (const WgDesc*)d_descs);
                    else hipLaunchKernelGGL(k_solve, dim3((unsigned)nd), dim3(ECNE_WG), dyn_lds, stream, (const Job*)d_jobs, (const WgDesc*)d_descs);
                }
            }
            if (side_launch && !fail) {
                // (the side stream started behind the uploads and the zeroed counters, its kernel is submitted behind the team's; the caller's stream
                //  goes on behind the side stream's kernel)
                const size_t first = launch_at[1];
                hipLaunchKernelGGL(k_solve, dim3((unsigned)(all_descs.size() - first)), dim3(ECNE_WG), dyn_lds, scratch.side, (const Job*)d_jobs, (const WgDesc*)(scratch.d_descs + first));
                if (hipEventRecord(scratch.e_side, scratch.side) != hipSuccess) fail = true;
            }
            if (side_launch && !fail && hipStreamWaitEvent(stream, scratch.e_side, 0) != hipSuccess) fail = true;
            if (sl && !fail && !refused) {
                // the parts' states into the file's arrays, the file's verdict counts (inside the timed region: part of the solve)
                const Job& PJ = hj[n - 1];
                for (size_t k = 0; k + 1 < n; ++k) {
                    const uint32_t nvk = (uint32_t)sys[k]->n_vars;
                    hipLaunchKernelGGL(k_scatter_part, dim3(std::max(1u, std::min(1024u, (nvk + 255u) / 256u))), dim3(256), 0, stream, hj[k], PJ, (const uint32_t*)sl->d_map[k], nvk);
                }
                hipLaunchKernelGGL(k_part_counts, dim3(std::max(1u, std::min(1024u, (PJ.nV + 255u) / 256u))), dim3(256), 0, stream, PJ);
            }
            (void)hipEventRecord(e1, stream);
            if (!fail && !refused) hipLaunchKernelGGL(k_gather_results, dim3((unsigned)n), dim3(256), 0, stream, (const Job*)d_jobs, scratch.d_res);
            if (refused) { rc = ECNE_ETIMEOUT; break; }      // the device cannot hold the job's workgroups together right now
            if (fail || hipEventSynchronize(e1) != hipSuccess || hipGetLastError() != hipSuccess) { rc = ECNE_ENODEVICE; break; }
        }
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned char> h_res(ECNE_RESULT_BYTES * n);
        if (hipMemcpyAsync(h_res.data(), scratch.d_res, h_res.size(), hipMemcpyDeviceToHost, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess) { rc = ECNE_ENODEVICE; break; }
        if (sl && fam_flags) {
            Family hf;
            if (hipMemcpy(&hf, sl->d_family, sizeof(Family), hipMemcpyDeviceToHost) != hipSuccess) { rc = ECNE_ENODEVICE; break; }
            fam_flags[0] = hf.abort; fam_flags[1] = hf.var1_bad;
        }
        for (size_t i = 0; i < n; ++i) {
            ecne_system& S = *sys[i];
            const Layout& L = S.L;
            ecne_result* r = new ecne_result();
            Counters c;
            std::memcpy(&c, h_res.data() + i * ECNE_RESULT_BYTES, ECNE_RESULT_BYTES);      // (only the result part is valid)
            S.generation++;
            r->sys = &S;
            r->generation = S.generation;
            r->sys_uid = S.uid;
            ecne_summary& s = r->sum;
            std::memset(&s, 0, sizeof s);
            s.status = (c.err_key != ~0ull && (c.err_key & 0xFFu) != 0) ? -(int)(c.err_key & 0xFFu) : c.error;   // (the pop the sequential run dies on)
            // ids above num_variables: variable_states[i] raises BoundsError in the setup loop over known_variables (:682-692),
            // for a target at the verdict (:1579-1583) -- the first only if nothing else stopped the run earlier... it IS the first
            bool known_oob = false, target_oob = false;
            for (int64_t v : S.knowns) known_oob |= v > S.n_vars;
            for (int64_t v : S.targets) target_oob |= v > S.n_vars;
            const bool secp = sys[i]->secp_solve_override >= 0 ? sys[i]->secp_solve_override != 0 : o.secp_solve != 0;
            if (secp && L.dsu_err) s.status = ECNE_EBOUNDS;      // secp_solve's dsu setup raises before anything else (:634-678)
            else if (known_oob) s.status = ECNE_EBOUNDS;
            else if (target_oob && s.status == 0) s.status = ECNE_EBOUNDS;
            s.unique_nontrivial = (int64_t)c.unique_nontrivial;
            s.n_nontrivial = (int64_t)c.n_nontrivial;
            s.unique_targets = (int64_t)c.unique_targets;
            s.n_targets = (int64_t)S.targets.size();
            s.function_good = (s.status == 0 && c.error == 0 && s.unique_targets == s.n_targets) ? 1 : 0;
            s.successful_steps = (int64_t)c.successful_steps;
            s.outer_iterations = (int64_t)c.outer_iterations;
            s.pops = (int64_t)c.pops;
            s.num_unique = (int64_t)c.num_unique;
            for (int k = 0; k < 16; ++k) s.rule_hits[k] = (int64_t)c.rule_hits[k];
            s.n_rows = (int64_t)L.nC;
            s.n_vars = S.n_vars;
            s.pop_nnz = (int64_t)c.pop_nnz;
            s.device_ms = ms;
            s.classify_ms = S.dev.classify_ms;
            for (int k = 0; k < 8; ++k) s.queue_ms[k] = (double)c.qticks[k] * 1e-5;
            for (int k = 0; k < 8; ++k) s.multi_ms[k] = (double)c.mticks[k] * 1e-5;
            for (int k = 0; k < 16; ++k) s.sched[k] = (int64_t)c.sched[k];
            for (int k = 0; k < 4; ++k) s.team[k] = (int64_t)c.team_stat[k];
            for (int k = 0; k < 8; ++k) s.phase_ms[k] = (k == 6) ? (double)c.phase_ticks[k] : (double)c.phase_ticks[k] * 1e-5;
            out[i] = r;
        }
    } while (0);
    if (rc != ECNE_OK)
        for (size_t i = 0; i < n; ++i) { delete out[i]; out[i] = nullptr; }
    return rc;
}

// ---- the plan of a split file (SplitPlan). Groups: union-find over variables and rows -- a row with every variable it names (zero
// coefficients included) except the constant wire, and with the next row where the two could be one of P5's pairs (:1492-1536: a row
// with two non-zero C terms followed by one with no C and a single B term). Bins: the groups, largest first, each into the bin with
// the fewest rows so far, at most `cap` bins.
static std::atomic<int>& split_setting() {
    static std::atomic<int> m{[]() { const char* e = getenv("ECNE_SPLIT"); return e ? atoi(e) : 1; }()};
    return m;
}
static int split_mode() { return split_setting().load(std::memory_order_relaxed); }
// The device screen of build_split: classification (SH_TOUCH1 comes from there), rows that can write the constant wire's state, the
// groups of rows (k_cc_*). P.screen = {groups, rows of the largest, SH_TOUCH1 rows, 1 = the long-variable list overflowed: no answer}.
// (round 6) On the CALLER's stream, under the device's launch lock (shared: the screen's kernels are ordinary launches, but they must not be
// queued next to another thread's team launch, which holds the lock exclusively), its scratch kept per thread (no hipMalloc / hipFree -- both
// synchronise the device -- per first solve; buffers above 32 MB are not kept), ONE result copy of five words, launch errors checked.
struct ScreenScratch {
    int device = -1; uint32_t* d = nullptr; size_t bytes = 0;
    uint32_t* get(int dev, size_t need) {
        if (d && (device != dev || bytes < need)) { (void)hipFree(d); d = nullptr; bytes = 0; }
        if (!d) { if (hipMalloc((void**)&d, need) != hipSuccess) { (void)hipGetLastError(); d = nullptr; return nullptr; } bytes = need; device = dev; }
        return d;
    }
    void done() { if (d && bytes > (32u << 20)) { (void)hipFree(d); d = nullptr; bytes = 0; } }
};
static int split_screen(ecne_system& P, int device, hipStream_t stream) {
    const auto t0 = std::chrono::steady_clock::now();
    HIP_TRY(hipSetDevice(device));
    static thread_local ScreenScratch scratch;
    const uint32_t nC = P.dev.job.nC;
    // parent[nC], cnt[nC], long_vars[LONGCAP], out[8] (out[4] = SH_TOUCH1 rows), then a Job slot for the classification
    const size_t words = 2ull * nC + ECNE_CC_LONGCAP + 8;
    uint32_t* const d = scratch.get(device, 4 * words + sizeof(Job) + 256);
    if (!d) return K_ECAPACITY;
    uint32_t* const parent = d, * const cnt = d + nC, * const longv = d + 2ull * nC, * const out = longv + ECNE_CC_LONGCAP;
    Job* const slot = (Job*)(((uintptr_t)(out + 8) + 255) & ~(uintptr_t)255);
    std::shared_lock<std::shared_mutex> lock(device_launch_mutex(device));
    int rc = classify_system(P, stream, slot);
    uint32_t h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    hipError_t e = hipSuccess;
    if (rc == K_OK) {
        const uint32_t nb = std::min<uint32_t>(4096u, (std::max(nC, P.dev.job.nV) + 255u) / 256u);
        e = hipMemsetAsync(out, 0, 32, stream);
        hipLaunchKernelGGL(k_cc_init, dim3(nb), dim3(256), 0, stream, parent, cnt, nC);
        hipLaunchKernelGGL(k_cc_hook, dim3(nb), dim3(256), 0, stream, P.dev.job, parent, longv, out);
        hipLaunchKernelGGL(k_cc_hook_long, dim3(1024), dim3(256), 0, stream, P.dev.job, parent, (const uint32_t*)longv, (const uint32_t*)out);
        hipLaunchKernelGGL(k_cc_count, dim3(nb), dim3(256), 0, stream, parent, cnt, nC, out);
        hipLaunchKernelGGL(k_count_shape, dim3(std::min<uint32_t>(1024u, (nC + 255u) / 256u)), dim3(256), 0, stream, P.dev.job, (uint32_t)SH_TOUCH1, out + 4);
        if (e == hipSuccess) e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(h, out, 32, hipMemcpyDeviceToHost, stream);
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
    }
    lock.unlock();
    scratch.done();
    if (rc != K_OK) return rc;
    if (e != hipSuccess) { (void)hipGetLastError(); return ECNE_ENODEVICE; }
    P.screen[0] = h[0]; P.screen[1] = h[1]; P.screen[2] = h[4]; P.screen[3] = h[3] ? 1u : 0u;
    P.screen_done = true;
    P.screen_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return K_OK;
}

static int build_split(ecne_system& P, int device, uint32_t cap, bool eager, bool first_solve = false, hipStream_t stream = nullptr) {
    static const bool dbg = getenv("ECNE_SPLIT_DEBUG") != nullptr;
#define SPLIT_NO(why) do { if (dbg) fprintf(stderr, "[ecne split] no plan: %s\n", why); return K_OK; } while (0)
    const auto t0 = std::chrono::steady_clock::now();
    P.split_tried = true;
    if (!P.specials.empty() || !P.orig_var.empty() || !P.L.oob_blob.empty() || (int64_t)P.L.nV > P.n_vars) SPLIT_NO("trusted functions / ids above num_variables");
    if (std::find(P.knowns.begin(), P.knowns.end(), (int64_t)1) == P.knowns.end()) SPLIT_NO("known_variables without the constant wire");
    // Screen before anything is downloaded or planned: the parts share ONE variable, the constant wire, and a part that writes its state
    // (every `x <== 1` of a circuit is a row x - 1 = 0 whose R4 / R5 shapes can move the wire's bounds: SH_TOUCH1, set by k_classify_rows)
    // makes the split solve void (Family.var1_bad) -- plan, part uploads and a whole solve for nothing. Such a file stays one system.
    auto lap = [&](const char* what) { if (dbg) fprintf(stderr, "[ecne split] %-28s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count()); };
    // (round 5) ... and the groups themselves are counted on the device first (split_screen: union-find over the resident fan-out lists,
    // ~0.1-0.5 ms): a file that is one group -- nearly every file -- is known to be one before its rows are downloaded (18 ms per million)
    // and united on the host (17 ms)
    if (P.dev.arena && P.dev.device == device && P.dev.job.nC) {
        if (!P.screen_done) { const int rc = split_screen(P, device, stream); if (rc != K_OK) return rc; }
        if (dbg) fprintf(stderr, "[ecne split] device screen: %u groups, largest %u of %u rows, %u rows can write the constant wire, %.2f ms\n", P.screen[0], P.screen[1], P.dev.job.nC, P.screen[2], P.screen_ms);
        if (P.screen[2]) SPLIT_NO("rows that can write the constant wire's bounds");
        if (!P.screen[3]) {
            if (P.screen[0] < 2) SPLIT_NO("one group (device screen)");
            if (!eager && (uint64_t)P.screen[1] * 5 > (uint64_t)P.dev.job.nC * 3) SPLIT_NO("one group holds most of the file (device screen)");
            if (first_solve && (P.screen[0] < 8 || (uint64_t)P.screen[1] * 8 > (uint64_t)P.dev.job.nC)) { P.split_tried = false; SPLIT_NO("not before the first solve: fewer than eight groups or one with more than an eighth of the rows"); }
        } else if (first_solve) { P.split_tried = false; SPLIT_NO("not before the first solve: the screen's list of long variables overflowed"); }
    } else if (first_solve) { P.split_tried = false; SPLIT_NO("not before the first solve: not resident"); }
    lap("screen");
    { const int rc = sys_host_rows(P); if (rc != K_OK) return rc; }
    lap("host rows");
    const Rows& R = P.rows();
    const size_t nC = R.n();
    const uint32_t nV = (uint32_t)P.n_vars;
    if (nC < 2 || cap < 2) SPLIT_NO("too small");
    for (int p = 0; p < 3; ++p)
        for (uint32_t v : R.var[p]) if (v > nV || v == 0) SPLIT_NO("malformed ids");      // (one system)
    // nodes: variables 0 .. nV, then rows
    std::vector<uint32_t> uf((size_t)nV + 1 + nC);
    for (size_t i = 0; i < uf.size(); ++i) uf[i] = (uint32_t)i;
    auto find = [&](uint32_t x) { while (uf[x] != x) { uf[x] = uf[uf[x]]; x = uf[x]; } return x; };
    auto unite = [&](uint32_t a, uint32_t b) { a = find(a); b = find(b); if (a != b) uf[a < b ? b : a] = a < b ? a : b; };
    std::vector<uint32_t> nzc(nC, 0), nzb(nC, 0);
    for (size_t i = 0; i < nC; ++i) {
        const uint32_t rn = nV + 1 + (uint32_t)i;
        for (int p = 0; p < 3; ++p)
            for (uint64_t k = R.ptr[p][i]; k < R.ptr[p][i + 1]; ++k) {
                const uint32_t v = R.var[p][k];
                if (v != 1) unite(rn, v);
                if (!fp::is_zero(R.coef[p][k])) { if (p == 2) nzc[i]++; else if (p == 1) nzb[i]++; }
            }
    }
    for (size_t i = 0; i + 1 < nC; ++i)
        if (nzc[i] == 2 && nzc[i + 1] == 0 && nzb[i + 1] == 1) unite(nV + 1 + (uint32_t)i, nV + 1 + (uint32_t)(i + 1));
    lap("union-find");
    // groups by rows
    std::vector<uint32_t> grp_of_root(uf.size(), 0xFFFFFFFFu), grp_rows;
    std::vector<uint32_t> row_grp(nC);
    for (size_t i = 0; i < nC; ++i) {
        const uint32_t r = find(nV + 1 + (uint32_t)i);
        if (grp_of_root[r] == 0xFFFFFFFFu) { grp_of_root[r] = (uint32_t)grp_rows.size(); grp_rows.push_back(0); }
        row_grp[i] = grp_of_root[r];
        grp_rows[row_grp[i]]++;
    }
    const size_t nG = grp_rows.size();
    if (nG < 2) SPLIT_NO("one group");
    uint32_t big = 0;
    for (uint32_t c : grp_rows) big = std::max(big, c);
    if (!eager && (uint64_t)big * 5 > (uint64_t)nC * 3) SPLIT_NO("one group holds most of the file");      // nothing to gain
    const uint32_t nB = (uint32_t)std::min<size_t>(nG, std::min<uint32_t>(cap, 240u));
    std::vector<uint32_t> order(nG), grp_bin(nG);
    for (size_t g = 0; g < nG; ++g) order[g] = (uint32_t)g;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return grp_rows[a] > grp_rows[b]; });
    {
        std::priority_queue<std::pair<uint64_t, uint32_t>, std::vector<std::pair<uint64_t, uint32_t>>, std::greater<std::pair<uint64_t, uint32_t>>> pq;
        for (uint32_t b = 0; b < nB; ++b) pq.push({0, b});
        for (uint32_t g : order) { auto t = pq.top(); pq.pop(); grp_bin[g] = t.second; t.first += grp_rows[g]; pq.push(t); }
    }
    // variables: the bin of their group; variables no row names go with bin 0; the constant wire is in every bin
    std::vector<uint32_t> var_bin((size_t)nV + 1, 0);
    for (uint32_t v = 2; v <= nV; ++v) { const uint32_t g = grp_of_root[find(v)]; var_bin[v] = g == 0xFFFFFFFFu ? 0u : grp_bin[g]; }
    std::unique_ptr<SplitPlan> plan(new SplitPlan());
    plan->device = device;
    plan->n_groups = (uint32_t)nG;
    std::vector<uint32_t> child_id((size_t)nV + 1, 0);
    std::vector<std::vector<uint32_t>> bin_vars(nB), bin_rows(nB);
    for (uint32_t b = 0; b < nB; ++b) bin_vars[b] = {0u, 1u};          // ids 0 (unused) and 1 (the constant wire) keep their numbers
    for (uint32_t v = 2; v <= nV; ++v) { child_id[v] = (uint32_t)bin_vars[var_bin[v]].size(); bin_vars[var_bin[v]].push_back(v); }
    child_id[1] = 1;
    for (size_t i = 0; i < nC; ++i) bin_rows[grp_bin[row_grp[i]]].push_back((uint32_t)i);
    HIP_TRY(hipSetDevice(device));
    lap("groups, bins, ids");
    // the parts: rows copied out under their new ids, laid out and uploaded -- independent of each other: side by side on the host's
    // worker threads when the caller has asked for any (ecne_set_host_threads / ECNE_HOST_THREADS; every part's place in the plan is fixed
    // beforehand, so the plan does not depend on the thread count)
    std::vector<uint32_t> live_bins;
    for (uint32_t b = 0; b < nB; ++b) if (!bin_rows[b].empty()) live_bins.push_back(b);
    plan->kids.assign(live_bins.size(), nullptr);
    plan->d_map.assign(live_bins.size(), nullptr);
    std::vector<int> part_rc(live_bins.size(), K_OK);
    std::vector<double> part_t(2 * live_bins.size(), 0.0);
    for_chunks(live_bins.size(), [&](size_t slot, unsigned) {
        const uint32_t b = live_bins[slot];
        const auto tk0 = std::chrono::steady_clock::now();
        ecne_system* k = new ecne_system();
        plan->kids[slot] = k;
        Rows& K = k->reduced;
        K.start();
        for (uint32_t i : bin_rows[b])
            for (int p = 0; p < 3; ++p) {
                for (uint64_t e = R.ptr[p][i]; e < R.ptr[p][i + 1]; ++e) { K.var[p].push_back(child_id[R.var[p][e]]); K.coef[p].push_back(R.coef[p][e]); }
                K.ptr[p].push_back(K.var[p].size());
            }
        k->cur = &k->reduced;
        k->n_vars = (int64_t)bin_vars[b].size() - 1;
        k->n_rows_main = (int64_t)bin_rows[b].size();
        k->orig_var = bin_vars[b];
        k->orig_row = bin_rows[b];
        for (int64_t v : P.knowns) if (v == 1 || (v >= 2 && v <= (int64_t)nV && var_bin[(size_t)v] == b)) k->knowns.push_back(child_id[(size_t)v]);
        for (int64_t v : P.targets) if (v == 1 || (v >= 2 && v <= (int64_t)nV && var_bin[(size_t)v] == b)) k->targets.push_back(child_id[(size_t)v]);
        const auto tk1 = std::chrono::steady_clock::now();
        int rc = upload_system(*k, device);
        if (rc == K_OK) {
            uint32_t* dm = nullptr;
            if (hipMalloc((void**)&dm, 4ull * bin_vars[b].size()) != hipSuccess) rc = K_ENODEVICE;
            else {
                plan->d_map[slot] = dm;
                if (hipMemcpy(dm, bin_vars[b].data(), 4ull * bin_vars[b].size(), hipMemcpyHostToDevice) != hipSuccess) rc = K_ENODEVICE;
            }
        }
        part_rc[slot] = rc;
        part_t[2 * slot] = std::chrono::duration<double, std::milli>(tk1 - tk0).count();
        part_t[2 * slot + 1] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tk1).count();
    });
    double t_rows = 0, t_up = 0;
    for (size_t slot = 0; slot < live_bins.size(); ++slot) {
        t_rows += part_t[2 * slot]; t_up += part_t[2 * slot + 1];
        if (part_rc[slot] != K_OK) { if (dbg) fprintf(stderr, "[ecne split] part %u: upload failed (%d)\n", live_bins[slot], part_rc[slot]); return part_rc[slot]; }
    }
    lap("parts");
    if (dbg) fprintf(stderr, "[ecne split] parts: rows copied %.1f ms, layout + upload %.1f ms (summed over %u worker threads)\n", t_rows, t_up, for_chunks_workers(live_bins.size()));
    if (plan->kids.size() < 2) SPLIT_NO("fewer than two parts");
    HIP_TRY(hipMalloc((void**)&plan->d_family, sizeof(Family)));
    plan->ok = true;
    plan->plan_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (dbg) fprintf(stderr, "[ecne split] %zu groups -> %zu parts, plan %.1f ms\n", nG, plan->kids.size(), plan->plan_ms);
    P.split = std::move(plan);
    return K_OK;
#undef SPLIT_NO
}

static int ecne_solve_batch_impl(ecne_system** sys, size_t n, const ecne_opts* opts, ecne_result** out) {
    if (!sys || !out || n == 0) return ECNE_EINVAL;
    const int mode = split_mode();
    ecne_opts o;
    std::memset(&o, 0, sizeof o);
    if (opts) o = *opts;
    if (n == 1 && sys[0] && mode != 0 && o.queue_mode == 0 && o.debug == 0 && !o.secp_solve && sys[0]->secp_solve_override <= 0 && ecne_device_count() > o.device) {
        ecne_system& P = *sys[0];
        // (round 5: also before the FIRST solve of a file of more than two workgroups' rows -- the device screen costs a few launches, and only
        //  a file of eight or more groups none of which holds an eighth of the rows is planned then: the shape a team is worst at)
        const bool never_solved = P.last_kernel_ms == 0.0;
        if (!P.split_tried && (mode >= 2 || (P.n_rows() > 2ull * ECNE_ROWS_PER_WG && (P.last_kernel_ms >= 3.0 || never_solved)))) {
            RestoreDevice restore;
            int n_cu = 0;
            if (hipSetDevice(o.device) == hipSuccess && hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, o.device) == hipSuccess && n_cu > 16) {
                // (the file itself is laid out and uploaded first: its arrays receive the parts' states)
                int rc = upload_system(P, o.device);
                if (rc == K_OK) rc = build_split(P, o.device, (uint32_t)(n_cu - 8), mode >= 2, mode < 2 && never_solved, (hipStream_t)o.stream);
                if (rc != K_OK) { if (getenv("ECNE_SPLIT_DEBUG")) fprintf(stderr, "[ecne split] plan failed (%d)\n", rc); P.split.reset(); (void)hipGetLastError(); }
            }
        }
        if (P.split && P.split->ok && P.split->device == o.device) {
            SplitPlan& sp = *P.split;
            const size_t nk = sp.kids.size();
            std::vector<ecne_system*> xs(sp.kids.begin(), sp.kids.end());
            xs.push_back(&P);
            std::vector<ecne_result*> xr(nk + 1, nullptr);
            uint32_t ff[2] = {0, 0};
            int rc = solve_batch_core(xs.data(), nk + 1, opts, xr.data(), &sp, ff);
            bool good = rc == ECNE_OK && !ff[0] && !ff[1];
            if (good) for (size_t k = 0; k < nk; ++k) good = good && xr[k]->sum.status == 0;
            if (good) {
                ecne_result* r = xr[nk];
                ecne_summary& s = r->sum;
                s.successful_steps = s.pops = s.num_unique = s.pop_nnz = 0;
                for (int k = 0; k < 16; ++k) { s.rule_hits[k] = 0; s.sched[k] = 0; }
                for (int k = 0; k < 8; ++k) { s.queue_ms[k] = s.multi_ms[k] = s.phase_ms[k] = 0; }
                s.outer_iterations = xr[0]->sum.outer_iterations;
                for (size_t k = 0; k < nk; ++k) {
                    const ecne_summary& c = xr[k]->sum;
                    s.successful_steps += c.successful_steps; s.pops += c.pops; s.num_unique += c.num_unique; s.pop_nnz += c.pop_nnz;
                    for (int j = 0; j < 16; ++j) { s.rule_hits[j] += c.rule_hits[j]; s.sched[j] += c.sched[j]; }
                    for (int j = 0; j < 8; ++j) { s.queue_ms[j] = std::max(s.queue_ms[j], c.queue_ms[j]); s.multi_ms[j] = std::max(s.multi_ms[j], c.multi_ms[j]); s.phase_ms[j] = std::max(s.phase_ms[j], c.phase_ms[j]); }
                }
                s.function_good = (s.status == 0 && s.unique_targets == s.n_targets) ? 1 : 0;
                for (size_t k = 0; k < nk; ++k) delete xr[k];
                out[0] = r;
                return ECNE_OK;
            }
            if (getenv("ECNE_SPLIT_DEBUG")) fprintf(stderr, "[ecne split] split solve dropped: rc %d, part left early %u, constant wire written %u\n", rc, ff[0], ff[1]);
            for (auto* x : xr) delete x;
            if (rc != ECNE_OK && rc != K_ESPLIT) (void)hipGetLastError();
            P.split.reset();        // (an error in a part, the constant wire written, parts that do not fit: the file as one system, from now on)
        }
    }
    int rc = solve_batch_core(sys, n, opts, out, nullptr, nullptr);
    if (rc == ECNE_OK && tl_side_used) {
        // a team that gave up at its barrier (ECNE_ETIMEOUT) while a second kernel shared the device: the co-residency the occupancy
        // queries promised did not hold (another process, a dispatch order nobody controls) -- once more, one launch after the other
        bool timed_out = false;
        for (size_t i = 0; i < n; ++i) timed_out |= out[i] && out[i]->sum.status == ECNE_ETIMEOUT;
        if (timed_out) {
            for (size_t i = 0; i < n; ++i) { delete out[i]; out[i] = nullptr; }
            tl_side_forbid = true;
            rc = solve_batch_core(sys, n, opts, out, nullptr, nullptr);
            tl_side_forbid = false;
        }
    }
    if (rc == ECNE_OK && n == 1 && out[0]) sys[0]->last_kernel_ms = out[0]->sum.device_ms;
    return rc;
}

int ecne_solve(ecne_system* sys, const ecne_opts* opts, ecne_result** out) {
    return ecne_solve_batch(&sys, 1, opts, out);
}

int ecne_result_summary(const ecne_result* r, ecne_summary* out) {
    if (!r || !out) return ECNE_EINVAL;
    *out = r->sum;
    return ECNE_OK;
}
int ecne_result_summaries(ecne_result* const* r, size_t n, ecne_summary* out) {
    if ((!r || !out) && n) return ECNE_EINVAL;
    for (size_t i = 0; i < n; ++i) { if (!r[i]) return ECNE_EINVAL; out[i] = r[i]->sum; }
    return ECNE_OK;
}
static int ecne_result_states_impl(const ecne_result* r, const uint8_t** flags, const uint64_t** lb, const uint64_t** ub,
                       const int32_t** abz, const uint8_t** nvalues, const uint64_t** values) {
    if (!r) return ECNE_EINVAL;
    int st = fetch_states(const_cast<ecne_result*>(r));
    if (st != K_OK) return st;
    if (flags) *flags = r->flags.data();
    if (lb) *lb = r->lb.data();
    if (ub) *ub = r->ub.data();
    if (abz) *abz = r->abz.data();
    if (nvalues) *nvalues = r->nvalues.data();
    if (values) *values = r->values.data();
    return ECNE_OK;
}
static int ecne_result_bad_rows_impl(const ecne_result* r, const int64_t** rows, size_t* n) {
    if (!r) return ECNE_EINVAL;
    int st = fetch_states(const_cast<ecne_result*>(r));
    if (st != K_OK) return st;
    if (rows) *rows = r->bad_rows.data();
    if (n) *n = r->bad_rows.size();
    return ECNE_OK;
}
static int ecne_result_digest_impl(const ecne_result* r, uint64_t* out) {
    if (!r || !out) return ECNE_EINVAL;
    if (!r->sys || !system_is_live(r->sys) || r->sys->uid != r->sys_uid || r->sys->generation != r->generation || !r->sys->dev.arena) return ECNE_EINVAL;
    const ecne_system& S = *r->sys;
    RestoreDevice restore;
    HIP_TRY(hipSetDevice(S.dev.device));
    unsigned long long* d = nullptr;
    HIP_TRY(hipMalloc((void**)&d, 16));
    int rc = ECNE_OK;
    const uint32_t nv = (uint32_t)std::min<int64_t>(S.n_vars, (int64_t)S.L.nV);
    if (hipMemset(d, 0, 16) != hipSuccess) rc = ECNE_ENODEVICE;
    if (rc == ECNE_OK) {
        hipLaunchKernelGGL(k_state_digest, dim3(std::max(1u, std::min(2048u, (nv + 255u) / 256u))), dim3(256), 0, 0, S.dev.job, nv, d);
        if (hipMemcpy(out, d, 16, hipMemcpyDeviceToHost) != hipSuccess || hipGetLastError() != hipSuccess) rc = ECNE_ENODEVICE;
    }
    (void)hipFree(d);
    return rc;
}
int ecne_result_digest(const ecne_result* r, uint64_t out[2]) { return guarded([&] { return ecne_result_digest_impl(r, out); }); }
void ecne_result_free(ecne_result* r) { delete r; }
void ecne_results_free(ecne_result* const* r, size_t n) { if (r) for (size_t i = 0; i < n; ++i) delete r[i]; }

static int ecne_classify_impl(ecne_system* sys, const ecne_opts* opts, uint32_t* shape_out, double* kernel_ms, uint64_t* bytes) {
    if (!sys) return ECNE_EINVAL;
    ecne_opts o;
    std::memset(&o, 0, sizeof o);
    if (opts) o = *opts;
    if (ecne_device_count() <= o.device) return ECNE_ENODEVICE;
    RestoreDevice restore;
    HIP_TRY(hipSetDevice(o.device));
    int st = upload_system(*sys, o.device);
    if (st != K_OK) return st;
    Job* d_job = nullptr;
    HIP_TRY(hipMalloc((void**)&d_job, sizeof(Job)));
    sys->dev.classified = false;
    // restore the structural words as the front-end laid them down so that repeated calls time the same work
    if (sys->dev.lay) (void)hipMemcpy(sys->dev.job.rinfo, sys->dev.lay->dst.rinfo0, sizeof(RowInfo) * (size_t)sys->L.nC, hipMemcpyDeviceToDevice);
    else (void)hipMemcpy(sys->dev.job.rinfo, sys->L.rinfo.data(), sizeof(RowInfo) * sys->L.rinfo.size(), hipMemcpyHostToDevice);
    st = classify_system(*sys, (hipStream_t)o.stream, d_job);
    (void)hipFree(d_job);
    if (st != K_OK) return st;
    if (shape_out && sys->L.nC) {
        std::vector<RowInfo> ri(sys->L.nC);
        HIP_TRY(hipMemcpy(ri.data(), sys->dev.job.rinfo, sizeof(RowInfo) * ri.size(), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < ri.size(); ++i) shape_out[i] = ri[i].shape;
    }
    if (kernel_ms) *kernel_ms = sys->dev.classify_ms;
    if (bytes) *bytes = sys->L.stream_bytes;
    return ECNE_OK;
}

static int ecne_fp_selftest_impl(int device, int op, size_t n, const uint64_t* a, const uint64_t* b, uint64_t* out) {
    if (ecne_device_count() <= device) return ECNE_ENODEVICE;
    RestoreDevice restore;
    HIP_TRY(hipSetDevice(device));
    uint64_t *da = nullptr, *db = nullptr, *dout = nullptr;
    HIP_TRY(hipMalloc((void**)&da, 32 * n + 32));
    HIP_TRY(hipMalloc((void**)&db, 32 * n + 32));
    HIP_TRY(hipMalloc((void**)&dout, 32 * n + 32));
    HIP_TRY(hipMemcpy(da, a, 32 * n, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(db, b, 32 * n, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_fp_selftest, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, op, n, da, db, dout);
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, dout, 32 * n, hipMemcpyDeviceToHost));
    (void)hipFree(da); (void)hipFree(db); (void)hipFree(dout);
    return ECNE_OK;
}

int ecne_fp_sqrt(const uint64_t* a, uint64_t* root) {
    fp::u256 x = fp::make(a[0], a[1], a[2], a[3]), r;
    if (!fp::sqrt(fp::reduce(x), r)) return 0;
    for (int i = 0; i < 4; ++i) root[i] = r.w[i];
    return 1;
}


static void system_changed_rows(ecne_system* sys) {
    sys->generation++;
    sys->laid_out = false;
    sys->split.reset();
    sys->split_tried = false;
    sys->screen_done = false;
    if (sys->dev.arena) {
        RestoreDevice restore;
        (void)hipSetDevice(sys->dev.device);
        (void)hipFree(sys->dev.arena);
        sys->dev = DeviceImage();
    }
}
// ---- entry points proper: nothing throws across the ABI (guarded)
int ecne_r1cs_load(const char* path, ecne_r1cs** out) { return guarded([&] { return ecne_r1cs_load_impl(path, out); }); }
int ecne_r1cs_csr(const ecne_r1cs* f, int part, const uint64_t** rowptr, const uint32_t** col, const uint64_t** coeff) { return guarded([&] { return ecne_r1cs_csr_impl(f, part, rowptr, col, coeff); }); }
int ecne_system_from_r1cs(const ecne_r1cs* m, ecne_system** out) { return guarded([&] { return ecne_system_from_r1cs_impl(m, out); }); }
int ecne_abstract(ecne_system* sys, const ecne_r1cs* trusted, const char* name) { return guarded([&] { return ecne_abstract_impl(sys, trusted, name); }); }
int ecne_system_info_get(const ecne_system* sys, ecne_system_info* o) { return guarded([&] { return ecne_system_info_get_impl(sys, o); }); }
int ecne_system_rows(ecne_system* sys, int part, const uint32_t** rowptr, const uint32_t** col, const uint64_t** coeff) { return guarded([&] { return ecne_system_rows_impl(sys, part, rowptr, col, coeff); }); }
int ecne_solve_batch(ecne_system** sys, size_t n, const ecne_opts* opts, ecne_result** out) { return guarded([&] { return ecne_solve_batch_impl(sys, n, opts, out); }); }
int ecne_result_bad_rows(const ecne_result* r, const int64_t** rows, size_t* n) { return guarded([&] { return ecne_result_bad_rows_impl(r, rows, n); }); }
int ecne_classify(ecne_system* sys, const ecne_opts* opts, uint32_t* shape_out, double* kernel_ms, uint64_t* bytes) { return guarded([&] { return ecne_classify_impl(sys, opts, shape_out, kernel_ms, bytes); }); }
int ecne_fp_selftest(int device, int op, size_t n, const uint64_t* a, const uint64_t* b, uint64_t* out) { return guarded([&] { return ecne_fp_selftest_impl(device, op, n, a, b, out); }); }
int ecne_result_states(const ecne_result* r, const uint8_t** flags, const uint64_t** lb, const uint64_t** ub,
                       const int32_t** abz, const uint8_t** nvalues, const uint64_t** values) {
    return guarded([&] { return ecne_result_states_impl(r, flags, lb, ub, abz, nvalues, values); });
}

int ecne_set_host_threads(int n) { return (int)set_host_threads(n); }

// rows in DICTIONARY order (what readR1CS / abstraction hand to the solver: explicit zeros and the {1 => 0} placeholder of an empty
// part included), host copy; borrowed until the system changes
int ecne_system_dict_rows(ecne_system* sys, int part, const uint64_t** rowptr, const uint32_t** var, const uint64_t** coeff, uint64_t* n_rows) {
    if (!sys || part < 0 || part > 2) return ECNE_EINVAL;
    return guarded([&] {
        { const int rc = sys_host_rows(*sys); if (rc != K_OK) return rc; }
        const Rows& R = sys->rows();
        if (rowptr) *rowptr = R.ptr[part].data();
        if (var) *var = R.var[part].data();
        if (coeff) *coeff = reinterpret_cast<const uint64_t*>(R.coef[part].data());
        if (n_rows) *n_rows = R.n();
        return (int)ECNE_OK;
    });
}
// test hook: a host copy of one static array of the system's device image (uploaded / laid out on `device` first, not classified):
// 0-2 rp A/B/C, 3-5 col, 6-8 coef, 9 rinfo, 10 fo_ptr, 11 fo_rows, 12 nontrivial, 13 p4_list, 14 p4_b, 15 p4_s, 16 cls_list, 17 tbig,
// 18 bigrows, 19 long_list, 20 p5_rows, 21 p5_y, 22 rec, 23 foi, 24 {nC, nV, nP4, nP5, nBigCls, nLong, nBigRows, n_vals, device-laid}
int ecne_debug_static_array(ecne_system* sys, int device, int which, const void** data, size_t* bytes) {
    if (!sys || !data || !bytes) return ECNE_EINVAL;
    return guarded([&] {
        if (ecne_device_count() <= device) return (int)ECNE_ENODEVICE;
        const int prev = current_device();
        int rc = upload_system(*sys, device);
        if (rc != K_OK) { (void)hipSetDevice(prev); return rc; }
        const Job& J = sys->dev.job;
        const Layout& L = sys->L;
        const void* src = nullptr;
        size_t n = 0;
        const size_t nC = L.nC, nvar2 = (size_t)L.nV + 2;
        switch (which) {
            case 0: src = J.rpA; n = 4 * (nC + 1); break;
            case 1: src = J.rpB; n = 4 * (nC + 1); break;
            case 2: src = J.rpC; n = 4 * (nC + 1); break;
            case 3: src = J.colA; n = 4 * L.nnz[0]; break;
            case 4: src = J.colB; n = 4 * L.nnz[1]; break;
            case 5: src = J.colC; n = 4 * L.nnz[2]; break;
            case 6: src = J.coefA; n = 32 * L.nnz[0]; break;
            case 7: src = J.coefB; n = 32 * L.nnz[1]; break;
            case 8: src = J.coefC; n = 32 * L.nnz[2]; break;
            case 9: src = sys->dev.lay ? sys->dev.lay->dst.rinfo0 : nullptr; n = sizeof(RowInfo) * nC; break;
            case 10: src = J.fo_ptr; n = 4 * nvar2; break;
            case 11: src = J.fo_rows; n = 4 * (size_t)L.fo_total; break;
            case 12: src = J.nontrivial; n = (size_t)L.nV + 1; break;
            case 13: src = J.p4_list; n = 4 * (size_t)L.n_p4; break;
            case 14: src = J.p4_b; n = 4 * (size_t)L.n_p4; break;
            case 15: src = J.p4_s; n = 4 * (size_t)L.n_p4; break;
            case 16: src = J.cls_list; n = 4 * (size_t)L.n_cls; break;
            case 17: src = J.tbig; n = 2 * nC; break;
            case 18: src = J.bigrows; n = 4 * (size_t)J.nBigRows; break;
            case 19: src = J.long_list; n = 4 * (size_t)J.nLong; break;
            case 20: src = J.p5_rows; n = 4 * (size_t)L.n_p5; break;
            case 21: src = J.p5_y; n = 4 * (size_t)L.n_p5; break;
            case 22: src = J.rec; n = 64 * nC; break;
            case 23: src = J.foi; n = 16 * nvar2; break;
            case 24: break;
            default: (void)hipSetDevice(prev); return (int)ECNE_EINVAL;
        }
        std::vector<int64_t>& buf = sys->order_buf;
        if (which == 24) {
            const int64_t v[9] = {(int64_t)J.nC, (int64_t)J.nV, (int64_t)J.nP4, (int64_t)J.nP5, (int64_t)J.nBigCls, (int64_t)J.nLong, (int64_t)J.nBigRows,
                                  (int64_t)L.n_vals, sys->dev.lay ? 1 : 0};
            buf.assign(v, v + 9);
            n = sizeof v;
        } else if (which == 9 && !sys->dev.lay) {      // host layout: the structural words are the host's copy
            buf.assign((n + 7) / 8, 0);
            std::memcpy(buf.data(), L.rinfo.data(), n);
        } else {
            buf.assign((n + 7) / 8, 0);
            if (n && hipMemcpy(buf.data(), src, n, hipMemcpyDeviceToHost) != hipSuccess) { (void)hipSetDevice(prev); return (int)ECNE_ENODEVICE; }
        }
        (void)hipSetDevice(prev);
        *data = buf.data();
        *bytes = n;
        return (int)ECNE_OK;
    });
}
int ecne_system_split_info(const ecne_system* sys, double out4[4]) {
    if (!sys || !out4) return ECNE_EINVAL;
    const SplitPlan* sp = sys->split.get();
    out4[0] = sp && sp->ok ? (double)sp->kids.size() : 0.0;
    out4[1] = sp ? (double)sp->n_groups : 0.0;
    out4[2] = sp ? sp->plan_ms : 0.0;
    out4[3] = sys->split_tried ? 1.0 : 0.0;
    return ECNE_OK;
}
int ecne_set_split(int mode) {
    if (mode < 0 || mode > 2) return ECNE_EINVAL;
    split_setting().store(mode, std::memory_order_relaxed);
    return ECNE_OK;
}
static int ecne_warmup_impl(int device, double* ms_out) {
    const auto t0 = std::chrono::steady_clock::now();
    if (device < 0 || device >= ecne_device_count()) return ECNE_ENODEVICE;
    RestoreDevice restore;
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipFree(nullptr));
    {   // the runtime's copy path: pageable memory to the device and back (what the front-end's upload of a file does)
        const size_t bytes = 8u << 20;
        std::vector<unsigned char> h(bytes, 1);
        void* d = nullptr;
        HIP_TRY(hipMalloc(&d, bytes));
        const hipError_t e1 = hipMemcpy(d, h.data(), bytes, hipMemcpyHostToDevice), e2 = hipMemcpy(h.data(), d, 4096, hipMemcpyDeviceToHost);
        (void)hipFree(d);
        if (e1 != hipSuccess || e2 != hipSuccess) { (void)hipGetLastError(); return ECNE_ENODEVICE; }
    }
    { const int rc = fe::warmup(device); if (rc != K_OK) return rc; }
    // both solve kernels once: code object, kernel arguments, scratch memory. Three rows over six variables: c = a * b (R1), d = c (R5), e = 5 (R3)
    ecne_system s;
    Rows& K = s.reduced;
    K.start();
    auto term = [&](int p, uint32_t v, const fp::u256& c) { K.var[p].push_back(v); K.coef[p].push_back(c); };
    auto end_row = [&]() { for (int p = 0; p < 3; ++p) K.ptr[p].push_back(K.var[p].size()); };
    term(0, 3, fp::make(1)); term(1, 4, fp::make(1)); term(2, 2, fp::make(1)); end_row();
    term(2, 5, fp::make(1)); term(2, 2, fp::pminus1()); end_row();
    term(2, 6, fp::make(1)); term(2, 1, fp::neg(fp::make(5))); end_row();
    s.cur = &s.reduced;
    s.n_vars = 6; s.n_rows_main = 3;
    s.knowns = {1, 3, 4};
    s.targets = {2};
    s.split_tried = true;
    for (int nwg : {0, 2}) {
        ecne_opts o;
        std::memset(&o, 0, sizeof o);
        o.device = device; o.debug = nwg;
        ecne_system* ps = &s;
        ecne_result* r = nullptr;
        const int rc = solve_batch_core(&ps, 1, &o, &r, nullptr, nullptr);
        const bool ok = rc == ECNE_OK && r && r->sum.status == 0 && r->sum.function_good == 1;
        delete r;
        if (!ok) return rc != ECNE_OK ? rc : ECNE_ENODEVICE;
    }
    if (ms_out) *ms_out = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return ECNE_OK;
}
int ecne_warmup(int device, double* ms_out) { return guarded([&] { return ecne_warmup_impl(device, ms_out); }); }

int ecne_set_frontend(int mode) {
    if (mode >= 0 && mode <= 2) frontend_setting().store(mode, std::memory_order_relaxed);
    return frontend_setting().load(std::memory_order_relaxed);
}
int ecne_frontend_stats(double* out16) {
    if (!out16) return ECNE_EINVAL;
    const FrontendStats& f = g_fe_stats;
    const double v[16] = {(double)f.parse_dev, f.parse.upload_ms, f.parse.offsets_ms, f.parse.fill_ms, f.parse.total_ms, (double)f.parse.file_bytes,
                          (double)f.abs_dev, f.abs.prep_ms, f.abs.fp_ms, f.abs.scan_ms, f.abs.verify_ms, f.abs.compact_ms, (double)f.abs.n_cand,
                          (double)f.abs.n_matched + 1e6 * (double)f.abs.n_host_verified, (double)f.layout_dev, f.layout_ms};
    for (int i = 0; i < 16; ++i) out16[i] = v[i];
    return ECNE_OK;
}
int ecne_abstract_stats(double* out6) {
    if (!out6) return ECNE_EINVAL;
    const AbstractStats& st = g_last_abstract;
    out6[0] = st.used_device; out6[1] = st.fp_ms; out6[2] = st.scan_ms; out6[3] = st.bytes; out6[4] = st.upload_ms; out6[5] = (double)st.n_cand;
    return ECNE_OK;
}

// getVariables(constraints[i]) (:36-56) in Set order: keys of a, b, c (dictionary order, non-zero values) pushed into one Set
static void row_variables_in_set_order(const Rows& R, size_t i, jl::SlotTable& set, std::vector<int64_t>& out) {
    set.reset();
    bool ins;
    for (int p = 0; p < 3; ++p)
        for (uint64_t k = R.ptr[p][i]; k < R.ptr[p][i + 1]; ++k)
            if (!fp::is_zero(R.coef[p][k])) set.upsert((int64_t)R.var[p][k], 0, ins);
    set.for_each([&](int64_t key, int64_t) { out.push_back(key); });
}
int ecne_system_report_order(ecne_system* sys, int64_t row, const int64_t** vars, size_t* n) {
    if (!sys || row < 0) return ECNE_EINVAL;
    return guarded([&] {
        { const int rc = sys_host_rows(*sys); if (rc != K_OK) return rc; }
        if ((uint64_t)row > sys->rows().n()) return (int)ECNE_EINVAL;
        const Rows& R = sys->rows();
        std::vector<int64_t>& out = sys->order_buf;
        out.clear();
        jl::SlotTable set;
        if (row >= 1) row_variables_in_set_order(R, (size_t)row - 1, set, out);
        else {
            // l = [variables of every row ...; inputs and outputs of every special ...; targets ...]; Set(l) (:600-618):
            // union!(Set(), l) reserves room for length(l) keys before it pushes them
            std::vector<int64_t> l, one;
            for (size_t i = 0; i < R.n(); ++i) { one.clear(); row_variables_in_set_order(R, i, set, one); l.insert(l.end(), one.begin(), one.end()); }
            for (auto& sp : sys->specials) { l.insert(l.end(), sp.inputs.begin(), sp.inputs.end()); l.insert(l.end(), sp.outputs.begin(), sp.outputs.end()); }
            l.insert(l.end(), sys->targets.begin(), sys->targets.end());
            jl::SlotTable all;
            all.reserve((int64_t)l.size());
            bool ins;
            for (int64_t v : l) all.upsert(v, 0, ins);
            all.for_each([&](int64_t key, int64_t) { out.push_back(key); });
        }
        if (vars) *vars = out.data();
        if (n) *n = out.size();
        return (int)ECNE_OK;
    });
}

// A change of what the solve is asked (I/O lists, special constraints, rows) makes results of earlier solves
// unreadable (generation) and the flat layout / device image stale.
static void system_changed(ecne_system* sys) { system_changed_rows(sys); }
int ecne_system_set_io(ecne_system* sys, const int64_t* known, size_t nk, const int64_t* targets, size_t nt) {
    if (!sys || (nk && !known) || (nt && !targets)) return ECNE_EINVAL;
    return guarded([&] {
        for (size_t i = 0; i < nk; ++i) if (known[i] < 1) return (int)ECNE_EBOUNDS;
        for (size_t i = 0; i < nt; ++i) if (targets[i] < 1) return (int)ECNE_EBOUNDS;
        sys->knowns.assign(known, known + nk);
        sys->targets.assign(targets, targets + nt);
        system_changed(sys);
        return (int)ECNE_OK;
    });
}
int ecne_system_set_secp_solve(ecne_system* sys, int flag) {
    if (!sys) return ECNE_EINVAL;
    sys->secp_solve_override = flag < 0 ? -1 : (flag != 0);
    return ECNE_OK;
}
int ecne_system_clear_specials(ecne_system* sys) {
    if (!sys) return ECNE_EINVAL;
    sys->specials.clear();
    system_changed(sys);
    return ECNE_OK;
}
int ecne_system_add_special(ecne_system* sys, const char* name, const int64_t* inputs, size_t ni, const int64_t* outputs, size_t no) {
    if (!sys || !name || (ni && !inputs) || (no && !outputs)) return ECNE_EINVAL;
    return guarded([&] {
        for (size_t i = 0; i < ni; ++i) if (inputs[i] < 1 || inputs[i] > 0x7FFFFFFFll) return (int)ECNE_EBOUNDS;
        for (size_t i = 0; i < no; ++i) if (outputs[i] < 1 || outputs[i] > 0x7FFFFFFFll) return (int)ECNE_EBOUNDS;
        Special sp;
        sp.name = name;
        sp.inputs.assign(inputs, inputs + ni);
        sp.outputs.assign(outputs, outputs + no);
        sys->specials.push_back(std::move(sp));
        system_changed(sys);
        return (int)ECNE_OK;
    });
}
int ecne_system_io(const ecne_system* sys, const int64_t** known, size_t* nk, const int64_t** targets, size_t* nt) {
    if (!sys) return ECNE_EINVAL;
    if (known) *known = sys->knowns.data();
    if (nk) *nk = sys->knowns.size();
    if (targets) *targets = sys->targets.data();
    if (nt) *nt = sys->targets.size();
    return ECNE_OK;
}

int ecne_fp_solve_quadratic(const uint64_t* a_, const uint64_t* b_, const uint64_t* c_, int literal, uint64_t* roots, int* n_roots) {
    if (!a_ || !b_ || !c_ || !roots || !n_roots) return ECNE_EINVAL;
    auto ld = [](const uint64_t* p) { return fp::reduce(fp::make(p[0], p[1], p[2], p[3])); };
    auto st = [&](int i, const fp::u256& v) { for (int k = 0; k < 4; ++k) roots[4 * i + k] = v.w[k]; };
    const fp::u256 a = ld(a_), b = ld(b_), c = ld(c_);
    *n_roots = 0;
    if (fp::is_zero(a)) {
        if (fp::is_zero(b)) return fp::is_zero(c) ? 0 : 1;                       // "YES" / "NO" (:66-71)
        st(0, fp::mul(fp::neg(c), fp::inv(b)));                                  // divexact(-c, b) (:73)
        *n_roots = 1;
        return 2;
    }
    const fp::u256 two_a_inv = fp::inv(fp::add(a, a));
    const fp::u256 disc = fp::sub(fp::mul(b, b), fp::mul(fp::make(4), fp::mul(a, c)));   // b*b - 4*a*c (:77)
    fp::u256 rt;
    if (!fp::sqrt(disc, rt)) return 5;
    if (fp::is_zero(rt)) {                                                        // (:85)
        st(0, fp::mul(fp::neg(b), two_a_inv));
        *n_roots = 1;
        return 4;
    }
    const fp::u256 t = literal ? disc : rt;                                       // (:83-84) uses `disc` where `rt` was meant
    st(0, fp::mul(fp::add(fp::neg(b), t), two_a_inv));
    st(1, fp::mul(fp::sub(fp::neg(b), t), two_a_inv));
    *n_roots = 2;
    return 3;
}

const char* ecne_strerror(int st) {
    switch (st) {
        case ECNE_OK: return "ok";
        case ECNE_EFORMAT: return "malformed .r1cs (reference: AssertionError in readR1CS)";
        case ECNE_EBOUNDS: return "BoundsError (reference: variable_states[-1] or special-constraint indexing)";
        case ECNE_EDIVZERO: return "DivideError (reference: divexact by zero)";
        case ECNE_EUNDEF_DSU: return "UndefVarError: dsu (BigMultModP x BigLessThan needs secp_solve=true)";
        case ECNE_EKEY: return "KeyError in abstraction's variable map";
        case ECNE_EDETSIZE: return "linear-system group with more than 10 unknowns";
        case ECNE_EIO: return "cannot read file";
        case ECNE_ENODEVICE: return "no usable HIP device (the engine has no CPU fallback)";
        case ECNE_EINVAL: return "invalid argument";
        case ECNE_ECAPACITY: return "internal device table overflow (or out of memory)";
        case ECNE_ENOCONVERGE: return "the propagation queue does not drain on this input (the reference would not terminate): stopped after 4096 + 64 x nnz pops";
        case ECNE_ETIMEOUT: return "the workgroups of the solve cannot run together (cooperative launch refused, or they did not meet in time): is another process using this device?";
        default: return "unknown status";
    }
}
const char* ecne_version(void) { return "ecne-hip 0.1 (gfx950)"; }

}  // extern "C"
