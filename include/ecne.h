/* ecne.h — C ABI of libecne_hip, the MI355X-native replacement for the hot path of Ecne
 * (franklynwang/EcneProject): the fixed-point constraint-propagation loop SolveConstraintsSymbolic
 * and the two steps that feed it (readR1CS, abstraction).
 *
 * The reference has no FFI/plugin interface; its seam is three Julia functions. Each entry point
 * below names the reference interface it stands in for (paths relative to the reference tree):
 *
 *   ecne_r1cs_load / _info / _csr / _io      readR1CS(filename)          src/ParseR1CS.jl:50-124
 *   ecne_system_from_r1cs + ecne_abstract    the trusted-function loop   src/R1CSConstraintSolver.jl:513-544
 *                                            abstraction(...)            src/R1CSConstraintSolver.jl:237-395
 *   ecne_solve / ecne_solve_batch            SolveConstraintsSymbolic    src/R1CSConstraintSolver.jl:583-1646
 *   ecne_result_*                            its Bool result + the counts it prints (:1565-1592) and
 *                                            the per-variable VariableState (:135-160) it reports (:1609-1643)
 *
 * The Julia shim julia/EcneHIP.jl binds these with ccall and re-exports the reference's own
 * readR1CS / SolveConstraintsSymbolic / solveWithTrustedFunctions signatures.
 *
 * Conventions: plain pointers and sizes only; every function returns 0 or a negative ecne_status;
 * nothing throws across the ABI (allocation failures come back as ECNE_ECAPACITY); a handle is used by
 * one thread at a time, different handles may be used concurrently. Process-wide state is limited to
 * the host worker-thread count (ecne_set_host_threads) and per-thread launch scratch buffers. Variable ids are 1-based (wire id + 1, variable
 * 1 = the constant-one wire) exactly as in the reference. Field elements are 4 little-endian
 * uint64 limbs holding the canonical residue < p (BN254 scalar field).
 */
#ifndef ECNE_H
#define ECNE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum ecne_status {
    ECNE_OK = 0,
    ECNE_EFORMAT = -1,    /* @assert failures ParseR1CS.jl:58,62,69; truncated file                  */
    ECNE_EBOUNDS = -2,    /* BoundsError: variable_states[-1] (:916); special indexing (:762,:785)  */
    ECNE_EDIVZERO = -3,   /* DivideError from divexact by zero (:919-920, :1467)                    */
    ECNE_EUNDEF_DSU = -4, /* UndefVarError `dsu` (:762): BigMultModP x BigLessThan without secp_solve */
    ECNE_EKEY = -5,       /* KeyError in abstraction's variable map (:381-382)                      */
    ECNE_EDETSIZE = -6,   /* linear-system group with > 10 unknowns (reference would run k!*k steps) */
    ECNE_EIO = -7,
    ECNE_ENODEVICE = -8,  /* no usable HIP device: the engine never falls back to the CPU           */
    ECNE_EINVAL = -9,
    ECNE_ECAPACITY = -10, /* internal device table overflow (reported, never silently truncated)    */
    ECNE_ETIMEOUT = -11,  /* the workgroups of a solve cannot run together: the cooperative launch was refused, or they did not
                             meet at their barrier in time -- something else occupies the device (see ecne_solve)        */
    ECNE_ENOCONVERGE = -12 /* the queue of :805-1349 never drains on this input (rows that re-set each other's bounds / value and
                             re-queue each other forever, e.g. R3 lb=ub=v against R4's [0, 2^(l-1)-1]): the reference does not
                             terminate; the solve stops after 4096 + 64 x nnz pops. Not a resource failure (ECNE_ECAPACITY).    */
} ecne_status;

typedef struct ecne_r1cs ecne_r1cs;     /* a parsed .r1cs file                                   */
typedef struct ecne_system ecne_system; /* rows + special constraints + I/O lists fed to a solve  */
typedef struct ecne_result ecne_result; /* outcome of one solve                                  */

typedef struct ecne_info {
    uint32_t field_size, n_wires, n_pub_out, n_pub_in, n_prv_in, n_constraints;
    uint64_t n_labels;
    uint64_t nnz[3];   /* non-zero terms in A, B, C */
    int64_t n_vars;    /* nWires + 1 (ParseR1CS.jl:123) */
} ecne_info;

/* Host-side work (ecne_r1cs_load, ecne_abstract, the one-time flat-array layout of a system) runs on the
 * CALLING thread unless the caller opts in to worker threads: ecne_set_host_threads(n) (n <= 0: the cores
 * present, at most 32), or ECNE_HOST_THREADS=n in the environment. The library never spawns threads on its
 * own. Results do not depend on the count. Returns the count now in effect. */
int ecne_set_host_threads(int n);

/* Which front-end turns a file into the flat arrays the solve runs on: 0 = host (reader, abstraction and layout in host C++),
 * 1 = device whenever a HIP device is present (the constraint section is uploaded as it is and parsed, abstracted and laid out by
 * kernels on the calling thread's CURRENT HIP device; host copies of the rows are fetched only when an entry point needs them),
 * 2 = auto (device for files of 100 000 constraints and more) -- the default; ECNE_FRONTEND=host|device|auto in the environment
 * sets the initial value. Both produce the same arrays (tests/test_gpu_frontend.py); inputs the device path does not take (4 GiB
 * files, parts of 2^18 terms, a trusted function whose mapped inputs / outputs tie with other variables) go through the host
 * path on their own. mode < 0 only reads. Returns the mode in effect. */
int ecne_set_frontend(int mode);
/* One file, several independent parts. A file whose rows fall into groups that share no variable but the constant wire (N circuits
 * written into one file; the reference pops them from ONE queue, /root/reference/src/R1CSConstraintSolver.jl:805-1349, but a pop
 * only ever pushes rows of its own group) can be solved as a batch of single-workgroup jobs whose outer loops (:706-1556) run in
 * lockstep, and their states scattered back into the file's own arrays on the device: results, digests and bad rows are those of
 * the file as one system, bit for bit (tests/test_gpu_split.py). 0 = never, 1 = from the second ecne_solve of a system on, when
 * the first took long enough to pay for the plan -- and before the FIRST solve of a file of more than 6 144 rows whose groups, counted
 * on the device (union-find over the resident fan-out lists, ~1.5 ms per million rows), are eight or more with none holding an
 * eighth of the rows: many medium circuits in one file, what a team is worst at -- (the default; ECNE_SPLIT in the environment sets
 * the initial value), 2 = at the first solve. The plan's parts are built on the host's worker threads (ecne_set_host_threads). Only ecne_solve / a batch of one, queue_mode 0, no trusted functions; anything unusual in a part (an error, the
 * constant wire's state written) and the file is solved again as one system. Returns ECNE_OK, ECNE_EINVAL for another mode. */
int ecne_set_split(int mode);
/* out4 = {parts the system's next solve runs as (0: as one system), groups of rows found, host + upload time of the plan in ms,
 * 1 if a plan has been looked for}. */
int ecne_system_split_info(const ecne_system* sys, double out4[4]);
/* Timing of the calling thread's last trip through the front-end: out16 = {parse on device (0/1), upload ms, part-offset kernels ms,
 * row-fill kernels ms, parse total ms, file bytes, abstraction on device (0/1), pattern prep ms, fingerprint ms, window scan ms,
 * verification ms, compaction ms, candidate windows, matched windows (+ 1e6 x windows re-verified on the host), layout on device
 * (0/1), layout ms}. */
int ecne_frontend_stats(double* out16);

/* readR1CS — ParseR1CS.jl:50-124. */
int ecne_r1cs_load(const char* path, ecne_r1cs** out);
int ecne_r1cs_info(const ecne_r1cs* f, ecne_info* out);
/* CSR view of one part (0 = A, 1 = B, 2 = C) in file order, non-zero terms only; borrowed pointers,
 * valid until ecne_r1cs_free. coeff holds 4 limbs per term. col / coeff may be NULL when the part has no
 * non-zero term at all (rowptr[n] == 0); the same holds for ecne_system_rows. */
int ecne_r1cs_csr(const ecne_r1cs* f, int part, const uint64_t** rowptr, const uint32_t** col,
                  const uint64_t** coeff);
/* known_variables = [1] ++ inputs, target_variables = outputs (ParseR1CS.jl:123) */
int ecne_r1cs_io(const ecne_r1cs* f, const int64_t** known, size_t* n_known, const int64_t** targets,
                 size_t* n_targets);
void ecne_r1cs_free(ecne_r1cs* f);

/* solveWithTrustedFunctions :515-544: start from the main file, then abstract trusted functions away.
 * The caller passes trusted functions in the reference's order (already sorted long -> short, :527). */
int ecne_system_from_r1cs(const ecne_r1cs* main_file, ecne_system** out);
int ecne_abstract(ecne_system* sys, const ecne_r1cs* trusted, const char* name);
/* abstraction's O(rows) part -- row fingerprints and the window-candidate scan (:228-270) -- runs on HIP device 0 for main
 * files of 100 000 rows and more (ECNE_ABSTRACT_DEVICE=0 / 1 in the environment forces host / device); verification (:272-351)
 * and the greedy replacement (:368-388) are host code either way, the result does not depend on the choice. Statistics of the
 * calling thread's last ecne_abstract: out6 = {ran on the device (0/1), fingerprint kernel ms, scan + candidate kernels ms,
 * bytes the fingerprint kernel streamed, upload ms, candidate windows}. */
int ecne_abstract_stats(double* out6);
typedef struct ecne_system_info {
    int64_t n_rows;      /* rows handed to the solver (after abstraction) */
    int64_t n_rows_main; /* rows of the main file as read                 */
    int64_t n_vars, n_specials, n_known, n_targets;
    uint64_t nnz[3];
} ecne_system_info;
int ecne_system_info_get(const ecne_system* sys, ecne_system_info* out);
/* special constraint idx: name and mapped input / output variable lists (borrowed) */
int ecne_system_special(const ecne_system* sys, int64_t idx, const char** name, const int64_t** inputs,
                        size_t* n_inputs, const int64_t** outputs, size_t* n_outputs);
/* CSR of the rows handed to the solver, part 0/1/2, non-zero terms only, each row part in the order the
 * reference iterates nonzeroKeys(part) (a Julia Set): what printEquation (:431-456) walks when it
 * renders a constraint. Borrowed pointers, valid until the system is freed or abstracted again. */
int ecne_system_rows(ecne_system* sys, int part, const uint32_t** rowptr, const uint32_t** col, const uint64_t** coeff);
/* Iteration orders the text report walks (:1609-1643): row >= 1 -> the variables of constraint `row` in the order of
 * getVariables(constraints[row]) (a Julia Set filled from the a, b, c dictionaries, :36-56); row == 0 -> all_nontrivial_vars
 * (:600-618: every variable of every row, of the specials and the targets, as Set(l) iterates). Borrowed until the next
 * call on this system. */
int ecne_system_report_order(ecne_system* sys, int64_t row, const int64_t** vars, size_t* n);
/* The same rows in DICTIONARY order -- what readR1CS / abstraction hand to the solver (ParseR1CS.jl:94-115): every key of the part's
 * DefaultDict in iteration order, explicit zeros and the {1 => 0} placeholder of an empty part included. Host copy (a system that
 * came through the device front-end downloads it on first use); borrowed until the system changes. */
int ecne_system_dict_rows(ecne_system* sys, int part, const uint64_t** rowptr, const uint32_t** var, const uint64_t** coeff, uint64_t* n_rows);
/* Test hook: host copy of static array `which` of the system's device image after upload / layout on `device` (list in
 * ecne_engine.hip); borrowed until the next call on this system. tests/test_gpu_frontend.py compares host-laid and device-laid
 * systems with it, array by array. */
int ecne_debug_static_array(ecne_system* sys, int device, int which, const void** data, size_t* bytes);
/* SolveConstraintsSymbolic takes its known / target lists and special constraints as ARGUMENTS (:583-592); a system
 * made from a file starts with the file's lists (ParseR1CS.jl:123) and the specials abstraction produced. A caller
 * that edits them says so here (ids 1-based; copied). Changing them invalidates the device image of the system. */
int ecne_system_set_io(ecne_system* sys, const int64_t* known, size_t n_known, const int64_t* targets, size_t n_targets);
int ecne_system_clear_specials(ecne_system* sys);
/* kwarg secp_solve (:511) of THIS system when it is solved as part of a batch whose jobs do not all want the same (0 / 1; -1 = what
 * the launch's ecne_opts says, the default). */
int ecne_system_set_secp_solve(ecne_system* sys, int flag);
int ecne_system_add_special(ecne_system* sys, const char* name, const int64_t* inputs, size_t n_inputs,
                            const int64_t* outputs, size_t n_outputs);
int ecne_system_io(const ecne_system* sys, const int64_t** known, size_t* n_known, const int64_t** targets, size_t* n_targets);
/* Results made from this system stay readable (summary) but their per-variable state can no longer be fetched. */
void ecne_system_free(ecne_system* sys);

typedef struct ecne_opts {
    int32_t device;      /* HIP device ordinal                                                       */
    int32_t secp_solve;  /* kwarg secp_solve (:511): defines `dsu`, required when P2 has work (:762)  */
    int32_t debug;       /* test hook: > 0 forces that many cooperating workgroups per system        */
    int32_t queue_mode;  /* 0 = default schedule (rounds; wide frontiers of large systems as drain rounds on all
                            workgroups); 1 = strictly sequential pops on one wavefront: the reference's schedule
                            verbatim, for debugging and parity; 2 = the same on the chain executor; 3 = default
                            schedule with prefix rounds instead of drain rounds (round 2's schedule, for A/B runs);
                            4 = test hook: every frontier of two rows and more is drained. All modes give the same
                            results, bit for bit.                                                    */
    void* stream;        /* hipStream_t to launch on, or NULL for the device's default stream         */
} ecne_opts;

typedef struct ecne_summary {
    int32_t status;          /* ecne_status of the solve itself                                  */
    int32_t function_good;   /* the Bool SolveConstraintsSymbolic returns (:1594-1597, :1645)   */
    int64_t unique_nontrivial, n_nontrivial;   /* "Solved for U variables out of N" (:1565-1571) */
    int64_t unique_targets, n_targets;         /* "Solved for T target variables out of M" (:1586-1592) */
    int64_t successful_steps, outer_iterations, pops, num_unique;
    int64_t rule_hits[16];   /* 0..7 = R1..R8 (:827-1348), 8..12 = P1..P5 (:718-800, :1357-1550) */
    int64_t n_rows, n_vars;
    int64_t pop_nnz;         /* sum over queue pops of the popped row's non-zero count (roofline numerator) */
    double device_ms;        /* HIP-event time of the solve kernels on their stream              */
    double classify_ms;      /* HIP-event time of k_classify_rows                                */
    double queue_ms[8];      /* queue phase, master workgroup: head, mark, check+unmark, exec, flatten, resolve,
                                long rows popped alone + sequential bursts + wavefront rounds, multi-workgroup rounds */
    double multi_ms[8];      /* multi-workgroup rounds: mark, check + cut, exec + scan, expand, count + scan, write */
    double phase_ms[8];      /* in-kernel wall clock: 0 setup, 1 P1+P2+queue, 2 P3, 3 P4, 4 P5, 5 verdict; [6] = P3 rounds (count) */
    int64_t sched[16];       /* schedule diagnostics (master workgroup): 0 fast wavefront rounds, 1 rows they committed, 2 their
                                100 MHz ticks; 3..5 the same for general wavefront rounds; 6..12 why the fast round declined at
                                rank 0: no record / long row, other shape, error row, bound of the third kind, R7/R8 in reach
                                (x == y), R7/R8 in reach (sum); 12 = rounds of ONE pop whose events had more than three target
                                rows (committed by the fast round itself since round 2, not a decline); 13..15 multi-workgroup
                                rounds that committed < 64, < 4096, more rows */
    int64_t team[4];         /* (round 5, appended) a system on several workgroups: [0] outer iterations its master workgroup finished
                                alone -- P3 and P4 from the rows popped since the last pass, no job barrier --, [1] rows those passes
                                looked at, [2] full P3 / P4 sweeps with every workgroup, [3] 100 MHz ticks of the
                                iterations of [0], everything included */
} ecne_summary;

/* SolveConstraintsSymbolic :583-1646 on the GPU. Fails with ECNE_ENODEVICE when no HIP device is
 * usable. ecne_solve_batch runs n independent systems in one launch (one workgroup per system, more for large
 * ones). The workgroups of a large system meet at a barrier of their own and therefore have to be resident on
 * the device together: such a launch is refused at once (ECNE_ETIMEOUT) unless the device can hold its whole grid
 * (occupancy query; with ECNE_COOPERATIVE=1 in the environment it also goes through hipLaunchCooperativeKernel), holds at
 * most (CUs - 8) workgroups, and has the device to itself inside the process (single-workgroup solves share it with each
 * other). Other PROCESSES on the
 * device are not visible to the library -- run ONE solver process per device; as a last line of defence a barrier wait
 * WITHOUT PROGRESS is bounded (0.2 s + 2 us per row, ECNE_BARRIER_TIMEOUT_MS overrides; the master workgroup's heartbeat
 * restarts the clock) and ends in ECNE_ETIMEOUT instead of a hang. The caller's current HIP device is left as it was.
 * Out-of-range ids (malformed input): a known id above n_vars raises ECNE_EBOUNDS as the reference's setup does
 * (:682), a target id above n_vars at the verdict (:1580); a ROW or a special that mentions an id above n_vars raises
 * ECNE_EBOUNDS where the reference raises BoundsError: lazily, at the first rule that reads that state -- or never. */
int ecne_solve(ecne_system* sys, const ecne_opts* opts, ecne_result** out);
int ecne_solve_batch(ecne_system** sys, size_t n, const ecne_opts* opts, ecne_result** out);
int ecne_result_summary(const ecne_result* r, ecne_summary* out);
/* the summaries of a whole batch in one call (out[n]; a job runner that solves hundreds of small systems per pass -- the bulk runners of
 * /root/reference/src/Ecne.jl:9-37 -- spends more time crossing the FFI per result than the GPU spends on the batch) */
int ecne_result_summaries(ecne_result* const* r, size_t n, ecne_summary* out);
/* per-variable VariableState (:135-160), variable v at index v-1; borrowed, valid until ecne_result_free. The state
 * is downloaded on the first call: it must come before the system is solved again, abstracted, edited
 * (ecne_system_set_io / _add_special) or freed -- afterwards ECNE_EINVAL.
 * flags bit0 = unique, bit1 = is_known; lb/ub: 4 limbs; nvalues in {0,1,2}; values: 2 x 4 limbs. */
int ecne_result_states(const ecne_result* r, const uint8_t** flags, const uint64_t** lb, const uint64_t** ub,
                       const int32_t** abz, const uint8_t** nvalues, const uint64_t** values);
/* rows that still contain a non-uniquely-determined variable ("Bad Constraints", :1609-1618), 1-based */
int ecne_result_bad_rows(const ecne_result* r, const int64_t** rows, size_t* n);
/* 128-bit digest of the whole per-variable state of a finished solve (unique / is_known, lb, ub, abz, the candidate values of
 * variables 1..n_vars), computed ON THE DEVICE from the resident state -- two 64-bit sums over the variables of a splitmix64
 * chain over (v, flags & 3, abz, nvalues, lb, ub, values[0..nvalues)) -- so that repeated solves can be compared with each other
 * (run-to-run determinism, soak tests) without moving ~150 bytes per variable to the host. Same validity rule as
 * ecne_result_states. tests/test_gpu_soak.py restates the digest in numpy and checks it against fetched states. */
int ecne_result_digest(const ecne_result* r, uint64_t out[2]);
void ecne_result_free(ecne_result* r);
void ecne_results_free(ecne_result* const* r, size_t n);      /* ecne_result_free for every result of a batch */

/* k_classify_rows output for tests/profiling: per-row shape word (see ecne_engine.hip SH_*) */
int ecne_classify(ecne_system* sys, const ecne_opts* opts, uint32_t* shape_out /* n_rows */, double* kernel_ms,
                  uint64_t* bytes_streamed);

/* device self-tests of the field arithmetic: runs `n` vectors op(a,b) on the GPU.
 * op: 0 add, 1 sub, 2 mul, 3 inv(a), 4 neg(a), 5 a/b (field ops, operands < p); 6 integer quotient
 * a div b (R7, :1267-1268), 7 a*b > p as integers (R7, :1274; out = 0 or 1). a, b, out: n x 4 limbs. */
int ecne_fp_selftest(int device, int op, size_t n, const uint64_t* a, const uint64_t* b, uint64_t* out);
/* host-side utilities named by the north star (src/Math.jl:14-90 is dead code in the reference;
 * provided with self-consistency tests only): sqrt returns 1 and a root when one exists, else 0. */
int ecne_fp_sqrt(const uint64_t* a, uint64_t* root);
/* solveQuadratic (src/Math.jl:62-90; never called by the solver): a x^2 + b x + c = 0 over the field. Returns the kind of
 * answer: 0 = every x ("YES", :68), 1 = none ("NO", :70), 2 = the one root -c/b of a linear equation (:73), 3 = two values,
 * 4 = the double root -b/(2a) (:85), 5 = the discriminant has no square root (the reference's squareRoot never terminates on a
 * non-residue: no counterpart). n_roots values of 4 limbs each go to roots (room for 2). literal != 0 reproduces the
 * reference's two-value formula as written -- (-b +- disc)/(2a) with the DISCRIMINANT where the root was meant (:83-84) --
 * literal == 0 gives the true roots (-b +- sqrt(disc))/(2a). */
int ecne_fp_solve_quadratic(const uint64_t* a, const uint64_t* b, const uint64_t* c, int literal, uint64_t* roots, int* n_roots);

/* Optional, once per process and device, before the first file: pays the one-off costs of a cold process up front instead of inside
 * the first ecne_r1cs_load / ecne_solve -- the HIP runtime's copy path (its first host-to-device copy sets up staging buffers: 27-100 ms
 * on ROCm 7.2, whoever copies first), the library's two code objects (loaded by their first launch) and the device's scratch memory for
 * the solve kernels (grown by the runtime at their first launch: the first k_solve_team launch took 9.3 instead of 6.1 ms). It solves a
 * three-row system once on one workgroup and once on a team of two. Replaces nothing of the reference (readR1CS on a CPU has no such
 * phase, /root/reference/src/ParseR1CS.jl:50-124); without the call everything works, the first file just pays. ms_out (may be NULL):
 * wall-clock of the call. Works on `device` and leaves the calling thread's current device as it found it. */
int ecne_warmup(int device, double* ms_out);
int ecne_device_count(void);
const char* ecne_strerror(int status);
const char* ecne_version(void);

#ifdef __cplusplus
}
#endif
#endif /* ECNE_H */
