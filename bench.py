#!/usr/bin/env python3
"""bench.py — constraints resolved/sec of the HIP propagation engine on MI355X.

Metric (BASELINE.json): "constraints resolved/sec + wall-clock to fixed-point, ecdsa.r1cs".
A "step" is one full SolveConstraintsSymbolic-equivalent pass (setup + fixed point, kernel
k_solve) over the config-5 workload: ecdsa_like(26) — the deterministic synthetic stand-in for
the absent ecdsa.r1cs (tests/ecdsa_like.py) — with secp256k1.r1cs abstracted away as the trusted
function Secp256k1AddUnequal. The system (CSR, fan-out lists, row classification) is already
resident in HBM when the timed region starts; parsing, abstraction and classification are
reported separately. constraints/sec = rows of the main file as handed in / wall time per step.

Multi-GPU: a single circuit does not shard (its fixed point is one dependency chain), so
`--gpus N` runs one replica per rank ("replicas only") and finishes every step with
an RCCL all-reduce (MIN) of the 4-byte verdict/done word; value = total rows solved by all ranks
per second (weak scaling).

One JSON line on rank 0, with `roofline` (k_solve, HBM bound, algorithmic bytes B_alg of
SURVEY.md §8d taken from the sequential oracle's pop/iteration counters) and `cpu_baseline`
(the sequential CPU oracle timed on the same workload: ~17 s of CPU work on one core).
"""
import argparse
import gc
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "tests"))


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cold", action="store_true", help="skip the two fresh-subprocess cold-start measurements (config.cold_process)")
    ap.add_argument("--S", type=int, default=26, help="strides of ecdsa_like (26 = ECDSAPrivToPub(86,3))")
    ap.add_argument("--stride", type=int, default=10)
    ap.add_argument("--cpu-sample-S", type=int, default=26, help="strides of the CPU-baseline sample (26 = the whole workload: ~13 s solve + ~4 s parse / abstraction on one core)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--job-lines", choices=["auto", "on", "off"], default="auto",
                    help="default workload: attach the DAG / suite / many lines of the same run (config.job_workloads); auto = when --gpus > 1")
    ap.add_argument("--queue-mode", type=int, default=0)
    ap.add_argument("--copies", type=int, default=8, help="--workload many: copies of every mid-depth circomlib file (8 -> 504 jobs of 63 files)")
    ap.add_argument("--workload", choices=["ecdsa", "suite", "poseidon", "secp", "dag", "many"], default="ecdsa",
                    help="ecdsa = BASELINE.json config 5 (the metric's configuration, one circuit: replicas at N > 1); suite = config 4, the 67 "
                         "circomlib files sharded file-per-GPU; poseidon = config 2; secp = config 3; dag = config 5's trusted-subcircuit "
                         "verification DAG as four sharded jobs (ecneproject_amd/jobs.py); many = hundreds of mid-depth circuits (the suite without "
                         "its four long chains, --copies times): the job mix DESIGN.md section 6 predicts to scale over GPUs")
    ap.add_argument("--host-threads", type=int, default=0, help="host worker threads for parse / abstraction / layout (0 = the cores present, at most 32)")
    return ap.parse_args()


def csrc_sha16():
    """sha256 over the library's sources (ecneproject_amd/csrc/*, include/ecne.h), first 16 hex digits: profiles/traffic_latest.json
    carries the value of the build its PMC passes were taken on; a line printed by another build drops `roofline.traffic`"""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(HERE, "ecneproject_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".hpp")):
            h.update(name.encode())
            with open(os.path.join(d, name), "rb") as f:
                h.update(f.read())
    with open(os.path.join(HERE, "include", "ecne.h"), "rb") as f:
        h.update(f.read())
    # (round 5) what the sources are compiled WITH belongs to the build: the recipe (compiler flags live in ecneproject_amd/build.py) and
    # the developer's extra flags (ECNE_BUILD_FLAGS, e.g. -DECNE_ROUNDLOG)
    with open(os.path.join(HERE, "ecneproject_amd", "build.py"), "rb") as f:
        h.update(f.read())
    h.update(os.environ.get("ECNE_BUILD_FLAGS", "").encode())
    return h.hexdigest()[:16]


def schedule_env():
    """the ECNE_* environment switches of this process: most of them change the schedule the engine runs (ECNE_CREW, ECNE_LEVEL, ECNE_DRAIN,
    ECNE_SIDE_LAUNCH, ECNE_ROWS_PER_WG, ECNE_SPLIT, ECNE_LDS_BYTES, ...), so counter traffic measured under one set is not another set's"""
    skip = ("ECNE_FULL_ORACLE", "ECNE_FE_DEBUG", "ECNE_SPLIT_DEBUG", "ECNE_BUILD_FLAGS")
    return {k: v for k, v in sorted(os.environ.items()) if k.startswith("ECNE_") and k not in skip}


# DESIGN.md section 6's prediction of ms per step at N = 1, 2, 4, 8 GPUs (LPT packing of the measured single-job times of one MI355X; a batch takes as
# long as its longest job) lives in ONE place, profiles/predicted_scaling.json; `bench.py --gpus N` prints measured next to predicted so that the first
# run on an 8-GPU node grades the model by itself. "ecdsa": one circuit does not shard -- one instance per rank, the time per step stays.
def _predicted_table():
    try:
        with open(os.path.join(HERE, "profiles", "predicted_scaling.json")) as f:
            t = json.load(f)
        return {w: {int(n): ms for n, ms in row.items()} for w, row in t["ms_per_step"].items()}, t.get("source", "")
    except Exception:      # noqa: BLE001  (no table: the check prints measured values only)
        return {}, "profiles/predicted_scaling.json missing"


DESIGN_PREDICTED_MS, DESIGN_PREDICTED_SOURCE = _predicted_table()


def scaling_check(workload, world, ms_per_step, rank_ms, extra=None):
    """measured against DESIGN.md section 6's prediction for this workload and N (no efficiency is reported: the driver computes it from
    its own per-N runs; this is the model's self-check)"""
    pred = DESIGN_PREDICTED_MS.get(workload, {})
    out = {"n_gpus": world, "measured_ms_per_step": round(ms_per_step, 3), "predicted_ms_per_step": pred.get(world),
           "measured_over_predicted": round(ms_per_step / pred[world], 3) if pred.get(world) else None,
           "rank_ms_per_step": [round(x, 3) for x in rank_ms],
           "slowest_over_mean_rank": round(max(rank_ms) / max(sum(rank_ms) / len(rank_ms), 1e-9), 3) if rank_ms else None,
           "predicted_table_ms": {str(k): v for k, v in sorted(pred.items())},
           "source": "profiles/predicted_scaling.json = DESIGN.md section 6 (%s; no multi-GPU run behind it until a SCALE file exists)" % DESIGN_PREDICTED_SOURCE}
    if extra:
        out.update(extra)
    return out


COLD_CHILD = r"""
import json, os, sys, time
t0 = time.perf_counter()
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import ecneproject_amd as E                      # (no torch in this process)
import fixtures
t_import = time.perf_counter() - t0
out = {"import_ms": round(t_import * 1e3, 1)}
if %(warm)r:
    out["ecne_warmup_ms"] = round(E.warmup(%(device)d), 1)
t = time.perf_counter()
f = E.R1CS(%(path)r)
tr = E.R1CS(fixtures.path("secp256k1.r1cs"))
s = E.System(f)
s.abstract(tr, "Secp256k1AddUnequal")
r = E.solve_batch([s], device=%(device)d, fetch_states=False)[0]
out["first_file_to_verdict_ms"] = round((time.perf_counter() - t) * 1e3, 1)
out["first_solve_kernel_ms"] = round(float(r.summary.device_ms), 3)
out["verdict"] = bool(r.function_good)
fs = E.frontend_stats()
out["first_upload_ms"] = round(fs.get("upload_ms", 0.0), 1)
r2 = E.solve_batch([s], device=%(device)d, fetch_states=False)[0]
out["second_solve_kernel_ms"] = round(float(r2.summary.device_ms), 3)
print("COLD " + json.dumps(out))
"""


def cold_process(path, device, warm):
    """a FRESH process without torch: [ecne_warmup] -> readR1CS of the bench file -> abstraction -> first solve -> verdict (wall clock), and the
    second solve's kernel time next to the first's. What a one-shot solveWithTrustedFunctions caller sees; the page cache is warm (the
    file was just generated / read by this process)."""
    import subprocess
    code = COLD_CHILD % {"root": HERE, "path": path, "device": device, "warm": bool(warm)}
    try:
        p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
        line = [l for l in p.stdout.splitlines() if l.startswith("COLD ")]
        return json.loads(line[0][5:]) if line else {"error": (p.stderr or p.stdout)[-300:]}
    except Exception as e:      # noqa: BLE001
        return {"error": repr(e)}


def step_invariants(r):
    """what every timed step must reproduce exactly (a schedule may change the time of a solve, never its result)"""
    s = r.summary
    return (int(r.status), bool(r.function_good), int(s.pops), int(s.successful_steps), int(s.num_unique), int(s.outer_iterations),
            int(s.unique_nontrivial), int(s.n_nontrivial), int(s.unique_targets), int(s.n_targets), tuple(int(x) for x in list(s.rule_hits)[:13]))


def probe_julia():
    """the reference itself can only be timed where Julia 1.7 and an instantiated Ecne checkout exist."""
    import shutil
    import subprocess
    exe = shutil.which("julia")
    if not exe:
        return {"julia": None, "note": "julia not on PATH: the reference cannot be timed here; the baseline is the sequential C++ restatement"}
    try:
        v = subprocess.run([exe, "--version"], capture_output=True, text=True, timeout=30).stdout.strip()
    except Exception as e:      # noqa: BLE001
        v = "error: %r" % (e,)
    return {"julia": v, "note": "julia found; timing the reference needs an instantiated Ecne checkout (ECNE_REFERENCE_DIR), see julia/dump_unique.jl"}


PORT_DETAIL = ("statement-by-statement restatement of the reference (oracle/ecne_oracle.cpp) with Julia's Dict / Set slot order emulated; NOT a tuned CPU solver -- "
               "it runs 83 k rows/s on config 5 and 0.4-1.2 M rows/s on configs 2 / 3; a GPU/CPU ratio against it says nothing about kernel quality")

MANY_EXCLUDES = ("EdDSAMiMCSpongeVerifier", "EdDSAMiMCVerifier", "EdDSAPoseidonVerifier", "BabyPbk")      # the suite's four long dependency chains (14-24 k rows each)


def workload_jobs(name, args):
    """The job lists of the BASELINE.json configurations that are batches of independent solves (ecneproject_amd.jobs.Job)."""
    import ecdsa_like
    import fixtures
    from ecneproject_amd import jobs as J
    fx = fixtures.path
    if name == "suite":        # config 4
        rels = fixtures.circomlib_suite()
        return [J.Job(fx(r), r) for r in rels], "ecne_circomlib_tests/*.r1cs (BASELINE.json config 4), sharded file-per-GPU, one batch launch per rank", "reference fixtures (67 circom outputs)"
    if name == "poseidon":     # config 2
        r = "ecne_circomlib_tests/Poseidon@poseidon.r1cs"
        return [J.Job(fx(r), r)], "ecne_circomlib_tests/Poseidon@poseidon.r1cs (BASELINE.json config 2)", "reference fixture"
    if name == "many":         # not a BASELINE configuration: a batch that DOES scale over GPUs -- many circuits of similar depth
        rels = [r for r in fixtures.circomlib_suite() if not any(k in r for k in MANY_EXCLUDES)]
        copies = max(1, int(getattr(args, "copies", 8)))
        return ([J.Job(fx(r), "%s#%d" % (r, c)) for c in range(copies) for r in rels],
                "many mid-depth circuits: ecne_circomlib_tests/*.r1cs without the four long chains (%s), %d copies of each = %d independent jobs, "
                "LPT-packed job-per-GPU, batch launches of up to 248 single-workgroup jobs" % (", ".join(MANY_EXCLUDES), copies, copies * len(rels)),
                "reference fixtures, replicated")
    secp = J.Job(fx("secp256k1.r1cs"), "secp256k1", [(fx("bigmultmodp.r1cs"), "BigMultModP"), (fx("biglessthan.r1cs"), "BigLessThan")], True)
    if name == "secp":         # config 3
        return [secp], "secp256k1.r1cs + trusted bigmultmodp.r1cs, biglessthan.r1cs, secp_solve=true (BASELINE.json config 3)", "reference fixtures"
    if name == "dag":          # config 5's trusted-subcircuit verification DAG: four solves, verdicts AND-ed
        main = ecdsa_like.cached(args.S, args.stride, directory="/tmp/ecne_bench_%d" % os.getuid())
        return ([J.Job(main, "ecdsa_like(%d)" % args.S, [(fx("secp256k1.r1cs"), "Secp256k1AddUnequal")]), secp,
                 J.Job(fx("bigmultmodp.r1cs"), "bigmultmodp"), J.Job(fx("biglessthan.r1cs"), "biglessthan")],
                "config-5 verification DAG: ecdsa_like(S=%d) <- secp256k1; secp256k1 <- bigmultmodp, biglessthan; bigmultmodp; biglessthan "
                "(four independent solves sharded job-per-GPU, verdicts AND-ed by the all-reduce)" % args.S,
                "synthetic ecdsa_like + reference fixtures")
    raise SystemExit("unknown workload " + name)


def _oracle_job(t):
    """(worker of the CPU baseline) one job through the sequential oracle; returns (rows of the main file, t_solve, pops, alg bytes, ...)"""
    import orc
    path, trusted, names, secp = t
    o = orc.run(path, trusted, names, secp, want_states=False)
    s = o.summary
    return int(s.n_rows_main), float(s.t_solve), int(s.pops), int(o.alg_bytes()), float(s.t_read + s.t_abstract), int(o.status), bool(o.verdict)


def cpu_baseline_jobs(jobs, label):
    """the sequential restatement on ONE core, job after job -- and the same jobs file-parallel, one process per
    job over the host cores (core count stated)."""
    import multiprocessing as mp
    items = [(j.r1cs, [f for f, _ in j.trusted], [n for _, n in j.trusted], j.secp_solve) for j in jobs]
    t0 = time.perf_counter()
    seq = [_oracle_job(t) for t in items]
    wall_seq = time.perf_counter() - t0
    rows, t_solve = sum(r[0] for r in seq), sum(r[1] for r in seq)
    out = {"value": rows / max(t_solve, 1e-9), "unit": "constraints/s", "cores": 1, "kind": "port", "kind_detail": PORT_DETAIL,
           "sample": "%s: the whole workload, %d job(s), %d rows, sequential oracle solve %.3f s in total (parse + abstraction %.3f s excluded, "
                     "as for the GPU); longest single job %.3f s" % (label, len(jobs), rows, t_solve, sum(r[4] for r in seq), max(r[1] for r in seq)),
           "host_cores_available": os.cpu_count(), "reference_probe": probe_julia(), "wall_s_one_core_incl_parse": round(wall_seq, 3),
           "_per_job": {j.name: (r[5], r[6], r[2]) for j, r in zip(jobs, seq)}}
    if len(items) > 1:
        ncore = min(len(items), os.cpu_count() or 1)
        order = sorted(range(len(items)), key=lambda i: -seq[i][1])          # longest first
        with mp.get_context("fork").Pool(ncore) as pool:
            t0 = time.perf_counter()
            par = pool.map(_oracle_job, [items[i] for i in order], chunksize=1)
            wall = time.perf_counter() - t0
        out["file_parallel"] = {"value": rows / max(wall, 1e-9), "unit": "constraints/s", "cores": ncore, "wall_s": round(wall, 4),
                                "longest_job_solve_s": round(max(r[1] for r in par), 4),
                                "note": "one process per job over %d of the %d host cores; wall clock includes parse + abstraction of every file" % (ncore, os.cpu_count() or 1)}
    return out


def run_jobs_workload(args, torch, dist, rank, local_rank, world, workload=None, emit=True):
    """Configs 2, 3, 4 and the config-5 DAG: independent jobs, LPT-packed onto the ranks, one batch launch per rank and step, one
    all-reduce (MIN) of the verdict word. One step = every job of the workload solved once. emit=False: rank 0 gets the line back as a
    dict instead of printing it (the default workload at N > 1 attaches the DAG / suite / many lines of the same run)."""
    if workload is not None:
        args = argparse.Namespace(**dict(vars(args), workload=workload))
    import ecneproject_amd as E
    from ecneproject_amd import jobs as J
    from ecneproject_amd import sharding
    E.set_host_threads(args.host_threads)
    jl, label, data = workload_jobs(args.workload, args)
    runner = J.Runner(jl, rank, world, local_rank, dist if world > 1 else None)
    stream = torch.cuda.current_stream().cuda_stream
    for _ in range(max(args.warmup, 1)):
        res, ok = runner.run(stream=stream)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    # (the cyclic collector is held off while the steps are timed: with torch loaded a full collection walks millions of objects -- 40 ms, once
    #  per ~15 steps of a workload that keeps hundreds of result objects per step -- and is no part of the solve)
    gc.collect(); gc.disable()
    t0 = time.perf_counter()
    inv = []
    for _ in range(args.steps):
        res, ok = runner.run(stream=stream)
        inv.append(res)
    torch.cuda.synchronize()
    t_rank = time.perf_counter() - t0
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    gc.enable()
    # (after the timed region -- until the end of round 5 this sat in front of `elapsed` and the line's ms_per_step carried it: 0.9 ms per step
    #  of Python over 504 jobs' counters, `--workload many` 3.0 ms where the steps themselves took 2.1; per_rank.ms_per_step never did)
    inv = [[step_invariants(r) for r in step] for step in inv]
    if any(step != inv[0] for step in inv):
        raise SystemExit("bench.py: the timed steps did not reproduce the same result (verdicts / counts / counters differ between steps)")
    # per-job figures of this rank: rows, device ms, pops, algorithmic bytes (from the solve's own counters -- equal
    # to the oracle's by the parity tests)
    mine = []
    for i, r in zip(runner.mine, res):
        s, inf = r.summary, runner.systems[runner.mine.index(i)].info
        nnz = int(inf.nnz[0] + inf.nnz[1] + inf.nnz[2])
        b_alg = 20 * int(s.pops) + 40 * int(s.pop_nnz) + (3 * int(s.outer_iterations) + 1) * (12 * int(inf.n_rows) + 40 * nnz)
        mine.append({"job": jl[i].name, "rows_main": int(inf.n_rows_main), "device_ms": float(s.device_ms), "pops": int(s.pops), "alg_bytes": b_alg,
                     "verdict": bool(r.function_good), "status": int(r.status), "outer_iterations": int(s.outer_iterations)})
    per_rank = [{"rank": rank, "ms_per_step": t_rank * 1e3 / max(args.steps, 1), "jobs": mine}]
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        gathered = [None] * world
        dist.all_gather_object(gathered, per_rank[0])
        per_rank = gathered
    if rank != 0:
        return None
    alljobs = [j for pr in per_rank for j in pr["jobs"]]
    rows = sum(j["rows_main"] for j in alljobs)
    ms_per_step = elapsed * 1e3 / max(args.steps, 1)
    b_alg = sum(j["alg_bytes"] for j in alljobs)
    longest = max(alljobs, key=lambda j: j["pops"])       # (device_ms is the batch launch's time, the same for every job of a rank)
    # the dominant kernel is k_solve: one launch per rank and step; its duration = the batch's longest job on that rank
    k_ms = max(max((j["device_ms"] for j in pr["jobs"]), default=0.0) for pr in per_rank)
    loads = [sum(runner.weights[i] for i in part) for part in sharding.assign(runner.weights, world)]
    achieved = b_alg / (k_ms * 1e-3) / 1e9
    out = {
        "metric": "constraints resolved/sec (wall-clock to fixed point), %s" % args.workload,
        "value": rows * args.steps / elapsed, "unit": "constraints/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "u64x4 (BN254 Fp limbs) + u8/u32 flags", "data": data,
        "config": {"workload": label, "jobs": len(jl), "rows": rows, "verdicts_true": sum(int(j["verdict"]) for j in alljobs),
                   "all_ran": bool(runner.all_ran), "all_sound": bool(ok),
                   "scaling_check": scaling_check(args.workload, world, ms_per_step, [pr["ms_per_step"] for pr in per_rank]),
                   "per_rank": [{"rank": pr["rank"], "ms_per_step": round(pr["ms_per_step"], 3), "jobs": [j["job"] for j in pr["jobs"]],
                                 "longest_job_ms": round(max((j["device_ms"] for j in pr["jobs"]), default=0.0), 3)} for pr in per_rank],
                   "lpt": {"weights": "non-zeros of the main file", "rank_loads": loads, "imbalance": (max(loads) / max(sum(loads) / len(loads), 1e-9)) if loads else None,
                           # what the packing alone says about N = 1, 2, 4, 8 (no run needed): the heaviest rank's share of the weight, and the
                           # time model t(N) ~ t(1) x that share, never below the share of the heaviest single job
                           "predicted_scaling": {str(nn): {"max_rank_share": round(max(sum(runner.weights[i] for i in part) for part in sharding.assign(runner.weights, nn)) / max(sum(runner.weights), 1), 4),
                                                           "imbalance": round(max(sum(runner.weights[i] for i in part) for part in sharding.assign(runner.weights, nn)) * nn / max(sum(runner.weights), 1), 3)}
                                                 for nn in (1, 2, 4, 8)},
                           "heaviest_job_share": round(max(runner.weights) / max(sum(runner.weights), 1), 4)},
                   "predicted_bound": {"ms_per_step_at_any_n": round(k_ms if world == 1 else longest["device_ms"], 3), "job": longest["job"],
                                       "note": "a batch takes as long as its longest job: more GPUs cannot go below it (strong scaling saturates at the number of jobs that take about that long)"}},
        # (two kernels since round 3: k_solve for batches of single-workgroup jobs, k_solve_team as soon as one job of the batch has a team)
        "roofline": {"bound": "hbm", "kernel": "k_solve_team" if args.workload == "dag" else "k_solve", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0, "traffic": None,
                     "alg_bytes_per_step": b_alg, "kernel_ms": k_ms,
                     "latency_model": {"longest_job": longest["job"], "pops": longest["pops"], "outer_iterations": longest["outer_iterations"],
                                       "us_per_pop_on_the_longest_chain": 1e3 * longest["device_ms"] / max(longest["pops"], 1),
                                       "model": "cache-resident and dependency-depth bound: t ~ pops of the longest chain x us per sequential pop (+ rounds where the frontier is wide)"},
                     "note": "the working set of these circuits (<= 12 MB) lives in L2 / Infinity Cache; the HBM fraction is reported for completeness, the latency model is the bound"},
    }
    out["config"]["python_gc"] = "cyclic collector held off during the timed steps (a full collection with torch loaded: ~40 ms)"
    out["config"]["invariants"] = {"steps_identical": True, "checked": "status, verdict, pops, successful_steps, num_unique, outer_iterations, the four printed counts, rule hits -- every job, every timed step"}
    if not args.no_cpu_baseline and world == 1 and emit:
        out["cpu_baseline"] = cpu_baseline_jobs(jl, args.workload)
        # the oracle leg doubles as a check of the timed steps' counters (cpu_baseline_jobs keeps the oracle's per job)
        want = out["cpu_baseline"].pop("_per_job")
        got = {j["job"]: (j["status"], j["verdict"], j["pops"]) for j in alljobs}
        bad = [n for n, w in want.items() if got.get(n) != w]
        if bad:
            raise SystemExit("bench.py: the solve disagrees with the sequential oracle on %r" % bad[:5])
        out["config"]["invariants"]["matches_oracle"] = "status, verdict, pops of every job equal the sequential oracle's"
    if emit:
        print(json.dumps(out))
    return out


def attached_job_lines(args, torch, dist, rank, local_rank, world):
    """the workloads that DO shard -- the config-5 verification DAG, the circomlib suite (config 4) and `many` -- measured in
    the same run as the default line at N > 1 (every rank takes part; rank 0 gets {workload: short line})."""
    lines = {}
    for w in ("dag", "suite", "many"):
        o = run_jobs_workload(args, torch, dist, rank, local_rank, world, workload=w, emit=False)
        if rank == 0 and o is not None:
            c = o["config"]
            lines[w] = {"value": o["value"], "unit": o["unit"], "ms_per_step": round(o["ms_per_step"], 4), "scaling": o["scaling"], "jobs": c["jobs"], "rows": c["rows"],
                        "all_ran": c["all_ran"], "all_sound": c["all_sound"], "verdicts_true": c["verdicts_true"],
                        "per_rank": [{"rank": pr["rank"], "ms_per_step": pr["ms_per_step"], "n_jobs": len(pr["jobs"]), "longest_job_ms": pr["longest_job_ms"]} for pr in c["per_rank"]],
                        "lpt_rank_loads": c["lpt"]["rank_loads"], "lpt_imbalance": c["lpt"]["imbalance"], "scaling_check": c["scaling_check"]}
    return lines


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the engine has no CPU path)")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)

    if args.workload != "ecdsa":
        run_jobs_workload(args, torch, dist, rank, local_rank, world)
        if world > 1:
            dist.destroy_process_group()
        return

    import ecneproject_amd as E
    import ecdsa_like
    import fixtures
    E.set_host_threads(args.host_threads)      # the caller opts in to host worker threads (include/ecne.h)

    # ---- build the workload (host side, untimed): generate, parse, abstract, lay out, upload, classify
    t0 = time.time()
    # One circuit does not shard: at N > 1 every rank solves its OWN instance -- the same template with its own table constants
    # (tests/ecdsa_like.py seed = rank; rank 0 = the committed workload whose oracle digest is on file) -- N independent jobs, not one job N times.
    path = ecdsa_like.cached(args.S, args.stride, directory="/tmp/ecne_bench_%d" % os.getuid(), seed=rank)
    t_gen = time.time() - t0
    # The first host-to-device copy of a process sets up the HIP runtime's copy path: ~100 ms with torch's code objects loaded, 27 ms
    # without (tools/fe_first_upload.py) -- whoever copies first pays it. Round 5: the LIBRARY's own warm-up (ecne_warmup: copy path, code
    # objects, scratch memory of the solve kernels) instead of a torch copy, so that `config.frontend` below reports the library's
    # front-end and not the runtime's start-up (reported here, untimed like all of the preparation; config.cold_process has the fresh-process figures).
    runtime_warmup_ms = E.warmup(local_rank)
    t0 = time.time()
    main_file = E.R1CS(path)
    parse_stats = E.frontend_stats()
    trusted = E.R1CS(fixtures.path("secp256k1.r1cs"))
    t_parse = time.time() - t0
    t0 = time.time()
    system = E.System(main_file)
    system.abstract(trusted, "Secp256k1AddUnequal")
    t_abstract = time.time() - t0
    abstract_stats = E.System.last_abstract_stats()
    n_main = len(main_file)
    info = system.info
    stream = torch.cuda.current_stream().cuda_stream

    from ecneproject_amd import sharding

    def step():
        r = E.solve_batch([system], device=local_rank, stream=stream, fetch_states=False,
                          queue_mode=args.queue_mode)[0]
        # the done / verdict flag: MIN all-reduce of one int32, RCCL over xGMI (a single circuit does not shard: replicas).
        # The word stays on the device inside the timed region (read after it: reading it here would add a device-to-host round trip per step)
        ok = r.status == 0 and bool(r.function_good)
        if world > 1:
            word = torch.tensor([1 if ok else 0], dtype=torch.int32, device="cuda")
            dist.all_reduce(word, op=dist.ReduceOp.MIN)
        else:
            word = ok
        return r, word

    _shape, classify_ms_cold, classify_bytes = E.classify(system, device=local_rank)     # first launch: code object load, cold caches
    t_first = time.perf_counter()
    res, _w = step()                   # first solve: layout upload + classification happen here (untimed)
    torch.cuda.synchronize()
    first_call = {"wall_ms": round((time.perf_counter() - t_first) * 1e3, 3), "kernel_ms": round(float(res.summary.device_ms), 3)}
    frontend_stats = E.frontend_stats()                 # (abstraction and layout of this system; the parse figures are the main file's)
    for k in ("parse_device", "upload_ms", "offsets_ms", "fill_ms", "parse_ms", "file_bytes"):
        frontend_stats[k] = parse_stats[k]
    gc.collect(); gc.disable()          # (see run_jobs_workload)
    # W - 1 more untimed steps, barrier + synchronize, exactly K timed steps, synchronize + barrier, MAX over ranks (sharding.timed_replica_steps;
    # the same function runs under gloo in tests/test_multirank_gloo.py)
    elapsed, elapsed_rank, timed, words = sharding.timed_replica_steps(step, args.steps, max(args.warmup - 1, 0), dist, world, sync=torch.cuda.synchronize, device="cuda")
    gc.enable()
    words = [bool(int(w.item())) if hasattr(w, "item") else bool(w) for w in words]
    res = timed[-1]
    dev_ms = [r.summary.device_ms for r in timed]
    inv = [step_invariants(r) for r in timed]                             # (after the timed region)
    if any(x != inv[0] for x in inv):
        raise SystemExit("bench.py: the timed steps did not reproduce the same result: %r" % (sorted(set(inv))[:2],))
    # whole-state parity of THIS build on THIS workload, every run: the device-side digest of the per-variable state (ecne_result_digest)
    # of one more solve (untimed) against the oracle's, committed as tests/golden/scale_goldens.json by tests/golden/make_scale_goldens.py
    # (vectors, not the oracle: S = 104 costs the oracle minutes)
    state_check = None
    gold_path = os.path.join(HERE, "tests", "golden", "scale_goldens.json")
    if os.path.exists(gold_path) and args.stride == 10 and args.queue_mode == 0 and rank == 0:
        with open(gold_path) as f:
            gold = json.load(f).get("ecdsa_like(%d,10)+Secp256k1AddUnequal" % args.S)
        if gold is not None:
            rd = E.solve_batch([system], device=local_rank, stream=stream, fetch_states="digest")[0]
            got = ["%016x" % rd.digest[0], "%016x" % rd.digest[1]]
            gs = rd.summary
            tup = (int(rd.status), bool(rd.function_good), int(gs.pops), int(gs.successful_steps), int(gs.num_unique), int(gs.outer_iterations),
                   [int(x) for x in rd.counts()], [int(x) for x in list(gs.rule_hits)[:13]])
            want = (gold["status"], gold["verdict"], gold["pops"], gold["steps"], gold["num_unique"], gold["outer"], gold["counts"], gold["rule_hits"])
            if got != gold["digest"] or tup != want:
                raise SystemExit("bench.py: the state of this build differs from the oracle's golden digest: %r %r vs %r %r" % (got, tup, gold["digest"], want))
            state_check = {"digest": got, "what": "ecne_result_digest of the whole per-variable state (flags, bounds, tags, values) and the counters of one more "
                                                  "solve equal the sequential oracle's, committed in tests/golden/scale_goldens.json"}
    # k_classify_rows, warm (after the timed steps: clocks up, the same resident rows; the kernel is idempotent)
    warm = sorted(E.classify(system, device=local_rank)[1] for _ in range(7))
    classify_ms, classify_ms_best = warm[len(warm) // 2], warm[0]                             # the figure reported: their median
    ms_per_step = elapsed * 1e3 / max(args.steps, 1)
    value = n_main * world * args.steps / elapsed
    rank_ms = [elapsed_rank * 1e3 / max(args.steps, 1)]
    if world > 1:
        gathered = [None] * world
        dist.all_gather_object(gathered, (rank_ms[0], inv[0]))
        rank_ms = [g[0] for g in gathered]
        if rank == 0 and any(g[1] != inv[0] for g in gathered):      # (the instances differ in their constants only: same counters, same verdict)
            raise SystemExit("bench.py: the ranks' instances did not reproduce the same counters: %r" % ([g[1] for g in gathered][:3],))
    job_lines = None
    if args.job_lines == "on" or (args.job_lines == "auto" and world > 1):
        job_lines = attached_job_lines(args, torch, dist, rank, local_rank, world)

    if rank == 0:
        s = res.summary
        # algorithmic bytes: B_pop(i) = 20 + 40 nnz(i) per queue pop, B_sweep =
        # sum_i (12 + 40 nnz(i)) per full-system sweep, 3 sweeps per outer iteration + 1 at setup.
        # pops / outer_iterations are the sequential schedule's counters (identical on the GPU by
        # construction: the parity tests compare them with the oracle's).
        nnz = int(info.nnz[0] + info.nnz[1] + info.nnz[2])
        b_sweep = 12 * int(info.n_rows) + 40 * nnz
        b_pops = 20 * int(s.pops) + 40 * int(s.pop_nnz)
        b_alg = b_pops + (3 * int(s.outer_iterations) + 1) * b_sweep
        k_ms = sum(dev_ms) / len(dev_ms) if dev_ms else float(res.summary.device_ms)
        # HBM bytes per k_solve launch from the PMC passes of the same command (rocprofv3 --pmc FETCH_SIZE /
        # --pmc WRITE_SIZE, separate runs; summaries under profiles/). Not measurable from inside this process.
        traffic, traffic_src = None, None
        tj = os.path.join(HERE, "profiles", "traffic_latest.json")
        if os.path.exists(tj) and args.stride == 10 and world == 1:
            with open(tj) as f:
                t = json.load(f)
            ts = t.get("by_S", {}).get(str(args.S))          # one entry per workload size the PMC passes were taken on (26: the headline, 104: beyond the Infinity Cache)
            if ts is None:
                traffic_src = "no PMC passes on file for S = %d (tools/profile_r04.sh %d)" % (args.S, args.S)
            elif t.get("csrc_sha16") == csrc_sha16() and t.get("schedule_env", {}) == schedule_env():
                traffic, traffic_src = ts.get("k_solve_bytes_per_launch"), ts.get("source")
            elif t.get("csrc_sha16") == csrc_sha16():
                traffic_src = "not comparable: the PMC passes ran under ECNE_* switches %r, this process under %r" % (t.get("schedule_env", {}), schedule_env())
            else:      # the PMC passes were taken on another build of the library: not this line's traffic
                traffic_src = "stale: profiles/traffic_latest.json was measured on csrc %s, this build is %s (re-run tools/profile_r04.sh)" % (t.get("csrc_sha16"), csrc_sha16())
        achieved = b_alg / (k_ms * 1e-3) / 1e9
        # latency model: the solve is rounds x time per round; a round is one dependency level of the
        # queue schedule (or a P-phase pass)
        sd = list(s.sched)
        n_multi = int(s.rule_hits[14]) >> 16
        n_rounds = int(s.rule_hits[13])
        q_ms = float(s.phase_ms[1])
        multi_ms = float(s.queue_ms[7])
        latency = {"rounds": n_rounds, "us_per_round": 1e3 * q_ms / max(n_rounds, 1),
                   "multi_workgroup_rounds": n_multi, "us_per_multi_round": 1e3 * multi_ms / max(n_multi, 1),
                   "fast_wavefront_rounds": sd[0], "rows_in_fast_rounds": sd[1], "us_per_fast_round": sd[2] * 1e-2 / max(sd[0], 1),
                   "outer_iterations": int(s.outer_iterations),
                   # (round 5) outer iterations the master workgroup of the team finished alone: P3 / P4 from the rows popped since the last pass
                   # (no job barrier, no sweep over the system); the others ran the full sweeps with every workgroup
                   "iterations_on_the_master_alone": int(s.team[0]), "rows_their_passes_looked_at": int(s.team[1]), "iterations_with_full_sweeps": int(s.team[2]),
                   "ms_in_iterations_on_the_master_alone": round(int(s.team[3]) * 1e-5, 3), "P1_P2_ms": round(float(s.phase_ms[7]), 3),
                   # drain rounds (csrc/drain.hip.hpp): a window executed in dataflow order, level by level, pushes resolved once per window;
                   # a level costs 2 job barriers (nobody contested anything) to 4, the resolution 4
                   "drain_rounds": int(round(float(s.multi_ms[7]) * 1e5)), "drain_levels": int(round(float(s.multi_ms[6]) * 1e5)),
                   "solo_drain_rounds": sd[3], "rows_in_solo_drains": sd[4], "us_per_solo_drain": sd[5] * 1e-2 / max(sd[3], 1),
                   "model": "t_solve ~ rounds x us_per_round + P-phases; %d rounds at %.1f us" % (n_rounds, 1e3 * q_ms / max(n_rounds, 1))}
        out = {
            "metric": "constraints resolved/sec (wall-clock to fixed point), ecdsa-scale R1CS",
            "value": value, "unit": "constraints/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64x4 (BN254 Fp limbs) + u8/u32 flags", "data": "synthetic",
            "config": {"workload": ("ecdsa_like(S=%d,stride=%d) + trusted secp256k1.r1cs (config 5; ecdsa.r1cs absent from the reference)" % (args.S, args.stride)) if args.S == 26 else
                                   ("ecdsa_like(S=%d,stride=%d) + trusted secp256k1.r1cs: the SCALE-OUT variant of config 5 (%d strides instead of 26) whose static arrays and state exceed the 256 MiB Infinity Cache -- the HBM-true roofline line, not the headline" % (args.S, args.stride, args.S)),
                       "rows_main": n_main, "rows_reduced": int(info.n_rows), "nnz_reduced": nnz,
                       "specials": int(info.n_specials), "n_vars": int(info.n_vars),
                       "parallelism": ("%d independent instances of the circuit, one per GPU (own table constants each, tests/ecdsa_like.py seed = rank), RCCL all-reduce of the verdict word; "
                                       "a SINGLE circuit does not shard -- see config.one_circuit" % world) if world > 1 else "1 GPU",
                       # on ONE circuit more GPUs buy nothing -- the fixed point is one dependency chain. `value` above is the aggregate over N
                       # independent instances (weak scaling of a job queue); this is the rate of any one of them.
                       "one_circuit": {"value": n_main * args.steps / elapsed, "unit": "constraints/s", "ms_per_step": ms_per_step, "scaling": "replicated, no speed-up",
                                       "note": "time to the verdict of one circuit is the same at every N; the workloads that shard are in config.job_workloads"},
                       "scaling_check": scaling_check("ecdsa" if args.S == 26 else "ecdsa_S%d" % args.S, world, ms_per_step, rank_ms,
                                                      {"model": "replicas: value(N) = N x rows / t_step(slowest rank); the all-reduce of one word and the two barriers are the only coupling",
                                                       "all_ranks_agreed_on_the_verdict_word": bool(all(words)) == bool(res.status == 0 and res.function_good)}),
                       "verdict": bool(res.function_good), "status": int(res.status),
                       "outer_iterations": int(s.outer_iterations), "pops": int(s.pops),
                       "python_gc": "cyclic collector held off during the timed steps (a full collection with torch loaded: ~40 ms)",
                       "invariants": {"steps_identical": True, "checked": "status, verdict, pops, successful_steps, num_unique, outer_iterations, the four printed counts, rule hits -- every timed step"},
                       "host_prep_s": {"generate": round(t_gen, 3), "parse": round(t_parse, 3), "abstract": round(t_abstract, 3)},
                       "runtime_warmup": {"ms": round(runtime_warmup_ms, 1), "what": "ecne_warmup(device) before the front-end (the library's own: HIP copy path, its two code objects, scratch memory of the solve kernels); without it that time shows up inside the first file's load and first solve -- config.cold_process measures both ways in fresh processes"},
                       "classify_kernel": {"ms": classify_ms, "ms_best": classify_ms_best, "ms_first_call": classify_ms_cold, "bytes": classify_bytes,
                                           "GBps": classify_bytes / max(classify_ms, 1e-9) / 1e6,
                                           "frac_of_hbm_peak": classify_bytes / max(classify_ms, 1e-9) / 1e6 / 8000.0,
                                           "resident": ("infinity cache: the %d MB this pass streams fit the 256 MiB MALL and the launches are warm -- the fraction is of the HBM peak, the data does not come from HBM (see --S 104)" % (classify_bytes >> 20))
                                                       if classify_bytes < (256 << 20) else "hbm: %d MB per pass, beyond the 256 MiB Infinity Cache" % (classify_bytes >> 20),
                                           "note": "HIP events around the launch; ms = median of 7 warm launches, ms_best their minimum; the first call of a process also loads the code object (that was round 1's 1.3 ms)"},
                       "solve_first_call": dict(first_call, note="the first solve of the process: layout upload + classification + the runtime growing the device's scratch memory for k_solve_team; the timed steps follow it"),
                       "abstraction": abstract_stats,
                       "frontend": {"mode": {0: "host", 1: "device", 2: "auto"}[E.set_frontend()], **{k: round(v, 3) for k, v in frontend_stats.items()}}},
            "roofline": {"bound": "hbm", "kernel": "k_solve_team", "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                         "frac": achieved / 8000.0, "traffic": traffic, "traffic_source": traffic_src,
                         "traffic_frac": (traffic / (k_ms * 1e-3) / 1e9 / 8000.0) if traffic else None,
                         # the same two fractions under the names a reader looks for: frac = credited (algorithmic) bytes of SURVEY.md 8(d)'s formula / time / 8 TB/s;
                         # frac_counters = bytes the memory system actually moved (FETCH_SIZE + WRITE_SIZE) / time / 8 TB/s; frac_of_6p3TBps = frac against the
                         # ~6.3 TB/s a pure streaming kernel reaches on this part (MI355X_MICROARCH.md) instead of the 8 TB/s data-sheet peak
                         "frac_counters": (traffic / (k_ms * 1e-3) / 1e9 / 8000.0) if traffic else None,
                         "frac_of_6p3TBps": achieved / 6300.0,
                         "credited_sweeps": 3 * int(s.outer_iterations) + 1,
                         "latency_model": latency,
                         "alg_bytes_per_launch": b_alg, "kernel_ms": k_ms,
                         "phase_ms": {k: round(v, 3) for k, v in zip(["setup", "queue", "P3", "P4", "P5", "verdict", "P3_rounds"], list(s.phase_ms)[:7])},
                         "queue_ms": {k: round(v, 3) for k, v in zip(["head", "mark", "check", "exec", "flatten", "resolve", "alone_bursts_wave_rounds", "multi_rounds"], list(s.queue_ms)[:8])},
                         "multi_ms": {k: round(v, 3) for k, v in zip(["mark", "check_and_cut", "exec_and_scan", "expand", "count_and_scan", "write"], list(s.multi_ms)[:6])},
                         "note": ("fixed point is dependency-depth bound; see DESIGN.md. `frac` is an accounting of the reference's work, not of this kernel's traffic: the formula credits "
                                  "%d full sweeps of the system (3 per outer iteration + 1), the engine performs 3 (26 of 28 iterations at S = 26 are finished from the popped rows "
                                  "alone); the counters say %s of the HBM peak -- the kernel is latency-bound at every size, S = 104 included. A `frac` above 1 (S = 416: 1 255 credited "
                                  "sweeps of 17.6 M rows) only says that the engine does not perform the work the formula credits") %
                                 (3 * int(s.outer_iterations) + 1, ("%.1f %%" % (100.0 * traffic / (k_ms * 1e-3) / 1e9 / 8000.0)) if traffic else "(no PMC passes on file for this build)")},
        }
        if job_lines is not None:
            out["config"]["job_workloads"] = job_lines
        if world == 1 and not args.no_cold:
            # (after the timed region, in fresh subprocesses: the library's own answer to a cold start -- ecne_warmup -- next to no warm-up at all)
            out["config"]["cold_process"] = {"with_ecne_warmup": cold_process(path, local_rank, True), "without": cold_process(path, local_rank, False),
                                             "note": "fresh python processes WITHOUT torch; first_file_to_verdict_ms = readR1CS + abstraction + first solve, wall clock"}
        if state_check is not None:
            out["config"]["invariants"]["state_matches_oracle_digest"] = state_check
            if args.cpu_sample_S != args.S or args.no_cpu_baseline or world > 1:      # (else the live oracle leg below says it)
                out["config"]["invariants"]["matches_oracle"] = "counters and the digest of the whole state equal the oracle's golden vectors for this workload (tests/golden/scale_goldens.json)"
        if not args.no_cpu_baseline and world == 1:      # reported on rank 0 at N = 1 only
            import orc
            sp = ecdsa_like.cached(args.cpu_sample_S, args.stride, directory="/tmp/ecne_bench_%d" % os.getuid())
            o = orc.run(sp, [fixtures.path("secp256k1.r1cs")], ["Secp256k1AddUnequal"], want_states=False)
            if args.cpu_sample_S == args.S:      # the oracle leg doubles as a check of the timed steps (same workload)
                os_ = o.summary
                want = (int(o.status), bool(o.verdict), int(os_.pops), int(os_.successful_steps), int(os_.num_unique), int(os_.outer_iterations),
                        int(os_.unique_nontrivial), int(os_.n_nontrivial), int(os_.unique_targets), int(os_.n_targets), tuple(int(x) for x in list(os_.rule_hits)[:13]))
                if want != inv[0]:
                    raise SystemExit("bench.py: the timed solve disagrees with the sequential oracle: %r vs %r" % (inv[0], want))
                out["config"]["invariants"]["matches_oracle"] = "the same tuple equals the sequential oracle's on this workload"
            out["cpu_baseline"] = {"value": o.summary.n_rows_main / max(o.summary.t_solve, 1e-9), "unit": "constraints/s",
                                   "cores": 1, "kind": "port", "kind_detail": PORT_DETAIL,
                                   "sample": "ecdsa_like(S=%d,stride=%d): %d rows, sequential oracle solve %.2f s "
                                             "(parse %.1f s and abstraction %.1f s excluded, as for the GPU)" %
                                             (args.cpu_sample_S, args.stride, o.summary.n_rows_main, o.summary.t_solve,
                                              o.summary.t_read, o.summary.t_abstract),
                                   "host_cores_available": os.cpu_count(), "reference_probe": probe_julia()}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
