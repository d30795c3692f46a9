"""Developer aid (here, after `gpurun -- bash tools/gp_final_r06.sh`): copies gpurun_out/final_r06/* into profiles/ under their round-6 names,
rewrites profiles/traffic_latest.json (per workload size, with the hash of csrc/ the passes were taken on) and prints the numbers the
documents quote."""
import json, os, shutil, sys
HERE = os.path.dirname(os.path.abspath(__file__)); ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
import bench
F = os.path.join(ROOT, "gpurun_out", "final_r06"); P = os.path.join(ROOT, "profiles")
last = lambda p: open(p).read().strip().splitlines()[-1]
for w in ("ecdsa", "dag", "suite", "secp", "poseidon", "many", "ecdsa_S104", "ecdsa_S416", "ecdsa_with_job_lines"):
    try:
        line = last(os.path.join(F, w + ".json")); json.loads(line)
        open(os.path.join(P, "r06_bench_%s.json" % w), "w").write(line + "\n")
    except Exception as e:      # noqa: BLE001
        print("missing bench line", w, e)
for src, dst in (("bench_under_rocprof.json", "r06_bench_under_rocprof.json"), ("S104_bench_under_rocprof.json", "r06_S104_bench_under_rocprof.json")):
    try: open(os.path.join(P, dst), "w").write(last(os.path.join(F, src)) + "\n")
    except Exception as e: print("missing", src, e)      # noqa: E701,BLE001
for src, dst in (("trace", "r06_kernel_trace_stats"), ("fetch", "r06_pmc_FETCH_SIZE"), ("write", "r06_pmc_WRITE_SIZE"), ("sq", "r06_pmc_SQ_waves_busy_wait"), ("insts", "r06_pmc_SQ_insts"),
                 ("tcc", "r06_pmc_TCC_hit_miss"), ("suite", "r06_suite_kernel_trace_stats"), ("S104_trace", "r06_S104_kernel_trace_stats"), ("S104_fetch", "r06_S104_pmc_FETCH_SIZE"),
                 ("S104_write", "r06_S104_pmc_WRITE_SIZE"), ("S104_sq", "r06_S104_pmc_SQ_waves_busy_wait"), ("per_file_vs_oracle", "r06_per_file_vs_oracle"),
                 ("round_log_summary", "r06_round_log_summary"), ("scale_variants", "r06_scale_variants"), ("level_round_stages", "r06_level_round_stages"),
                 ("crew_ab", "r06_crew_rounds_on_off"), ("soak_crew", "r06_soak_crew"), ("soak_determinism", "r06_soak_determinism"), ("classify_time", "r06_classify_time"), ("suite_per_file", "r06_suite_per_file"), ("dag_side_ab", "r06_dag_side_ab")):
    try: shutil.copy(os.path.join(F, src + ".txt"), os.path.join(P, dst + ".txt"))
    except Exception as e: print("missing", src, e)      # noqa: E701,BLE001


def kern(path, name, col):
    """the line of `name` with the most calls x time (the workload's launches; ecne_warmup's launch on a three-row system is a line of its own)"""
    best = None
    for ln in open(path):
        if name in ln and (col is None or col in ln):
            f = ln.split()
            if best is None or float(f[2 if col is None else 3]) > float(best[2 if col is None else 3]):
                best = f
    return best


t = {"csrc_sha16": bench.csrc_sha16(), "schedule_env": {}, "by_S": {}}      # (the PMC passes run without ECNE_* switches: tools/profile_r06.sh)
for S, pre in ((26, "r06_"), (104, "r06_S104_")):
    try:
        tr = kern(os.path.join(P, pre + "kernel_trace_stats.txt"), "k_solve_team", None)
        fe = kern(os.path.join(P, pre + "pmc_FETCH_SIZE.txt"), "k_solve_team", "FETCH_SIZE"); wr = kern(os.path.join(P, pre + "pmc_WRITE_SIZE.txt"), "k_solve_team", "WRITE_SIZE")
        t["by_S"][str(S)] = {"k_solve_bytes_per_launch": int((float(fe[-2]) + float(wr[-2])) * 1024), "fetch_size_kb": float(fe[-2]), "write_size_kb": float(wr[-2]),
                             "kernel_avg_us_in_trace": float(tr[3]),
                             "source": "profiles/%spmc_FETCH_SIZE.txt + profiles/%spmc_WRITE_SIZE.txt (rocprofv3 --pmc, separate passes of `python bench.py --S %d --steps 5 --warmup 2 --no-cpu-baseline`, kernel "
                                       "k_solve_team, round 6; raw counters x 1024). Calibration on this stack (profiles/r03_counter_calibration.txt): WRITE_SIZE exact; FETCH_SIZE = 0.50 x the bytes of a coalesced "
                                       "stream; k_solve's reads are narrow gathers: reported uncorrected" % (pre, pre, S)}
        print("S", S, "trace k_solve_team avg us", tr[3], "| FETCH KB", fe[-2], "WRITE KB", wr[-2])
    except Exception as e:      # noqa: BLE001
        print("no PMC passes for S", S, e)
json.dump(t, open(os.path.join(P, "traffic_latest.json"), "w"))
for w in ("ecdsa", "ecdsa_S104", "ecdsa_S416", "dag", "suite", "secp", "poseidon", "many"):
    try:
        d = json.load(open(os.path.join(P, "r06_bench_%s.json" % w))); cb = d.get("cpu_baseline", {})
        print("%-11s ms %.3f value %.4g frac %.4f kernel_ms %.3f cpu %.3g fp %s" % (w, d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline"].get("kernel_ms", 0), cb.get("value", 0), cb.get("file_parallel", {}).get("value")))
    except Exception as e:      # noqa: BLE001
        print(w, e)
