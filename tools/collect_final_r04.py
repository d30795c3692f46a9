"""Developer aid (here, after `gpurun -- bash tools/gp_final_r04.sh`): copies gpurun_out/final_r04/* into profiles/ under their round-4 names,
rewrites profiles/traffic_latest.json (per workload size, with the hash of csrc/ the passes were taken on) and prints the numbers the
documents quote."""
import json, os, shutil, sys
HERE = os.path.dirname(os.path.abspath(__file__)); ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
import bench
F = os.path.join(ROOT, "gpurun_out", "final_r04"); P = os.path.join(ROOT, "profiles")
last = lambda p: open(p).read().strip().splitlines()[-1]
for w in ("ecdsa", "dag", "suite", "secp", "poseidon", "many", "ecdsa_S104"):
    try:
        line = last(os.path.join(F, w + ".json")); json.loads(line)
        open(os.path.join(P, "r04_bench_%s.json" % w), "w").write(line + "\n")
    except Exception as e:      # noqa: BLE001
        print("missing bench line", w, e)
for src, dst in (("bench_under_rocprof.json", "r04_bench_under_rocprof.json"), ("S104_bench_under_rocprof.json", "r04_S104_bench_under_rocprof.json")):
    try: open(os.path.join(P, dst), "w").write(last(os.path.join(F, src)) + "\n")
    except Exception as e: print("missing", src, e)      # noqa: E701,BLE001
for src, dst in (("trace", "r04_kernel_trace_stats"), ("fetch", "r04_pmc_FETCH_SIZE"), ("write", "r04_pmc_WRITE_SIZE"), ("sq", "r04_pmc_SQ_waves_busy_wait"), ("insts", "r04_pmc_SQ_insts"),
                 ("tcc", "r04_pmc_TCC_hit_miss"), ("suite", "r04_suite_kernel_trace_stats"), ("S104_trace", "r04_S104_kernel_trace_stats"), ("S104_fetch", "r04_S104_pmc_FETCH_SIZE"),
                 ("S104_write", "r04_S104_pmc_WRITE_SIZE"), ("S104_sq", "r04_S104_pmc_SQ_waves_busy_wait"), ("per_file_vs_oracle", "r04_per_file_vs_oracle"),
                 ("round_log_summary", "r04_round_log_summary"), ("scale_variants", "r04_scale_variants"), ("level_round_stages", "r04_level_round_stages"),
                 ("crew_ab", "r04_crew_rounds_on_off"), ("soak_crew", "r04_soak_crew"), ("suite_per_file", "r04_suite_per_file")):
    try: shutil.copy(os.path.join(F, src + ".txt"), os.path.join(P, dst + ".txt"))
    except Exception as e: print("missing", src, e)      # noqa: E701,BLE001


def kern(path, name, col):
    for ln in open(path):
        if name in ln and (col is None or col in ln):
            return ln.split()


t = {"csrc_sha16": bench.csrc_sha16(), "by_S": {}}
for S, pre in ((26, "r04_"), (104, "r04_S104_")):
    try:
        tr = kern(os.path.join(P, pre + "kernel_trace_stats.txt"), "k_solve_team", None)
        fe = kern(os.path.join(P, pre + "pmc_FETCH_SIZE.txt"), "k_solve_team", "FETCH_SIZE"); wr = kern(os.path.join(P, pre + "pmc_WRITE_SIZE.txt"), "k_solve_team", "WRITE_SIZE")
        t["by_S"][str(S)] = {"k_solve_bytes_per_launch": int((float(fe[-1]) + float(wr[-1])) * 1024), "fetch_size_kb": float(fe[-1]), "write_size_kb": float(wr[-1]),
                             "kernel_avg_us_in_trace": float(tr[3]),
                             "source": "profiles/%spmc_FETCH_SIZE.txt + profiles/%spmc_WRITE_SIZE.txt (rocprofv3 --pmc, separate passes of `python bench.py --S %d --steps 5 --warmup 2 --no-cpu-baseline`, kernel "
                                       "k_solve_team, round 4; raw counters x 1024). Calibration on this stack (profiles/r03_counter_calibration.txt): WRITE_SIZE exact; FETCH_SIZE = 0.50 x the bytes of a coalesced "
                                       "stream; k_solve's reads are narrow gathers: reported uncorrected" % (pre, pre, S)}
        print("S", S, "trace k_solve_team avg us", tr[3], "| FETCH KB", fe[-1], "WRITE KB", wr[-1])
    except Exception as e:      # noqa: BLE001
        print("no PMC passes for S", S, e)
json.dump(t, open(os.path.join(P, "traffic_latest.json"), "w"))
for w in ("ecdsa", "ecdsa_S104", "dag", "suite", "secp", "poseidon", "many"):
    try:
        d = json.load(open(os.path.join(P, "r04_bench_%s.json" % w))); cb = d.get("cpu_baseline", {})
        print("%-11s ms %.3f value %.4g frac %.4f kernel_ms %.3f cpu %.3g fp %s" % (w, d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline"].get("kernel_ms", 0), cb.get("value", 0), cb.get("file_parallel", {}).get("value")))
    except Exception as e:      # noqa: BLE001
        print(w, e)
