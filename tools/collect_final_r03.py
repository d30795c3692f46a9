"""Developer aid (here, after `gpurun -- bash tools/gp_final_r03.sh`): copies gpurun_out/final_r03/* into profiles/ under their round-3 names,
rewrites the end-of-round table of profiles/r03_scale_variants.txt and profiles/traffic_latest.json, prints the numbers the documents quote."""
import json, os, re, shutil
HERE = os.path.dirname(os.path.abspath(__file__)); ROOT = os.path.dirname(HERE)
F = os.path.join(ROOT, "gpurun_out", "final_r03"); P = os.path.join(ROOT, "profiles")
last = lambda p: open(p).read().strip().splitlines()[-1]
for w in ("ecdsa", "dag", "suite", "secp", "poseidon"):
    open(os.path.join(P, "r03_bench_%s.json" % w), "w").write(last(os.path.join(F, w + ".json")) + "\n")
open(os.path.join(P, "r03_bench_under_rocprof.json"), "w").write(last(os.path.join(F, "bench_under_rocprof.json")) + "\n")
for src, dst in (("trace", "r03_kernel_trace_stats"), ("fetch", "r03_pmc_FETCH_SIZE"), ("write", "r03_pmc_WRITE_SIZE"), ("sq", "r03_pmc_SQ_waves_busy_wait"), ("insts", "r03_pmc_SQ_insts"),
                 ("tcc", "r03_pmc_TCC_hit_miss"), ("suite", "r03_suite_kernel_trace_stats"), ("per_file_vs_oracle", "r03_per_file_vs_oracle"), ("round_log_summary", "r03_round_log_summary")):
    shutil.copy(os.path.join(F, src + ".txt"), os.path.join(P, dst + ".txt"))
# scale variants: the four JSON lines into the formatted table
rows = [json.loads(l) for l in open(os.path.join(F, "scale_variants.txt")) if l.strip().startswith("{")]
sv = os.path.join(P, "r03_scale_variants.txt")
lines = open(sv).read().split("\n")
names = ["A ecdsa_like(26,10)", "B ecdsa_like(64,8)", "C 45 x EdDSAMiMCSponge", "D 1400 x Poseidon"]
i = 0
for k, ln in enumerate(lines):
    if i < 4 and ln.startswith(names[i]) and k < 12:
        d = rows[i]
        lines[k] = "%-26s | %8d -> %8d | %9.3f | %11d | %8d | %3d | %5d / %4d (%6.2f) / %5d (%7d rows, %7.2f ms) | %7.1f | %7.1f" % (
            names[i], d["rows_main"], d["rows"], d["kernel_ms"], d["constraints_per_s"], d["pops"], d["outer_iterations"], d["rounds"], d["multi_workgroup_rounds"], d["multi_ms"],
            d["fast_wave_rounds"], d["rows_in_fast_rounds"], d["fast_ms"], d["speedup_vs_1_core"], d["file_to_verdict_ms"])
        i += 1
open(sv, "w").write("\n".join(lines))
def kern(path, name, col):
    for ln in open(path):
        if name in ln and (col is None or col in ln): return ln.split()
tr = kern(os.path.join(P, "r03_kernel_trace_stats.txt"), "k_solve_team", None)
fe = kern(os.path.join(P, "r03_pmc_FETCH_SIZE.txt"), "k_solve_team", "FETCH_SIZE"); wr = kern(os.path.join(P, "r03_pmc_WRITE_SIZE.txt"), "k_solve_team", "WRITE_SIZE")
t = json.load(open(os.path.join(P, "traffic_latest.json")))
t["fetch_size_kb"] = float(fe[-1]); t["write_size_kb"] = float(wr[-1]); t["k_solve_bytes_per_launch"] = int((float(fe[-1]) + float(wr[-1])) * 1024); t["kernel_avg_us_in_trace"] = float(tr[3])
json.dump(t, open(os.path.join(P, "traffic_latest.json"), "w"))
print("trace k_solve_team avg us", tr[3], "| FETCH KB", fe[-1], "WRITE KB", wr[-1], "-> bytes", t["k_solve_bytes_per_launch"])
cl = kern(os.path.join(P, "r03_kernel_trace_stats.txt"), "k_classify_rows", None); print("classify avg us", cl[3])
su = kern(os.path.join(P, "r03_suite_kernel_trace_stats.txt"), "k_solveEPK", None); print("suite k_solve avg us", su[3])
for w in ("ecdsa", "dag", "suite", "secp", "poseidon"):
    d = json.load(open(os.path.join(P, "r03_bench_%s.json" % w))); cb = d["cpu_baseline"]
    print("%-9s ms %.3f value %.4g frac %.4f kernel_ms %.3f cpu %.3g fp %s" % (w, d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline"].get("kernel_ms", 0), cb["value"], cb.get("file_parallel", {}).get("value")))
for d in rows: print(d["kernel_ms"], d["constraints_per_s"], d["rounds"])
print(open(os.path.join(P, "r03_round_log_summary.txt")).read().split("\n")[0:5])
