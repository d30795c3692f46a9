"""BASELINE.json config 4: the 67 ecne_circomlib_tests/*.r1cs files as ONE batch launch (one
workgroup(-group) per file) on one GPU, against the sequential oracle run file by file on one core.
python tools/suite_bench.py [reps]"""
import os, sys, time
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests"))
import ecneproject_amd as E, fixtures, orc

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
rels = fixtures.circomlib_suite()
systems = [E.System(E.R1CS(fixtures.path(r))) for r in rels]
rows = sum(len(s) for s in systems)
E.solve_batch(systems, fetch_states=False)          # upload + classify + warm-up
ts = []
for _ in range(reps):
    t = time.perf_counter()
    res = E.solve_batch(systems, fetch_states=False)
    ts.append(time.perf_counter() - t)
dev = max(r.summary.device_ms for r in res)
t_cpu = 0.0
for r in rels:
    o = orc.run(fixtures.path(r), want_states=False)
    t_cpu += o.summary.t_solve
best = min(ts)
print({"files": len(rels), "rows": rows, "gpu_wall_ms_best": round(best * 1e3, 2), "gpu_kernel_ms": round(dev, 2),
       "gpu_constraints_per_s": round(rows / best), "cpu_solve_s_sum_1core": round(t_cpu, 3),
       "cpu_constraints_per_s": round(rows / t_cpu), "all_ok": all(r.status == 0 for r in res),
       "verdicts_true": sum(r.function_good for r in res)})
