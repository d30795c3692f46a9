"""Developer aid (GPU box): wall-clock of every step from the .r1cs file to the verdict for ecdsa_like(S).
python tools/e2e_timing.py [S] [host_threads] [frontend: host|device|auto]      (host worker threads are opt-in: ecne_set_host_threads, include/ecne.h)"""
import os, sys, time
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests"))
import ecneproject_amd as E, fixtures, ecdsa_like
S = int(sys.argv[1]) if len(sys.argv) > 1 else 26
NT = int(sys.argv[2]) if len(sys.argv) > 2 else 1
E.set_host_threads(NT)
FE = sys.argv[3] if len(sys.argv) > 3 else "auto"
E.set_frontend({"host": 0, "device": 1, "auto": 2}[FE])
p = ecdsa_like.cached(S, 10)
E.solve_batch([E.System(E.R1CS(fixtures.path("target/division.r1cs")))])   # HIP runtime start-up, untimed
t = [time.perf_counter()]
m = E.R1CS(p); t.append(time.perf_counter())
tr = E.R1CS(fixtures.path("secp256k1.r1cs")); t.append(time.perf_counter())
s = E.System(m); t.append(time.perf_counter())
s.abstract(tr, "Secp256k1AddUnequal"); t.append(time.perf_counter())
s.info; t.append(time.perf_counter())
r = E.solve_batch([s], fetch_states=False)[0]; t.append(time.perf_counter())
r2 = E.solve_batch([s], fetch_states=False)[0]; t.append(time.perf_counter())
names = ["parse main", "parse trusted", "system from r1cs", "abstraction", "layout (flat arrays, host)", "first solve (upload + classify + solve)", "second solve"]
for n, a, b in zip(names, t, t[1:]):
    print("%-55s %8.1f ms" % (n, (b - a) * 1e3))
print("front-end %s:" % FE, {k: round(v, 3) for k, v in E.frontend_stats().items()})
print("host threads %d: file MB %.1f rows %d -> %d, verdict %s, kernel %.1f ms, end to end %.1f ms" % (NT, os.path.getsize(p) / 1e6, len(m), len(s), r.function_good, r2.summary.device_ms, (t[6] - t[0]) * 1e3))
