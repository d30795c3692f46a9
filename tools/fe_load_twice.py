"""Developer aid (GPU box): parse the ecdsa_like(S) file several times in one process (first-use costs of the runtime vs steady state)."""
import os, sys, time
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests"))
import ecneproject_amd as E, fixtures, ecdsa_like
S = int(sys.argv[1]) if len(sys.argv) > 1 else 26
p = ecdsa_like.cached(S, 10)
E.solve_batch([E.System(E.R1CS(fixtures.path("target/division.r1cs")))])
for i in range(4):
    t0 = time.perf_counter(); m = E.R1CS(p); t1 = time.perf_counter()
    st = E.frontend_stats()
    print("load %d: %.1f ms  (upload %.2f, offsets %.2f, fill %.2f, parse total %.2f)" % (i, (t1 - t0) * 1e3, st["upload_ms"], st["offsets_ms"], st["fill_ms"], st["parse_ms"]))
    del m
