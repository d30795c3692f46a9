"""Developer aid (GPU box, library built with ECNE_BUILD_FLAGS=-DECNE_W2PROF): stage clocks of the fast wavefront round on the bench solve."""
import os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests"))
import ecneproject_amd as E, fixtures, ecdsa_like
p = ecdsa_like.cached(int(sys.argv[1]) if len(sys.argv) > 1 else 26, 10)
s = E.System(E.R1CS(p)); s.abstract(E.R1CS(fixtures.path("secp256k1.r1cs")), "Secp256k1AddUnequal")
for _ in range(3): r = E.solve_batch([s], fetch_states=False)[0]
sm = r.summary
n = sm.sched[0]
names = ["queue + record + descriptor", "flag bytes", "decisions (+ long row scan)", "marks + check", "fan-out lists + candidate scan", "commit", "push resolution", "wipe + final fence"]
print("dev_ms %.3f fast rounds %d rows %d total %.3f ms" % (sm.device_ms, n, sm.sched[1], sm.sched[2] * 1e-5))
for k, nm in enumerate(names):
    print("  %-34s %7.3f ms  %6.2f us/round" % (nm, sm.sched[8 + k] * 1e-5, sm.sched[8 + k] * 1e-2 / max(n, 1)))
