"""Checker (GPU box): the same resident system solved N times must give the same state, counters and per-rule
hits every time, and the first of them must match the oracle (run once).  Catches schedule races that a single
parity run can miss.   python tests/tools/soak_determinism.py [S] [N]"""
import hashlib, os, sys, time
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE))); sys.path.insert(0, os.path.dirname(HERE))
import ecneproject_amd as E, fixtures, ecdsa_like, orc
from gpu_common import assert_bit_exact

S = int(sys.argv[1]) if len(sys.argv) > 1 else 26
N = int(sys.argv[2]) if len(sys.argv) > 2 else 200
p = ecdsa_like.cached(S, 10)
tp = fixtures.path("secp256k1.r1cs")
s = E.System(E.R1CS(p))
s.abstract(E.R1CS(tp), "Secp256k1AddUnequal")


def digest(r):
    h = hashlib.sha256()
    for a in (r.flags, r.lb, r.ub, r.abz, r.nvalues, r.values):
        h.update(a.tobytes())
    sm = r.summary
    h.update(repr((sm.status, sm.pops, sm.successful_steps, sm.outer_iterations, sm.num_unique, list(sm.rule_hits))).encode())
    return h.hexdigest()


first = E.solve_batch([s])[0]
t = time.time()
o = orc.run(p, [tp], ["Secp256k1AddUnequal"])
assert_bit_exact("soak first solve", first, o)
d0 = digest(first)
bad = 0
for i in range(1, N):
    r = E.solve_batch([s])[0]
    if digest(r) != d0:
        bad += 1
        print("solve %d differs" % i)
print({"S": S, "solves": N, "differing": bad, "oracle_s": round(time.time() - t, 1), "kernel_ms": round(first.summary.device_ms, 2)})
sys.exit(1 if bad else 0)
