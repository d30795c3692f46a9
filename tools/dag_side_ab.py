"""Developer aid (GPU box): the DAG's side kernel -- secp256k1 (+ BigMultModP / BigLessThan, secp_solve) as a single-workgroup job of k_solve -- ALONE
and BESIDE k_solve_team (the config-5 DAG: ecdsa_like(S) on a team, the three small jobs on a stream of their own). rocprofv3's counter passes
serialise the two kernels, so counters cannot see the interference; the job's own in-kernel clocks can: phase_ms (setup, P1+P2+queue, P3, P4,
P5, verdict) and queue_ms (the queue phase by executor) of the secp256k1 job in both settings, median of N launches.
    python tools/dag_side_ab.py [S] [N]"""
import os, statistics, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests"))
import ecneproject_amd as E, fixtures, ecdsa_like
from ecneproject_amd import jobs as J

S = int(sys.argv[1]) if len(sys.argv) > 1 else 26
N = int(sys.argv[2]) if len(sys.argv) > 2 else 15
fx = fixtures.path
secp = J.Job(fx("secp256k1.r1cs"), "secp256k1", [(fx("bigmultmodp.r1cs"), "BigMultModP"), (fx("biglessthan.r1cs"), "BigLessThan")], True)
dag = [J.Job(ecdsa_like.cached(S, 10), "ecdsa_like", [(fx("secp256k1.r1cs"), "Secp256k1AddUnequal")]), secp,
       J.Job(fx("bigmultmodp.r1cs"), "bigmultmodp"), J.Job(fx("biglessthan.r1cs"), "biglessthan")]
PH = ["setup", "P1+P2+queue", "P3", "P4", "P5", "verdict"]
QU = ["head", "mark", "check", "exec", "flatten", "resolve", "alone/bursts/wave rounds", "multi rounds"]


def med(rows):
    return [statistics.median(c) for c in zip(*rows)]


def run(jl, pick, label, env=None):
    for k, v in (env or {}).items():
        os.environ[k] = v
    r = J.Runner(jl, 0, 1, 0, None)
    ph, qu, dev = [], [], []
    for i in range(N + 3):
        res, _ = r.run()
        if i < 3:
            continue
        s = res[pick].summary
        ph.append(list(s.phase_ms)[:6]); qu.append(list(s.queue_ms)[:8]); dev.append(max(float(x.summary.device_ms) for x in res))
    for k in (env or {}):
        del os.environ[k]
    p, q = med(ph), med(qu)
    print("%-44s launch %.3f ms | job clocks: %s = %.3f ms | queue: %s" % (label, statistics.median(dev), ", ".join("%s %.3f" % (n, v) for n, v in zip(PH, p)), sum(p),
                                                                        ", ".join("%s %.3f" % (n, v) for n, v in zip(QU, q) if v > 0.0005)), flush=True)
    return sum(p)


a = run([secp], 0, "secp256k1 alone (k_solve, 1 workgroup)")
b = run(dag, 1, "secp256k1 beside k_solve_team (DAG, side stream)")
c = run(dag, 1, "secp256k1 inside the team's launch (no side)", {"ECNE_SIDE_LAUNCH": "0"})
# ... and beside ~190 INDEPENDENT single-workgroup jobs of the same k_solve launch (no team, no barrier polling: a busy device and nothing else)
rels = [r for r in fixtures.circomlib_suite() if not any(k in r for k in ("EdDSA", "BabyPbk"))]
crowd = [secp] + [J.Job(fx(r), "%s#%d" % (r, k)) for k in range(3) for r in rels]
d = run(crowd, 0, "secp256k1 beside %d small jobs (k_solve only)" % (len(crowd) - 1))
print("beside the team / alone = %.3f, inside the team's launch / alone = %.3f, beside %d independent jobs / alone = %.3f" % (b / a, c / a, len(crowd) - 1, d / a))
