"""Developer aid (GPU box): per-file timing of the circomlib suite (BASELINE.json config 4) and of the
trusted-function configurations, each solved alone and all together as one batch launch.
python tools/suite_stats.py [reps] [--seq]    (--seq adds the strictly sequential schedule, queue_mode=1)"""
import os, sys, time
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests"))
import ecneproject_amd as E, fixtures

reps = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 3
seq = "--seq" in sys.argv
seq_mode = 2 if "--chain" in sys.argv else 1
seq = seq or "--chain" in sys.argv


def build(rel, trusted=(), names=()):
    fl = sorted([(n, E.R1CS(fixtures.path(t))) for t, n in zip(trusted, names)], key=lambda x: -len(x[1]))
    s = E.System(E.R1CS(fixtures.path(rel)))
    for n, f in fl:
        s.abstract(f, n)
    return s


def best(systems, mode=0, secp=True):
    t = 1e9
    r = None
    for _ in range(reps):
        r = E.solve_batch(systems, secp_solve=secp, fetch_states=False, queue_mode=mode)
        t = min(t, max(x.summary.device_ms for x in r))
    return t, r


E.solve_batch([build("target/division.r1cs")])   # HIP runtime start-up
cases = [(r, (), ()) for r in fixtures.circomlib_suite()]
cases += [(c[0], c[1], c[2]) for c in fixtures.REFERENCE_ASSERTED if c[1]]
systems = []
print("%-62s %7s %7s %8s %6s %9s %8s %9s" % ("file", "rows", "vars", "pops", "outer", "dev_ms", "us/pop", "seq_ms" if seq else ""))
tot = 0.0
for rel, tr, nm in cases:
    s = build(rel, tr, nm)
    t, r = best([s])
    sm = r[0].summary
    ts = best([s], seq_mode)[0] if seq else 0.0
    tot += t
    print("%-62s %7d %7d %8d %6d %9.3f %8.2f %9.3f" % (rel[-62:], len(s), sm.n_vars, sm.pops, sm.outer_iterations, t, 1e3 * t / max(sm.pops, 1), ts))
    if not tr:
        systems.append(s)
tb, rb = best(systems)
t0 = time.perf_counter()
E.solve_batch(systems, fetch_states=False)
wall = (time.perf_counter() - t0) * 1e3
print("suite: %d files, sum of single solves %.2f ms, one batch launch %.2f ms (wall %.2f ms), pops %d" % (len(systems), tot, tb, wall, sum(x.summary.pops for x in rb)))
