"""Checker / report (GPU box): every circomlib file and the trusted-function configurations, solved alone on the GPU and by the
sequential oracle on one core: device time, oracle solve time, ratio -- and bit-exactness of the result while at it.
python tests/tools/per_file_vs_oracle.py [reps]"""
import os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE))); sys.path.insert(0, os.path.dirname(HERE))
import ecneproject_amd as E, fixtures, orc
from gpu_common import assert_bit_exact, build_system

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
E.solve_batch([build_system("target/division.r1cs")])          # HIP runtime start-up
cases = [(r, (), (), False) for r in fixtures.circomlib_suite()]
cases += [(c[0], c[1], c[2], c[0].startswith("secp")) for c in fixtures.REFERENCE_ASSERTED if c[1]]
print("%-64s %7s %8s %9s %10s %7s" % ("file (+ trusted functions)", "rows", "pops", "gpu_ms", "oracle_ms", "ratio"))
tg = to = 0.0
slower = []
for rel, tr, nm, secp in cases:
    s = build_system(rel, tr, nm)
    best = None
    for _ in range(reps):
        g = E.solve_batch([s], secp_solve=secp)[0]
        best = g if best is None or g.summary.device_ms < best.summary.device_ms else best
    o = min((orc.run(fixtures.path(rel), [fixtures.path(t) for t in tr], nm, secp) for _ in range(reps)), key=lambda x: x.summary.t_solve)
    assert_bit_exact(rel, best, o)
    gms, oms = best.summary.device_ms, o.summary.t_solve * 1e3
    tg += gms; to += oms
    if gms > oms:
        slower.append(rel)
    print("%-64s %7d %8d %9.3f %10.3f %7.2f" % ((rel + (" +T" if tr else ""))[-64:], len(s), best.summary.pops, gms, oms, oms / max(gms, 1e-9)))
print("sum: gpu %.1f ms, oracle %.1f ms; GPU slower than one CPU core on %d of %d (small files: a launch costs ~0.1 ms)" % (tg, to, len(slower), len(cases)))
