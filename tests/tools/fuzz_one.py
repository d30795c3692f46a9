"""Developer aid (GPU box): ONE stress-fuzz system (tests/tools/stress_fuzz.py's generator) on several workgroup counts and schedules, compared
with the oracle in detail.   python tests/tools/fuzz_one.py <seed> [scale]      (AB_PKG=<dir>: the library of another build of the package)"""
import os, sys, tempfile
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.environ.get("AB_PKG", os.path.dirname(os.path.dirname(HERE)))); sys.path.insert(0, os.path.dirname(HERE))
import numpy as np
import ecneproject_amd as E, fuzz_r1cs, orc
seed = int(sys.argv[1]); scale = int(sys.argv[2]) if len(sys.argv) > 2 else 1
p = os.path.join(tempfile.mkdtemp(prefix="ecne_one_"), "%d.r1cs" % seed)
fuzz_r1cs.write(p, fuzz_r1cs.make_wide(seed, scale) if (seed % 3 or scale > 1) else fuzz_r1cs.make(seed))
o = orc.run(p)
s = E.System(E.R1CS(p))
print("seed", seed, "rows", len(s), "oracle status", o.status, "verdict", o.verdict, "pops", o.summary.pops, "outer", o.summary.outer_iterations, "hits", list(o.summary.rule_hits[:13]))
for mode in (0, 3, 4, 1):
    for nwg in (0, 2, 5, 8, 24):
        for rep in range(3):
            g = E.solve_batch([s], force_nwg=nwg, queue_mode=mode)[0]
            bad = np.flatnonzero((g.lb != o.lb).any(axis=1) | (g.ub != o.ub).any(axis=1) | (g.flags != o.flags))
            ok = len(bad) == 0 and g.status == o.status and g.function_good == o.verdict and g.summary.pops == o.summary.pops
            if not ok or rep == 0:
                print("mode", mode, "nwg", nwg, "rep", rep, "OK" if ok else "FAIL", "status", g.status, "good", g.function_good, "pops", g.summary.pops, "outer", g.summary.outer_iterations, "diff vars", len(bad), bad[:6] + 1,
                      "hits", list(g.summary.rule_hits[:13]))
                for v in bad[:3]:
                    print("     var", v + 1, "gpu", g.flags[v], g.lb[v], g.ub[v], "| oracle", o.flags[v], o.lb[v], o.ub[v])
