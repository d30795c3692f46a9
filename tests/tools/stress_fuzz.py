"""Developer aid (GPU box): one-off randomized stress beyond the seeds the test suite pins.
python tests/tools/stress_fuzz.py [first_seed] [count] [scale]   -- wide systems with long rows (scale > 1: thousands of
rows, long rows up to 1 100 terms; scale 0: systems of long binary decompositions), several workgroup counts."""
import os, sys, tempfile
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.environ.get("AB_PKG", os.path.dirname(os.path.dirname(HERE)))); sys.path.insert(0, os.path.dirname(HERE))      # (AB_PKG: the library of another build of the package)
import ecneproject_amd as E, fuzz_r1cs, orc
from gpu_common import assert_bit_exact

first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 500
scale = int(sys.argv[3]) if len(sys.argv) > 3 else 1
d = tempfile.mkdtemp(prefix="ecne_stress_")
paths = []
for seed in range(first, first + count):
    p = os.path.join(d, "%d.r1cs" % seed)
    # (scale 0: the long binary decompositions of make_decomp -- fastrow.hip.hpp's long_r4_idle / exec_long_r4 -- on fresh seeds)
    fuzz_r1cs.write(p, fuzz_r1cs.make_decomp(seed) if scale == 0 else fuzz_r1cs.make_wide(seed, scale) if (seed % 3 or scale > 1) else fuzz_r1cs.make(seed))
    paths.append(p)
oracles = [orc.run(p) for p in paths]
systems = [E.System(E.R1CS(p)) for p in paths]
for nwg in (0, 2, 5, 8, 24):
    res = []
    for i in range(0, len(systems), 100):
        res += E.solve_batch(systems[i:i + 100], force_nwg=nwg)
    nfail = 0
    for p, g, o in zip(paths, res, oracles):
        try:
            assert_bit_exact("%s nwg=%d" % (os.path.basename(p), nwg), g, o)
        except AssertionError as e:      # say what differs and go on: a rare failure is worth all the detail it can give
            nfail += 1
            sm, so = g.summary, o.summary
            print("FAIL", os.path.basename(p), "nwg", nwg, "| gpu status", g.status, "good", g.function_good, "counts", list(g.counts()), "pops", sm.pops, "outer", sm.outer_iterations, "hits", list(sm.rule_hits[:13]),
                  "| oracle status", o.status, "verdict", o.verdict, "pops", so.pops, "outer", so.outer_iterations, "hits", list(so.rule_hits[:13]), "|", str(e)[:200].replace("\n", " "), flush=True)
    print("nwg", nwg, "ok:" if not nfail else "FAILURES %d:" % nfail, len(res), "systems,", sum(o.status != 0 for o in oracles), "with error status")
    if nfail: sys.exit(1)
