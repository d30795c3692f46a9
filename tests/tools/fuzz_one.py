"""Developer aid (GPU box): one fuzz seed on the engine (default and sequential schedule) and on the oracle."""
import os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE))); sys.path.insert(0, os.path.dirname(HERE))
import ecneproject_amd as E, fuzz_r1cs, orc
seed = int(sys.argv[1]); wide = len(sys.argv) > 2 and sys.argv[2] == "wide"
p = "/tmp/fuzz_%d.r1cs" % seed
fuzz_r1cs.write(p, fuzz_r1cs.make_wide(seed) if wide else fuzz_r1cs.make(seed))
o = orc.run(p)
print("oracle  status", o.status, "pops", o.summary.pops, "outer", o.summary.outer_iterations, list(o.summary.rule_hits[:13]))
s = E.System(E.R1CS(p))
for mode in (0, 1):
    for nwg in (0, 2):
        g = E.solve_batch([s], queue_mode=mode, force_nwg=nwg)[0]
        sm = g.summary
        print("gpu mode", mode, "nwg", nwg, "status", g.status, "pops", sm.pops, "outer", sm.outer_iterations, list(sm.rule_hits[:16]))
