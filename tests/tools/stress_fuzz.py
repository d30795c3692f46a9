"""Developer aid (GPU box): one-off randomized stress beyond the seeds the test suite pins.
python tests/tools/stress_fuzz.py [first_seed] [count] [scale]   -- wide systems with long rows (scale > 1: thousands of
rows, long rows up to 1 100 terms), several workgroup counts."""
import os, sys, tempfile
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE))); sys.path.insert(0, os.path.dirname(HERE))
import ecneproject_amd as E, fuzz_r1cs, orc
from gpu_common import assert_bit_exact

first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 500
scale = int(sys.argv[3]) if len(sys.argv) > 3 else 1
d = tempfile.mkdtemp(prefix="ecne_stress_")
paths = []
for seed in range(first, first + count):
    p = os.path.join(d, "%d.r1cs" % seed)
    fuzz_r1cs.write(p, fuzz_r1cs.make_wide(seed, scale) if (seed % 3 or scale > 1) else fuzz_r1cs.make(seed))
    paths.append(p)
oracles = [orc.run(p) for p in paths]
systems = [E.System(E.R1CS(p)) for p in paths]
for nwg in (0, 2, 5, 8, 24):
    res = []
    for i in range(0, len(systems), 100):
        res += E.solve_batch(systems[i:i + 100], force_nwg=nwg)
    for p, g, o in zip(paths, res, oracles):
        assert_bit_exact("%s nwg=%d" % (os.path.basename(p), nwg), g, o)
    print("nwg", nwg, "ok:", len(res), "systems,", sum(o.status != 0 for o in oracles), "with error status")
