"""Developer aid (GPU box, -DECNE_JITTER library): one fuzz system solved again and again on a forced team under changing jitter seeds until a
solve differs from the oracle (an ECNE_ETIMEOUT is a barrier that did not complete: the device prints which one).
    ECNE_LIB=libecne_hip_jitter.so [ECNE_DRAIN=2] python tests/tools/jitter_repro.py <seed> <scale> <nwg> [tries] [neighbours]"""
import os, sys, tempfile
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE))); sys.path.insert(0, os.path.dirname(HERE))
import ecneproject_amd as E, fuzz_r1cs, orc

seed, scale, nwg = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
tries = int(sys.argv[4]) if len(sys.argv) > 4 else 200
nb = int(sys.argv[5]) if len(sys.argv) > 5 else 0
d = tempfile.mkdtemp(prefix="ecne_jit_")
paths = []
for s in range(seed - nb, seed + nb + 1):
    p = os.path.join(d, "%d.r1cs" % s)
    fuzz_r1cs.write(p, fuzz_r1cs.make_decomp(s) if scale == 0 else fuzz_r1cs.make_wide(s, scale) if (s % 3 or scale > 1) else fuzz_r1cs.make(s))
    paths.append(p)
oracles = [orc.run(p, want_states=False) for p in paths]
systems = [E.System(E.R1CS(p)) for p in paths]
bad = 0
for t in range(tries):
    res = E.solve_batch(systems, force_nwg=nwg, fetch_states=False)
    for p, g, o in zip(paths, res, oracles):
        if (g.status, g.summary.pops) != (o.status, o.summary.pops):
            bad += 1
            print("try", t, os.path.basename(p), "gpu status", g.status, "pops", g.summary.pops, "| oracle", o.status, o.summary.pops, flush=True)
print("jitter_repro: seed %d scale %d nwg %d: %d tries x %d systems, %d differing" % (seed, scale, nwg, tries, len(paths), bad))
