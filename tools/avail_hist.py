"""Developer aid (GPU box, library built with ECNE_BUILD_FLAGS="-DECNE_POPPROF -DECNE_AVAILHIST"): queue length seen by
each pop of the chain executor (queue_mode 2)."""
import os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests"))
import ecneproject_amd as E, fixtures
CASES = {"secp": ("secp256k1.r1cs", ["bigmultmodp.r1cs", "biglessthan.r1cs"], ["BigMultModP", "BigLessThan"]),
         "withdraw": ("tornadocash_circuits/withdraw.r1cs", fixtures.PED, fixtures.PED_NAMES)}
for a in sys.argv[1:]:
    rel, tr, nm = CASES.get(a, (a, [], []))
    fl = sorted([(n, E.R1CS(fixtures.path(t))) for t, n in zip(tr, nm)], key=lambda x: -len(x[1]))
    s = E.System(E.R1CS(fixtures.path(rel)))
    for n, f in fl:
        s.abstract(f, n)
    r = E.solve_batch([s], secp_solve=True, fetch_states=False, queue_mode=2)[0]
    h = [int(round(x * 1e5)) for x in r.summary.queue_ms]
    print(rel, "pops", r.summary.pops, dict(zip(["1", "2", "3-4", "5-8", "9-16", "17-64", "65-512", ">512"], h)))
