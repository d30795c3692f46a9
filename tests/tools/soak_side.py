"""Checker (GPU box): batches that hold a multi-workgroup job NEXT TO single-workgroup ones -- the engine launches the team's kernel and, on a
stream of its own, k_solve for the single-workgroup jobs (ecne_engine.hip, side launch) -- solved again and again, every result compared
with the oracle's state and counters.   python tests/tools/soak_side.py [iterations]"""
import os, sys, tempfile
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE))); sys.path.insert(0, os.path.dirname(HERE))
import ecneproject_amd as E, fixtures, fuzz_r1cs, ecdsa_like, orc
from gpu_common import build_system, assert_bit_exact

N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
big = ecdsa_like.cached(6, 10)
cases = [(None, ["secp256k1.r1cs"], ["Secp256k1AddUnequal"], big),
         ("secp256k1.r1cs", ["bigmultmodp.r1cs", "biglessthan.r1cs"], ["BigMultModP", "BigLessThan"], None),
         ("ecne_circomlib_tests/Poseidon@poseidon.r1cs", [], [], None), ("target/division.r1cs", [], [], None),
         ("ecne_circomlib_tests/EdDSAPoseidonVerifier@eddsaposeidon.r1cs", [], [], None)]
d = tempfile.mkdtemp(prefix="ecne_side_")
for seed in range(12):
    p = os.path.join(d, "%d.r1cs" % seed)
    fuzz_r1cs.write(p, fuzz_r1cs.make_wide(7000 + seed, 1) if seed % 3 else fuzz_r1cs.make(7000 + seed))
    cases.append((None, [], [], p))
systems, oracles, tags = [], [], []
for rel, tr, nm, path in cases:
    systems.append(build_system(rel, tr, nm, path=path))
    oracles.append(orc.run(path or fixtures.path(rel), [fixtures.path(t) for t in tr], nm, True))
    tags.append(rel or os.path.basename(path))
nfail = 0
for it in range(N):
    order = list(range(len(systems)))
    if it % 2: order = order[1:] + order[:1]          # (the team job first / last in the batch)
    res = E.solve_batch([systems[k] for k in order], secp_solve=True)
    for k, g in zip(order, res):
        try:
            assert_bit_exact(tags[k], g, oracles[k])
        except AssertionError as e:
            nfail += 1
            print("FAIL it", it, tags[k], str(e)[:300])
print("soak_side: %d iterations x %d systems (the first one on a team of workgroups), failures %d" % (N, len(systems), nfail))
