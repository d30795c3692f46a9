"""Checker (GPU box): the chained circuits that run on crew rounds (crew.hip.hpp) solved N times each, alone and all together as one
batch launch, every result compared with the oracle's state and counters -- a soak for rare cross-wavefront races in the crew's LDS
protocol (marks by round parity, read sets, candidate lists).   python tests/tools/soak_crew.py [iterations]"""
import os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE))); sys.path.insert(0, os.path.dirname(HERE))
import ecneproject_amd as E, fixtures, orc
from gpu_common import build_system, assert_bit_exact

CASES = [("ecne_circomlib_tests/Poseidon@poseidon.r1cs", [], []),
         ("ecne_circomlib_tests/EdDSAMiMCSpongeVerifier@eddsamimcsponge.r1cs", [], []),
         ("ecne_circomlib_tests/EdDSAPoseidonVerifier@eddsaposeidon.r1cs", [], []),
         ("ecne_circomlib_tests/EdDSAMiMCVerifier@eddsamimc.r1cs", [], []),
         ("ecne_circomlib_tests/BabyPbk@babyjub.r1cs", [], []),
         ("ecne_circomlib_tests/Pedersen@pedersen_old.r1cs", [], []),
         ("ecne_circomlib_tests/MiMCSponge@mimcsponge.r1cs", [], []),
         ("secp256k1.r1cs", ["bigmultmodp.r1cs", "biglessthan.r1cs"], ["BigMultModP", "BigLessThan"]),
         ("tornadocash_circuits/withdraw.r1cs", fixtures.PED, fixtures.PED_NAMES)]
N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
systems, oracles = [], []
for rel, tr, nm in CASES:
    systems.append(build_system(rel, tr, nm))
    oracles.append(orc.run(fixtures.path(rel), [fixtures.path(t) for t in tr], nm, True))
nfail = 0
for it in range(N):
    res = [E.solve_batch([s], secp_solve=True)[0] for s in systems] if it % 2 == 0 else E.solve_batch(systems, secp_solve=True)
    for (rel, _, _), g, o in zip(CASES, res, oracles):
        try:
            assert_bit_exact(rel, g, o)
        except AssertionError as e:
            nfail += 1
            print("FAIL it", it, rel, str(e)[:300])
print("soak_crew: %d iterations x %d circuits, failures %d" % (N, len(CASES), nfail))
