"""Developer aid (GPU box): where a single-workgroup solve spends its queue phase -- fast wavefront rounds, general rounds, chain bursts.
python tools/sched_stats.py [file ...]   (default: Poseidon, EdDSAMiMCSponge, EdDSAPoseidon, secp256k1 + trusted functions)"""
import os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests"))
import ecneproject_amd as E, fixtures
from gpu_common import build_system

cases = [("ecne_circomlib_tests/Poseidon@poseidon.r1cs", (), (), False),
         ("ecne_circomlib_tests/EdDSAMiMCSpongeVerifier@eddsamimcsponge.r1cs", (), (), False),
         ("ecne_circomlib_tests/EdDSAPoseidonVerifier@eddsaposeidon.r1cs", (), (), False),
         ("ecne_circomlib_tests/BabyPbk@babyjub.r1cs", (), (), False),
         ("secp256k1.r1cs", ["bigmultmodp.r1cs", "biglessthan.r1cs"], ["BigMultModP", "BigLessThan"], True)]
if len(sys.argv) > 1:
    cases = [(a, (), (), False) for a in sys.argv[1:]]
E.solve_batch([build_system("target/division.r1cs")])
for rel, tr, nm, secp in cases:
    s = build_system(rel, tr, nm)
    best = None
    for _ in range(5):
        r = E.solve_batch([s], secp_solve=secp, fetch_states=False)[0]
        if best is None or r.summary.device_ms < best.summary.device_ms:
            best = r
    sm = best.summary
    sd = list(sm.sched)
    print("%s\n  rows %d pops %d outer %d device_ms %.3f | rounds %d | fast rounds %d rows %d (%.2f us each) | general/solo rounds %d rows %d (%.2f us each) | multi %d | alone %d" % (
        rel, len(s), sm.pops, sm.outer_iterations, sm.device_ms, sm.rule_hits[13], sd[0], sd[1], sd[2] * 1e-2 / max(sd[0], 1), sd[3], sd[4], sd[5] * 1e-2 / max(sd[3], 1),
        sm.rule_hits[14] >> 16, sm.rule_hits[14] & 0xFFFF))
    print("  phase_ms", [round(x, 3) for x in sm.phase_ms], "\n  queue_ms", [round(x, 3) for x in sm.queue_ms], "\n  why", sd[6:13])
