"""Developer aid (GPU box): kernel time of the scale variants that stress the team master's narrow schedule (no oracle run), as one
system and -- from the second solve on, ecne_set_split's default -- as independent parts.
python tools/variant_times.py"""
import os, sys, time
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests"))
import ecneproject_amd as E, multi_copy, ecdsa_like, fixtures
from gpu_common import build_system
EDDSA = "ecne_circomlib_tests/EdDSAMiMCSpongeVerifier@eddsamimcsponge.r1cs"
POS = "ecne_circomlib_tests/Poseidon@poseidon.r1cs"
for name, p, tr in (("3 x Sponge", multi_copy.cached(EDDSA, 3), False), ("45 x Sponge", multi_copy.cached(EDDSA, 45), False),
                    ("1400 x Poseidon", multi_copy.cached(POS, 1400), False), ("ecdsa_like(26)", ecdsa_like.cached(26, 10), True)):
    s = build_system(None, ["secp256k1.r1cs"] if tr else [], ["Secp256k1AddUnequal"] if tr else [], path=p)
    ms, dg, wall = [], [], []
    for _ in range(5):
        t0 = time.perf_counter()
        r = E.solve_batch([s], fetch_states="digest")[0]
        wall.append((time.perf_counter() - t0) * 1e3)
        ms.append(r.summary.device_ms); dg.append(r.digest)
    info = s.split_info()
    print(name, "rows", len(s), "kernel_ms", [round(x, 3) for x in ms], "wall_ms", [round(x, 1) for x in wall], "parts %d groups %d plan_ms %.1f" % info[:3],
          "digests equal", len(set(dg)) == 1, "pops", r.summary.pops, "outer", r.summary.outer_iterations)
