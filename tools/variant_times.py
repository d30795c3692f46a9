"""Developer aid (GPU box): kernel time of the scale variants that stress the team master's narrow schedule (no oracle run).
python tools/variant_times.py"""
import os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests"))
import ecneproject_amd as E, multi_copy, ecdsa_like, fixtures
from gpu_common import build_system
EDDSA = "ecne_circomlib_tests/EdDSAMiMCSpongeVerifier@eddsamimcsponge.r1cs"
for name, p, tr in (("3 x Sponge", multi_copy.cached(EDDSA, 3), False), ("45 x Sponge", multi_copy.cached(EDDSA, 45), False), ("ecdsa_like(26)", ecdsa_like.cached(26, 10), True)):
    s = build_system(None, ["secp256k1.r1cs"] if tr else [], ["Secp256k1AddUnequal"] if tr else [], path=p)
    ms = [E.solve_batch([s], fetch_states=False)[0].summary.device_ms for _ in range(4)]
    print(name, "kernel_ms %.3f" % min(ms))
