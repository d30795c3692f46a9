"""Developer aid (GPU box): bit-exact comparison of the HIP engine with the sequential oracle on a
scale-out ecdsa_like(S) (state of every variable, counts, pops, per-rule hits).  The oracle needs
~25 s for S = 26 and ~7 min for S = 104 on one core.   python tests/tools/full_parity_scale.py [S]"""
import os, sys, time
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE))); sys.path.insert(0, os.path.dirname(HERE))
import ecneproject_amd as E, ecdsa_like, fixtures, orc
from gpu_common import assert_bit_exact, build_system

S = int(sys.argv[1]) if len(sys.argv) > 1 else 26
path = ecdsa_like.cached(S, 10)
s = build_system(None, ["secp256k1.r1cs"], ["Secp256k1AddUnequal"], path=path)
t = time.time(); g = E.solve_batch([s])[0]; tg = time.time() - t
t = time.time(); o = orc.run(path, [fixtures.path("secp256k1.r1cs")], ["Secp256k1AddUnequal"]); to = time.time() - t
assert_bit_exact("ecdsa_like(%d,10)" % S, g, o)
print({"S": S, "rows_main": s.info.n_rows_main, "rows": len(s), "bit_exact": True, "verdict": g.function_good,
       "pops": int(g.summary.pops), "outer_iterations": int(g.summary.outer_iterations),
       "gpu_first_solve_s": round(tg, 3), "gpu_kernel_ms": round(g.summary.device_ms, 2),
       "oracle_total_s": round(to, 1), "oracle_solve_s": round(o.summary.t_solve, 1)})
