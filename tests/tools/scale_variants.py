"""Developer aid (GPU box): is the schedule policy (rounds.hip.hpp: 15 thresholds measured on ecdsa_like(26)) fitted to one
generator? Million-row inputs of other shapes, each solved through the device front-end, compared with the sequential oracle on
the WHOLE state (bit-exact or the script fails) and timed; the round mix (fast wavefront / multi-workgroup / general) is printed
beside the constraints/s.     python tests/tools/scale_variants.py [quick]
  A  ecdsa_like(26, stride 10)            Multiplexer(3, 1024): 1 025-term sums, the bench workload (for reference)
  B  ecdsa_like(64, stride 8)             Multiplexer(3, 256): 257-term sums, 63 adders, 2.5x the strides
  C  45 x EdDSAMiMCSpongeVerifier         45 independent chains of ~28 000 sequential pops each, no long rows, no trusted function
  D  1400 x Poseidon                      1 400 short chains (~400 levels), products and sums only
"""
import os, sys, time, json
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE))); sys.path.insert(0, os.path.dirname(HERE))
import ecneproject_amd as E, ecdsa_like, fixtures, multi_copy, orc
from gpu_common import assert_bit_exact, build_system

quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
EDDSA = "ecne_circomlib_tests/EdDSAMiMCSpongeVerifier@eddsamimcsponge.r1cs"
POS = "ecne_circomlib_tests/Poseidon@poseidon.r1cs"
cases = [("A ecdsa_like(26,10)", lambda: ecdsa_like.cached(26, 10), True),
         ("B ecdsa_like(64,8)", lambda: ecdsa_like.cached(64, 8), True),
         ("C 45 x EdDSAMiMCSponge", lambda: multi_copy.cached(EDDSA, 45), False),
         ("D 1400 x Poseidon", lambda: multi_copy.cached(POS, 1400), False)]
if quick:
    cases = [("B' ecdsa_like(6,8)", lambda: ecdsa_like.cached(6, 8), True), ("C' 3 x EdDSAMiMCSponge", lambda: multi_copy.cached(EDDSA, 3), False),
             ("D' 30 x Poseidon", lambda: multi_copy.cached(POS, 30), False)]
E.solve_batch([E.System(E.R1CS(fixtures.path("target/division.r1cs")))])
for name, mk, trusted in cases:
    p = mk()
    tr, nm = (["secp256k1.r1cs"], ["Secp256k1AddUnequal"]) if trusted else ([], [])
    t0 = time.perf_counter()
    s = build_system(None, tr, nm, path=p)
    g = E.solve_batch([s])[0]
    t_first = time.perf_counter() - t0
    ms = []
    for _ in range(3):     # (from the second solve on a file of independent circuits runs as parts: ecne_set_split, include/ecne.h)
        r = E.solve_batch([s], fetch_states=False)[0]
        ms.append(r.summary.device_ms)
    parts, groups, plan_ms, _tried = s.split_info()
    g_parts = E.solve_batch([s])[0] if parts else None
    t0 = time.perf_counter()
    o = orc.run(p, [fixtures.path(t) for t in tr], nm)
    t_or = time.perf_counter() - t0
    assert_bit_exact(name, g, o)
    if g_parts is not None:
        assert_bit_exact(name + " as parts", g_parts, o)
    sm = g.summary
    sd = list(sm.sched)
    n_multi = int(sm.rule_hits[14]) >> 16
    print(json.dumps({"case": name, "rows_main": int(s.info.n_rows_main), "rows": len(s), "bit_exact": True, "verdict": bool(g.function_good),
                      "kernel_ms_first_solve": round(float(sm.device_ms), 3), "parts": parts, "groups": groups, "plan_ms": round(plan_ms, 1),
                      "kernel_ms": round(min(ms), 3), "constraints_per_s": round(int(s.info.n_rows_main) / (min(ms) * 1e-3)),
                      "file_to_verdict_ms": round(t_first * 1e3, 1), "pops": int(sm.pops), "outer_iterations": int(sm.outer_iterations),
                      "rounds": int(sm.rule_hits[13]), "multi_workgroup_rounds": n_multi, "multi_ms": round(float(sm.queue_ms[7]), 2),
                      "multi_committing_lt64_lt4096_more": sd[13:16], "fast_wave_rounds": sd[0], "rows_in_fast_rounds": sd[1],
                      "fast_ms": round(sd[2] * 1e-5, 2), "general_wave_rounds": sd[3], "phase_ms": [round(x, 2) for x in list(sm.phase_ms)[:6]],
                      "oracle_solve_s": round(o.summary.t_solve, 2), "oracle_total_s": round(t_or, 1),
                      "speedup_vs_1_core": round(o.summary.t_solve / (min(ms) * 1e-3), 1)}))
