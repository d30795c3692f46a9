import os, sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import ecneproject_amd as E, multi_copy, fixtures
E.warmup(0)
for rel, n in (("ecne_circomlib_tests/EdDSAMiMCSpongeVerifier@eddsamimcsponge.r1cs", 45), ("ecne_circomlib_tests/Poseidon@poseidon.r1cs", 1400)):
    p = multi_copy.cached(rel, n)
    t = time.perf_counter(); s = E.System(E.R1CS(p)); r = E.solve_batch([s], fetch_states=False)[0]; t1 = time.perf_counter() - t
    t = time.perf_counter(); r2 = E.solve_batch([s], fetch_states=False)[0]; t2 = time.perf_counter() - t
    r3 = E.solve_batch([s], fetch_states=False)[0]
    print(n, "first %.1f ms (kernel %.1f) second %.1f ms (kernel %.1f) third kernel %.1f" % (t1 * 1e3, r.summary.device_ms, t2 * 1e3, r2.summary.device_ms, r3.summary.device_ms), s.split_info())
