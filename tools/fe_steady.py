"""Developer aid (GPU box): the device front-end in steady state -- ecdsa_like(S): parse, abstraction, layout + first solve, three
times in one process (the first trip pays the runtime's one-time costs) -- for rocprofv3 kernel traces / PMC passes of the
front-end kernels.   python tools/fe_steady.py [S]"""
import os, sys, time
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests"))
import ecneproject_amd as E, fixtures, ecdsa_like
S = int(sys.argv[1]) if len(sys.argv) > 1 else 26
p = ecdsa_like.cached(S, 10)
E.solve_batch([E.System(E.R1CS(fixtures.path("target/division.r1cs")))])
tr = E.R1CS(fixtures.path("secp256k1.r1cs"))
for i in range(3):
    t = [time.perf_counter()]
    m = E.R1CS(p); t.append(time.perf_counter()); st_parse = E.frontend_stats()
    s = E.System(m); s.abstract(tr, "Secp256k1AddUnequal"); t.append(time.perf_counter())
    s.info; t.append(time.perf_counter()); st = E.frontend_stats()
    r = E.solve_batch([s], fetch_states=False)[0]; t.append(time.perf_counter())
    print("trip %d: parse %.2f ms (upload %.2f, offsets %.2f, fill %.2f) | abstraction %.2f ms (prep %.2f, fp %.3f, scan %.3f, verify %.2f, compact %.2f) | layout %.2f ms | "
          "upload + classify + solve %.2f ms (kernel %.2f) | file -> verdict %.1f ms, verdict %s, %d -> %d rows" %
          (i, (t[1] - t[0]) * 1e3, st_parse["upload_ms"], st_parse["offsets_ms"], st_parse["fill_ms"], (t[2] - t[1]) * 1e3, st["prep_ms"], st["fingerprint_ms"], st["scan_ms"],
           st["verify_ms"], st["compact_ms"], (t[3] - t[2]) * 1e3, (t[4] - t[3]) * 1e3, r.summary.device_ms, (t[4] - t[0]) * 1e3, r.function_good, len(m), len(s)))
    del s, m
