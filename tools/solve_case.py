"""Developer aid (GPU box): one reference configuration (fixture + trusted functions), timing + schedule diagnostics.
python tools/solve_case.py secp|withdraw|commit|<fixture relpath> [mode ...]"""
import os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests"))
import ecneproject_amd as E, fixtures
CASES = {"secp": ("secp256k1.r1cs", ["bigmultmodp.r1cs", "biglessthan.r1cs"], ["BigMultModP", "BigLessThan"]),
         "withdraw": ("tornadocash_circuits/withdraw.r1cs", fixtures.PED, fixtures.PED_NAMES),
         "commit": ("tornadocash_circuits/commitHasher.r1cs", fixtures.PED, fixtures.PED_NAMES)}
rel, tr, nm = CASES.get(sys.argv[1], (sys.argv[1], [], []))
if sys.argv[1].startswith("ecdsa"):          # ecdsa[:S]  -> the bench workload, ecdsa_like(S) + trusted secp256k1.r1cs
    import ecdsa_like
    S_ = int(sys.argv[1].split(":")[1]) if ":" in sys.argv[1] else 26
    E.set_host_threads(16)
    rel, tr, nm = ecdsa_like.cached(S_, 10, directory="/tmp/ecne_bench_%d" % os.getuid()), ["secp256k1.r1cs"], ["Secp256k1AddUnequal"]
fl = sorted([(n, E.R1CS(fixtures.path(t))) for t, n in zip(tr, nm)], key=lambda x: -len(x[1]))
s = E.System(E.R1CS(rel if os.path.isabs(rel) else fixtures.path(rel)))
for n, f in fl:
    s.abstract(f, n)
for mode in ([int(m) for m in sys.argv[2:]] or [0]):
    best = None
    for rep in range(3):
        r = E.solve_batch([s], secp_solve=True, fetch_states=False, queue_mode=mode, force_nwg=int(os.environ.get('FORCE_NWG', '0')))[0]
        if best is None or r.summary.device_ms < best.summary.device_ms:
            best = r
    sm = best.summary
    print(rel, "mode", mode, "rows", len(s), "status", best.status, "dev_ms %.3f" % sm.device_ms, "pops", sm.pops, "outer", sm.outer_iterations,
          "rounds", sm.rule_hits[13], "alone", sm.rule_hits[14] & 0xFFFF, "\n   phases[setup,P1+P2+queue,P3,P4,P5,verdict]", [round(x, 3) for x in sm.phase_ms[:6]], "P3 passes", int(sm.phase_ms[6]), "P1+P2 alone %.3f" % sm.phase_ms[7],
          "\n   queue[head,mark,check,exec,flatten,resolve,alone+bursts+wave,multi]", [round(x, 3) for x in sm.queue_ms[:8]], "\n   hits", list(sm.rule_hits[:13]))
    mm = list(sm.multi_ms)
    print("   multi[mark,check,exec+scan,expand,count+scan,write]", [round(x, 3) for x in mm[:6]], "drain levels %d rounds %d" % (round(mm[6] * 1e5), round(mm[7] * 1e5)))
    sd = list(sm.sched)
    print("   fast rounds %d rows %d ms %.3f | general wave rounds %d rows %d ms %.3f | declines[norec/big, other shape, R2 err, xy slow, xy R7/R8, sum R7/R8] %s solo pops (long row lists) %d | multi rounds by rows [<64, <4096, more] %s"
          % (sd[0], sd[1], sd[2] * 1e-5, sd[3], sd[4], sd[5] * 1e-5, sd[6:12], sd[12], sd[13:16]))
