"""Developer aid: solve one system on the GPU and print timing + schedule diagnostics.
python tools/solve_one.py ecdsa S stride | <fixture relpath>"""
import os, sys, time
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests"))
import ecneproject_amd as E, fixtures
if sys.argv[1] == "ecdsa":
    import ecdsa_like
    p = ecdsa_like.cached(int(sys.argv[2]), int(sys.argv[3]))
    s = E.System(E.R1CS(p)); s.abstract(E.R1CS(fixtures.path("secp256k1.r1cs")), "Secp256k1AddUnequal")
else:
    s = E.System(E.R1CS(fixtures.path(sys.argv[1])))
for mode in ([int(m) for m in sys.argv[4:]] if len(sys.argv) > 4 else [0]):
    for rep in range(2):
        r = E.solve_batch([s], fetch_states=False, queue_mode=mode, force_nwg=int(os.environ.get('ECNE_FORCE_NWG', '0')))[0]
    sm = r.summary
    print("mode", mode, "rows", len(s), "status", r.status, "good", r.function_good, "dev_ms %.2f" % sm.device_ms, "pops", sm.pops, "outer", sm.outer_iterations,
          "rounds", sm.rule_hits[13], "bigfb", sm.rule_hits[14] & 0xFFFF, "multi", sm.rule_hits[14] >> 16, "multi_rows", sm.rule_hits[15] >> 8, "candfb", sm.rule_hits[15] & 0xFF,
          "phases", [round(x, 2) for x in sm.phase_ms[:8]], "queue", [round(x, 2) for x in sm.queue_ms[:8]], "multi[mark,check,exec+scan,expand,compact+scan,final]", [round(x, 2) for x in sm.multi_ms[:6]], "\n   wave rounds fast: n %d rows %d ms %.2f; general: n %d rows %d ms %.2f; declined[norec/long, shape, err, bound3, xy r78, sum r78, bigfan]" % (sm.sched[0], sm.sched[1], sm.sched[2] * 1e-5, sm.sched[3], sm.sched[4], sm.sched[5] * 1e-5), list(sm.sched[6:13]), "multi rounds by rows committed [<64, <4096, more]", list(sm.sched[13:16]))
