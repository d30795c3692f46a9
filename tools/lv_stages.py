"""Developer aid (GPU box, library built with ECNE_BUILD_FLAGS=-DECNE_LVPROF): stage clocks of the level rounds (level.hip.hpp) and of the crew
rounds (crew.hip.hpp), which share the slots -- a crew round: record + descriptor | flag bytes + inline fan-out lists | decisions | marks, read sets,
barrier, prefix ("marks + check") | commit + barrier | push resolution by wavefront 0 + barrier; "fan-out lists" is the level rounds' own stage.
python tools/lv_stages.py <fixture relpath> ..."""
import os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests"))
import ecneproject_amd as E, fixtures
from gpu_common import build_system
names = ["entry", "record + descriptor", "flag bytes", "decisions", "marks + check", "fan-out lists", "commit", "push resolution"]
for rel in sys.argv[1:]:
    secp = rel == "secp"
    if rel.startswith("ecdsa:"):      # ecdsa_like(S) + Secp256k1AddUnequal: the master of a team (state in device memory)
        import ecdsa_like
        s = build_system(None, ["secp256k1.r1cs"], ["Secp256k1AddUnequal"], path=ecdsa_like.cached(int(rel.split(":")[1]), 10))
    else:
        s = build_system("secp256k1.r1cs", ["bigmultmodp.r1cs", "biglessthan.r1cs"], ["BigMultModP", "BigLessThan"]) if secp else build_system(rel)
    for _ in range(3): r = E.solve_batch([s], secp_solve=secp, fetch_states=False)[0]
    sm = r.summary
    sd = list(sm.sched)
    n = sm.rule_hits[13]
    print("%s dev_ms %.3f pops %d rounds %d" % (rel, sm.device_ms, sm.pops, n))
    for k, nm in enumerate(names):
        print("  %-24s %8.3f ms  %6.2f us/round" % (nm, sd[k] * 1e-5, sd[k] * 1e-2 / max(n, 1)))
    print("  %-24s %8.3f ms" % ("exit", sd[15] * 1e-5))
    print("  declines by reason (0 no record/long, 1 other shape, 2 error shape, 3 x == y limbs, 4/5 R7/R8 in reach, 6 long row list):", sd[8:15])
