"""Developer aid (GPU box): the same small ecdsa_like systems solved over and over on 1, 3 and 16 workgroups, every result compared with
the oracle -- a soak for rare schedule races.   python tests/tools/repeat_solves.py [iterations]"""
import sys, os
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE))); sys.path.insert(0, os.path.dirname(HERE))
import numpy as np
import ecneproject_amd as E, ecdsa_like, fixtures, orc
from gpu_common import build_system
cases = []
for S_, st in ((3, 7), (4, 8), (3, 9), (2, 10)):
    path = ecdsa_like.cached(S_, st)
    s = build_system(None, ["secp256k1.r1cs"], ["Secp256k1AddUnequal"], path=path)
    o = orc.run(path, [fixtures.path("secp256k1.r1cs")], ["Secp256k1AddUnequal"])
    cases.append((S_, st, s, o))
nfail = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    for S_, st, s, o in cases:
        for nwg in (0, 3, 16):
            g = E.solve_batch([s], force_nwg=nwg)[0]
            bad = np.flatnonzero((g.lb != o.lb).any(axis=1) | (g.ub != o.ub).any(axis=1) | (g.flags != o.flags))
            if len(bad) or g.summary.pops != o.summary.pops or g.status != o.status:
                nfail += 1
                print("FAIL it", it, "S", S_, st, "nwg", nwg, "status", g.status, "diff vars", bad[:8], len(bad), "pops", g.summary.pops, o.summary.pops, "solo", g.summary.sched[3], "drains", round(g.summary.multi_ms[7]*1e5))
                for v in bad[:2]:
                    print("   var", v, "gpu lb", g.lb[v], "ub", g.ub[v], "flags", g.flags[v], "| oracle lb", o.lb[v], "ub", o.ub[v], "flags", o.flags[v])
print("failures", nfail)
