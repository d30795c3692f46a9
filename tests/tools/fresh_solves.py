"""Developer aid (GPU box): FRESH systems (parse + abstraction + upload + classification + first solve) of the small ecdsa_like cases, over
and over, every result compared with the oracle -- the soak for whatever only a first solve does (tests/tools/repeat_solves.py re-solves
resident systems).   python tests/tools/fresh_solves.py [iterations]"""
import sys, os
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE))); sys.path.insert(0, os.path.dirname(HERE))
import numpy as np
import ecneproject_amd as E, ecdsa_like, fixtures, orc
from gpu_common import build_system
cases = []
for S_, st in ((3, 7), (4, 8), (3, 9), (2, 2), (5, 6)):
    path = ecdsa_like.cached(S_, st)
    o = orc.run(path, [fixtures.path("secp256k1.r1cs")], ["Secp256k1AddUnequal"])
    cases.append((S_, st, path, o))
nfail = n = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 100):
    for fe in (E.FRONTEND_HOST, E.FRONTEND_DEVICE):
        E.set_frontend(fe)
        for S_, st, path, o in cases:
            for nwg in ((0, 3) if it % 2 else (0,)):
                s = build_system(None, ["secp256k1.r1cs"], ["Secp256k1AddUnequal"], path=path)
                g = E.solve_batch([s], force_nwg=nwg)[0]
                n += 1
                bad = np.flatnonzero((g.lb != o.lb).any(axis=1) | (g.ub != o.ub).any(axis=1) | (g.flags != o.flags))
                if len(bad) or g.summary.pops != o.summary.pops or g.status != o.status:
                    nfail += 1
                    print("FAIL it", it, "fe", fe, "S", S_, st, "nwg", nwg, "status", g.status, "diff vars", bad[:8], len(bad), "pops", g.summary.pops, o.summary.pops, flush=True)
                    for v in bad[:3]:
                        print("   var", v + 1, "gpu lb", g.lb[v], "ub", g.ub[v], "flags", g.flags[v], "| oracle lb", o.lb[v], "ub", o.ub[v], "flags", o.flags[v])
                    g2 = E.solve_batch([s], force_nwg=nwg)[0]
                    bad2 = np.flatnonzero((g2.lb != o.lb).any(axis=1) | (g2.ub != o.ub).any(axis=1) | (g2.flags != o.flags))
                    print("   solved again: diff vars", len(bad2))
                del s, g
print("fresh solves", n, "failures", nfail)
