"""Developer aid (GPU box): the bench solve on forced team sizes.   python tools/nwg_sweep.py 170 200 226 248"""
import os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests"))
import ecneproject_amd as E, fixtures, ecdsa_like
E.warmup(0)
S = 26
s = E.System(E.R1CS(ecdsa_like.cached(S, 10))); s.abstract(E.R1CS(fixtures.path("secp256k1.r1cs")), "Secp256k1AddUnequal")
for n in [int(a) for a in sys.argv[1:]] or [0]:
    ms = []
    for _ in range(5):
        r = E.solve_batch([s], fetch_states=False, force_nwg=n)[0]
        ms.append(r.summary.device_ms)
    sm = r.summary
    print("nwg %3d: kernel ms best %.3f median %.3f | phases %s | multi %s | team %s" % (n, min(ms), sorted(ms)[2], [round(x, 2) for x in sm.phase_ms[:6]], [round(x, 2) for x in sm.multi_ms[:6]], list(sm.team)))
