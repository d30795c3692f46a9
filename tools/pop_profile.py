"""Developer aid (GPU box, library built with ECNE_BUILD_FLAGS=-DECNE_POPPROF): stage times of strictly sequential pops."""
import os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests"))
import ecneproject_amd as E, fixtures
mode = 2 if "--chain" in sys.argv else 1
names2 = ["record arrival", "flags + ballots", "R1 (+requeue)", "shape rules (xy / R2)", "general executor fallback", "R7/R8 tail", "-", "loop top (incl. plain-sum tail, f1 continue)"]
names = ["queue+inq+fence", "rp x6 + solved", "rinfo+rp (exec_row entry)", "colA/B + flags", "colC + flags + abz", "mark + requeue", "R2 + rest", "loop top"]
for rel in [a for a in sys.argv[1:] if not a.startswith('--')]:
    s = E.System(E.R1CS(fixtures.path(rel)))
    for _ in range(2):
        r = E.solve_batch([s], fetch_states=False, queue_mode=mode)[0]
    sm = r.summary
    print(rel, "pops", sm.pops, "dev_ms %.3f" % sm.device_ms, "R1 hits", sm.rule_hits[0])
    print('   phases', [round(x, 3) for x in sm.phase_ms[:6]])
    for n, t in zip(names2 if mode == 2 else names, sm.queue_ms):
        print("   %-28s %8.3f ms  %7.1f ns/pop" % (n, t, 1e6 * t / max(sm.pops, 1)))
