"""Developer aid (GPU box, library built with ECNE_BUILD_FLAGS="-DECNE_W2PROF [-DECNE_CHAIN_BURST_C=0]"): stage clocks of the fast
wavefront round on a single-workgroup (LDS-resident) solve.   python tools/w2_stages_file.py <fixture relpath>"""
import os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests"))
import ecneproject_amd as E, fixtures
from gpu_common import build_system
for rel in sys.argv[1:]:
    s = build_system(rel)
    for _ in range(3): r = E.solve_batch([s], fetch_states=False)[0]
    sm = r.summary
    n = sm.sched[0]
    names = ["queue + record + descriptor", "flag bytes", "decisions (+ long row scan)", "marks + check", "fan-out lists + candidate scan", "commit", "push resolution", "wipe + final fence"]
    print("%s dev_ms %.3f pops %d fast rounds %d rows %d total %.3f ms | bursts %d pops %d %.3f ms" % (rel, sm.device_ms, sm.pops, n, sm.sched[1], sm.sched[2] * 1e-5, sm.sched[3], sm.sched[4], sm.sched[5] * 1e-5))
    for k, nm in enumerate(names):
        print("  %-34s %7.3f ms  %6.2f us/round" % (nm, sm.sched[8 + k] * 1e-5, sm.sched[8 + k] * 1e-2 / max(n, 1)))
