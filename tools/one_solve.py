"""Developer aid (GPU box): N solves of one fixture in one mode, for profiling.  python tools/one_solve.py <rel> <mode> [n]"""
import os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests"))
import ecneproject_amd as E, fixtures
s = E.System(E.R1CS(fixtures.path(sys.argv[1])))
for _ in range(int(sys.argv[3]) if len(sys.argv) > 3 else 3):
    r = E.solve_batch([s], fetch_states=False, queue_mode=int(sys.argv[2]))[0]
print(sys.argv[1], "mode", sys.argv[2], "pops", r.summary.pops, "dev_ms %.3f" % r.summary.device_ms)
