"""Developer aid (GPU box): solve one .r1cs (absolute path, or poseidon:N / eddsa:N = N renumbered copies side by side) on several
workgroup counts and queue modes.   python tools/solve_file.py <path|poseidon:N|eddsa:N> [nwg,nwg,...] [mode,mode,...]"""
import os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests"))
import ecneproject_amd as E, fixtures
src = sys.argv[1]
if ":" in src and not os.path.exists(src):
    import multi_copy
    kind, n = src.split(":")
    rel = {"poseidon": "ecne_circomlib_tests/Poseidon@poseidon.r1cs", "eddsa": "ecne_circomlib_tests/EdDSAMiMCSpongeVerifier@eddsamimcsponge.r1cs"}[kind]
    src = multi_copy.cached(rel, int(n))
s = E.System(E.R1CS(src))
nwgs = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0]
modes = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0]
for mode in modes:
    for nwg in nwgs:
        best = None
        for rep in range(3):
            r = E.solve_batch([s], fetch_states=False, queue_mode=mode, force_nwg=nwg)[0]
            if best is None or r.summary.device_ms < best.summary.device_ms:
                best = r
        sm = best.summary
        mm = list(sm.multi_ms)
        print("mode %d nwg %3d rows %d status %d dev_ms %.3f pops %d rounds %d multi %d | queue[wave,multi] %.2f %.2f | multi[mark,check,exec,expand,count,write] %s levels %d drains %d | fast rounds %d rows %d"
              % (mode, nwg, len(s), best.status, sm.device_ms, sm.pops, sm.rule_hits[13], sm.rule_hits[14] >> 16, sm.queue_ms[6], sm.queue_ms[7],
                 [round(x, 2) for x in mm[:6]], round(mm[6] * 1e5), round(mm[7] * 1e5), sm.sched[0], sm.sched[1]))
