"""Developer aid (GPU box): the same solves on two builds of the library (this tree and a copy of an older commit under old_build/),
in separate processes on the same box.   python tools/ab_old_new.py"""
import os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CODE = r'''
import os, sys
sys.path.insert(0, os.environ["AB_PKG"]); sys.path.insert(0, os.path.join(os.environ["AB_ROOT"], "tests"))
import ecneproject_amd as E, fixtures
def best(systems, **kw):
    b = None
    for rep in range(4):
        rs = E.solve_batch(systems, fetch_states=False, **kw)
        ms = max(r.summary.device_ms for r in rs)
        if b is None or ms < b:
            b = ms
            if len(rs) == 1:
                sm = rs[0].summary
                det = " | phases %s queue %s fast[n,rows,ms] %d %d %.2f general[n,rows,ms] %d %d %.2f rounds %d pops %d" % ([round(x, 2) for x in sm.phase_ms[:6]], [round(x, 2) for x in sm.queue_ms[:8]], sm.sched[0], sm.sched[1], sm.sched[2] * 1e-5, sm.sched[3], sm.sched[4], sm.sched[5] * 1e-5, sm.rule_hits[13], sm.pops)
            else:
                det = ""
    return "%.3f ms%s" % (b, det)
one = lambda rel: E.System(E.R1CS(fixtures.path(rel)))
s1 = one("ecne_circomlib_tests/EdDSAMiMCSpongeVerifier@eddsamimcsponge.r1cs")
print("EdDSAMiMCSponge %s" % best([s1]))
s2 = one("ecne_circomlib_tests/Poseidon@poseidon.r1cs")
print("Poseidon %s" % best([s2]))
s3 = one("secp256k1.r1cs")
for t, n in (("bigmultmodp.r1cs", "BigMultModP"), ("biglessthan.r1cs", "BigLessThan")):
    s3.abstract(E.R1CS(fixtures.path(t)), n)
print("secp256k1 + trusted %s" % best([s3], secp_solve=True))
import glob
files = sorted(glob.glob(os.path.join(os.environ["AB_ROOT"], "tests", "data", "ecne_circomlib_tests", "*.r1cs.xz")))
sysl = [one("ecne_circomlib_tests/" + os.path.basename(f)[:-3]) for f in files]
print("circomlib batch (%d files) %s" % (len(sysl), best(sysl)))
'''
builds = [("old", os.path.join(ROOT, "old_build")), ("mid", os.path.join(ROOT, "old_build2")), ("nodrain", os.path.join(ROOT, "old_build3")), ("new", ROOT)]
for name, pkg in [b for b in builds if os.path.isdir(os.path.join(b[1], "ecneproject_amd"))]:
    env = dict(os.environ, AB_PKG=pkg, AB_ROOT=ROOT)
    out = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True)
    print("==", name); print(out.stdout.strip()); 
    if out.returncode: print(out.stderr[-600:])
