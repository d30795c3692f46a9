"""Developer aid: run fixtures on the GPU engine and on the oracle, print first differences.
Usage (on the GPU box): python tests/tools/gpu_parity_debug.py [substring ...]"""
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
import numpy as np  # noqa: E402

import ecneproject_amd as E  # noqa: E402
import fixtures  # noqa: E402
import orc  # noqa: E402

CASES = [(rel, [], [], False) for rel in fixtures.all_r1cs()]
CASES += [("tornadocash_circuits/commitHasher.r1cs", fixtures.PED, fixtures.PED_NAMES, False),
          ("tornadocash_circuits/withdraw.r1cs", fixtures.PED, fixtures.PED_NAMES, False),
          ("secp256k1.r1cs", ["bigmultmodp.r1cs", "biglessthan.r1cs"], ["BigMultModP", "BigLessThan"], True),
          ("secp256k1.r1cs", ["bigmultmodp.r1cs", "biglessthan.r1cs"], ["BigMultModP", "BigLessThan"], False)]


def build_system(rel, trusted, names):
    main = E.R1CS(fixtures.path(rel))
    fl = [(n, E.R1CS(fixtures.path(t))) for t, n in zip(trusted, names)]
    fl.sort(key=lambda x: -len(x[1]))
    s = E.System(main)
    for n, f in fl:
        s.abstract(f, n)
    return s


def compare(rel, g, o):
    msgs = []
    if g.status != o.status:
        msgs.append("status gpu=%d oracle=%d" % (g.status, o.status))
        return msgs
    if o.status != 0:
        return msgs
    if g.function_good != o.verdict:
        msgs.append("verdict gpu=%s oracle=%s" % (g.function_good, o.verdict))
    if tuple(g.counts()) != tuple(o.counts()):
        msgs.append("counts gpu=%s oracle=%s" % (g.counts(), o.counts()))
    for name, a, b in [("flags", g.flags, o.flags), ("abz", g.abz.astype(np.int64), o.abz), ("nvalues", g.nvalues, o.nvalues)]:
        d = np.nonzero(a != b)[0]
        if len(d):
            msgs.append("%s differ at %d vars, first v=%d gpu=%s oracle=%s" % (name, len(d), d[0] + 1, a[d[0]], b[d[0]]))
    for name, a, b in [("lb", g.lb, o.lb), ("ub", g.ub, o.ub)]:
        d = np.nonzero((a != b).any(axis=1))[0]
        if len(d):
            msgs.append("%s differ at %d vars, first v=%d gpu=%s oracle=%s" % (name, len(d), d[0] + 1, orc.limbs_to_int(a[d[0]]), orc.limbs_to_int(b[d[0]])))
    gv = g.values.reshape(len(g.flags), 8)
    ov = o.values.reshape(len(o.flags), 8)
    d = np.nonzero((gv != ov).any(axis=1))[0]
    if len(d):
        msgs.append("values differ at %d vars, first v=%d" % (len(d), d[0] + 1))
    gs, os_ = g.summary, o.summary
    for f in ("successful_steps", "outer_iterations", "pops", "num_unique"):
        if getattr(gs, f) != getattr(os_, f):
            msgs.append("%s gpu=%d oracle=%d" % (f, getattr(gs, f), getattr(os_, f)))
    gh, oh = list(gs.rule_hits[:13]), list(os_.rule_hits[:13])
    if gh != oh:
        msgs.append("rule_hits gpu=%s oracle=%s" % (gh, oh))
    if g.bad_rows.tolist() != o.bad_rows.tolist():
        msgs.append("bad_rows differ")
    return msgs


def main():
    nwg = int(os.environ.get("ECNE_FORCE_NWG", "0"))
    subs = sys.argv[1:]
    cases = [c for c in CASES if not subs or any(s in c[0] for s in subs)]
    nbad = 0
    t0 = time.time()
    for rel, trusted, names, secp in cases:
        sysm = build_system(rel, trusted, names)
        t = time.time()
        g = E.solve_batch([sysm], secp_solve=secp, force_nwg=nwg)[0]
        tg = time.time() - t
        o = orc.run(fixtures.path(rel), [fixtures.path(x) for x in trusted], names, secp)
        msgs = compare(rel, g, o)
        tag = "OK " if not msgs else "BAD"
        nbad += bool(msgs)
        print("%s %-66s rows=%6d st=%d dev=%.2fms cls=%.3fms wall=%.1fms oracle_solve=%.1fms" %
              (tag, rel + ("+T" if trusted else ""), len(sysm), g.status, g.summary.device_ms, g.summary.classify_ms,
               tg * 1e3, o.summary.t_solve * 1e3), flush=True)
        for m in msgs[:8]:
            print("      ", m)
    print("cases", len(cases), "bad", nbad, "total %.1fs" % (time.time() - t0))


if __name__ == "__main__":
    main()
