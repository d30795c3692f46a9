"""Compare a per-variable dump of the REFERENCE solver (julia/dump_unique.jl, run by somebody who has Julia 1.7 and an
Ecne checkout) with the HIP engine's result for the same input — the one route by which the per-variable state can be
pinned to the real reference. Never run by the test-suite (no Julia in the build image).

    python tests/tools/compare_julia_dump.py dump.tsv main.r1cs [--secp] [trusted.r1cs Name]...      (needs a GPU)
"""
import os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))


def read_dump(path):
    verdict, rows = None, {}
    with open(path) as f:
        for line in f:
            line = line.rstrip("\n")
            if line.startswith("# verdict"):
                verdict = line.split("\t")[1].strip() == "true"
            elif line and not line.startswith("#"):
                c = line.split("\t")
                vals = sorted(int(x) for x in c[6].split(",")) if len(c) > 6 and c[6] else []
                rows[int(c[0])] = (int(c[1]), int(c[2]), int(c[3]), int(c[4]), int(c[5]), vals)
    return verdict, rows


def main():
    import ecneproject_amd as E
    args = [a for a in sys.argv[1:] if a != "--secp"]
    secp = "--secp" in sys.argv
    dump, main_path, rest = args[0], args[1], args[2:]
    verdict, ref = read_dump(dump)
    fl = sorted(((n, E.R1CS(t)) for t, n in zip(rest[0::2], rest[1::2])), key=lambda x: -len(x[1]))
    s = E.System(E.R1CS(main_path))
    for n, f in fl:
        s.abstract(f, n)
    g = E.solve_batch([s], secp_solve=secp)[0]
    g.raise_for_status()
    to_int = lambda a: sum(int(a[i]) << (64 * i) for i in range(4))      # noqa: E731
    bad = 0
    for v, (u, k, lb, ub, abz, vals) in sorted(ref.items()):
        mine = (int(g.flags[v - 1] & 1), int((g.flags[v - 1] >> 1) & 1), to_int(g.lb[v - 1]), to_int(g.ub[v - 1]), int(g.abz[v - 1]),
                sorted(to_int(g.values[v - 1][i]) for i in range(int(g.nvalues[v - 1]))))
        if mine != (u, k, lb, ub, abz, vals):
            bad += 1
            if bad <= 20:
                print("variable %d: reference %r, engine %r" % (v, (u, k, lb, ub, abz, vals), mine))
    print("verdict: reference %s, engine %s; %d of %d variables differ" % (verdict, g.function_good, bad, len(ref)))
    return 0 if (bad == 0 and verdict == g.function_good) else 1


if __name__ == "__main__":
    sys.exit(main())
