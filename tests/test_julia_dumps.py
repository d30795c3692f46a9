"""Drop-in slot for per-variable dumps of the REAL reference (julia/dump_unique.jl, run by anybody who has Julia 1.7 and an Ecne
checkout): every tests/golden/julia/*.tsv is compared -- verdict and the whole per-variable state -- with the oracle and with the
second reading tests/ref2.py. The build image has no Julia, so the directory is empty here and the test is skipped; the day a dump
is dropped in, the per-variable state stops being "parity unpinned"  without any code change."""
import glob
import os

import pytest

import fixtures
import orc
import ref2

HERE = os.path.dirname(os.path.abspath(__file__))
DUMPS = sorted(glob.glob(os.path.join(HERE, "golden", "julia", "*.tsv")))


def read_dump(path):
    verdict, rows, inp = None, {}, None
    with open(path) as f:
        for line in f:
            line = line.rstrip("\n")
            if line.startswith("# input"):
                c = line.split("\t")
                inp = (c[1], c[2] == "1", [tuple(x.split("=")) for x in c[3].split(",") if x] if len(c) > 3 else [])
            elif line.startswith("# verdict"):
                verdict = line.split("\t")[1].strip() == "true"
            elif line and not line.startswith("#"):
                c = line.split("\t")
                vals = sorted(int(x) for x in c[6].split(",")) if len(c) > 6 and c[6] else []
                rows[int(c[0])] = (int(c[1]), int(c[2]), int(c[3]), int(c[4]), int(c[5]), vals)
    return inp, verdict, rows


def _fixture_by_basename(name):
    hits = [r for r in fixtures.all_r1cs() if os.path.basename(r) == name]
    return fixtures.path(hits[0]) if hits else None


@pytest.mark.skipif(not DUMPS, reason="no reference dumps under tests/golden/julia/ (Julia is not available in the build image)")
@pytest.mark.parametrize("dump", DUMPS)
def test_dump_matches_both_restatements(dump):
    inp, verdict, rows = read_dump(dump)
    assert inp is not None, "dump without an '# input' line: re-create it with the current julia/dump_unique.jl"
    main = _fixture_by_basename(inp[0])
    assert main, inp[0]
    trusted = [_fixture_by_basename(t) for t, _n in inp[2]]
    names = [n for _t, n in inp[2]]
    o = orc.run(main, trusted, names, inp[1])
    r = ref2.run(main, trusted, names, inp[1])
    assert o.status == 0 and r.status == 0
    assert o.verdict == verdict and r.verdict == verdict
    bad_o, bad_r = [], []
    for v, (u, k, lb, ub, abz, vals) in sorted(rows.items()):
        mine = (int(o.flags[v - 1] & 1), int((o.flags[v - 1] >> 1) & 1), orc.limbs_to_int(o.lb[v - 1]), orc.limbs_to_int(o.ub[v - 1]), int(o.abz[v - 1]),
                sorted(orc.limbs_to_int(o.values[v - 1][i]) for i in range(int(o.nvalues[v - 1]))))
        if mine != (u, k, lb, ub, abz, vals):
            bad_o.append(v)
        st = r.states[v - 1]
        if (int(st.unique), int(st.is_known), st.lb, st.ub, st.abz, sorted(st.values)) != (u, k, lb, ub, abz, vals):
            bad_r.append(v)
    assert not bad_o and not bad_r, (bad_o[:10], bad_r[:10])


def test_dump_reader_on_a_synthetic_dump(tmp_path):
    """the reader and the comparison themselves, on a dump written from the oracle's state in the script's format"""
    rel = "target/division.r1cs"
    o = orc.run(fixtures.path(rel))
    p = tmp_path / "division.tsv"
    with open(p, "w") as f:
        f.write("# input\tdivision.r1cs\t0\t\n# verdict\t%s\n# var\tunique\tis_known\tlb\tub\tabz\tvalues\n" % ("true" if o.verdict else "false"))
        for i in range(len(o.flags)):
            vals = sorted(orc.limbs_to_int(o.values[i][k]) for k in range(int(o.nvalues[i])))
            f.write("%d\t%d\t%d\t%d\t%d\t%d\t%s\n" % (i + 1, o.flags[i] & 1, (o.flags[i] >> 1) & 1, orc.limbs_to_int(o.lb[i]), orc.limbs_to_int(o.ub[i]),
                                                    o.abz[i], ",".join(map(str, vals))))
    inp, verdict, rows = read_dump(str(p))
    assert inp == ("division.r1cs", False, []) and verdict is False and len(rows) == 8
    r = ref2.run(fixtures.path(rel))
    for v, (u, k, lb, ub, abz, vals) in rows.items():
        st = r.states[v - 1]
        assert (int(st.unique), int(st.is_known), st.lb, st.ub, st.abz, sorted(st.values)) == (u, k, lb, ub, abz, vals)
