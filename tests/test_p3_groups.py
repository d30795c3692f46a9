"""P3 (:1357-1417): k linear rows over the same k unknowns.  The group fires when its k-th row arrives and
the reference's `slow_det` (the sum over ODD permutations only, :1389-1400) is non-zero.  k = 4..7 with
random coefficients, a group whose odd-permutation sum is zero, and the documented deviation for more than
10 unknowns (ECNE_EDETSIZE instead of k! * k steps)."""
import random

import pytest

import orc
import r1cs_py

P = r1cs_py.P


def _system(k, seed, singular=False, extra_rows=0):
    rng = random.Random(seed)
    xs = list(range(2, 2 + k))                      # unknowns (internal signals; no inputs, no outputs)
    w = {x: rng.randrange(1, 1000) for x in xs}
    rows = []
    for r in range(k + extra_rows):
        cs = [rng.randrange(1, 50) for _ in xs]
        if singular and r == k - 1:
            cs = [c for c in rows_coefs[0]]          # repeat the first row: every permutation product pairs up
        rows_coefs = rows_coefs + [cs] if r else [cs]
        const = (-sum(c * w[x] for c, x in zip(cs, xs))) % P
        rows.append(([], [], [(x, c) for x, c in zip(xs, cs)] + [(1, const)]))
    return dict(nwires=k + 1, nout=0, npub=0, nprv=0, rows=rows)


CASES = {
    "k4": (_system(4, 1), 0), "k5": (_system(5, 2), 0), "k6": (_system(6, 3), 0), "k7": (_system(7, 4), 0),
    "k5_plus_rows": (_system(5, 5, extra_rows=3), 0),
    "k4_repeated_row": (_system(4, 6, singular=True), 0),
    "k11": (_system(11, 7), -6),
}


@pytest.fixture(scope="module")
def p3_dir(tmp_path_factory):
    d = tmp_path_factory.mktemp("p3")
    for name, (spec, _) in CASES.items():
        r1cs_py.write(str(d / (name + ".r1cs")), spec["nwires"], spec["nout"], spec["npub"], spec["nprv"], spec["rows"])
    return d


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_p3_groups(p3_dir, name):
    o = orc.run(str(p3_dir / (name + ".r1cs")))
    assert o.status == CASES[name][1]
    if name in ("k4", "k5", "k6", "k7", "k5_plus_rows"):
        assert o.summary.rule_hits[10] == 1 and o.unique[1:-1].all()     # the group fired, every unknown unique


@pytest.mark.gpu
@pytest.mark.parametrize("force_nwg", [0, 2])
def test_gpu_p3_groups_parity(p3_dir, force_nwg):
    import ecneproject_amd as E
    from gpu_common import assert_bit_exact
    names = sorted(CASES)
    systems = [E.System(E.R1CS(str(p3_dir / (n + ".r1cs")))) for n in names]
    for n, g in zip(names, E.solve_batch(systems, force_nwg=force_nwg)):
        assert_bit_exact("p3 " + n, g, orc.run(str(p3_dir / (n + ".r1cs"))))
