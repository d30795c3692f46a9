"""SolveConstraintsSymbolic's own list arguments (src/R1CSConstraintSolver.jl:583-592): `known_variables` is taken as the caller gives it
-- the setup (:682-693) makes exactly those variables unique, so a list WITHOUT the constant wire leaves variable 1 like any other
unknown (no values, not unique); duplicates count twice in `unknown_variable_count` only; order does not matter. The engine pads short
rows / special input lists with variable 1 in several fetch loops and its parallel schedules rely on the wire's `unique` never changing:
such a list takes the strictly sequential path and the padding is masked (round-4 advisor finding, P1 :723-736).

CPU: oracle (orc.run(known_variables=...)) against the second reading (ref2.solve takes the lists directly).
GPU: the HIP engine (System.set_io) against the oracle, bit for bit, with trusted functions (9 and 6 inputs: not multiples of four)."""
import random

import pytest

import fixtures
import fuzz_r1cs
import orc

N_SEEDS = 160


def known_lists(rng, file_knowns, nv):
    """lists to try for one system: without the constant wire, empty, shuffled, with a duplicate, with extra variables"""
    rest = [v for v in file_knowns if v != 1]
    out = [list(rest), []]
    sh = list(file_knowns)
    rng.shuffle(sh)
    out.append(sh)
    if rest:
        out.append(rest + [rng.choice(rest)])                       # duplicate, no constant wire
        out.append([v for v in rest if rng.random() < 0.6])        # a subset
    extra = [v for v in range(2, nv + 1) if v not in file_knowns]
    if extra:
        out.append(rest + rng.sample(extra, min(len(extra), rng.randint(1, 3))))
    return out


@pytest.fixture(scope="module")
def sys_dir(tmp_path_factory):
    d = tmp_path_factory.mktemp("known_lists")
    for seed in range(N_SEEDS):
        fuzz_r1cs.write(str(d / ("%d.r1cs" % seed)), fuzz_r1cs.make(seed + 31000, allow_errors=seed % 5 == 0))
    return d


def cases(sys_dir):
    rng = random.Random(77)
    for seed in range(N_SEEDS):
        p = str(sys_dir / ("%d.r1cs" % seed))
        st, info = orc.read_info(p)
        assert st == 0
        for kn in known_lists(rng, info["knowns"], info["nVars"]):
            yield p, kn, info["outputs"], seed % 3 == 0


def _ref2(path, knowns, targets, secp, trusted=(), names=()):
    import ref2
    R = ref2.Result()
    try:
        eqs, _kn, _outs, nv = ref2.read_r1cs(path)
        fl = []
        for pth, nm in zip(trusted, names):
            e, k, o, _ = ref2.read_r1cs(pth)
            fl.append((nm, e, k, o))
        fl.sort(key=lambda x: -len(x[1]))
        specials, red = [], eqs
        for nm, e, k, o in fl:
            new, red = ref2.abstraction(nm, red, k, e, o)
            specials.extend(new)
        out = ref2.solve(red, specials, list(knowns), list(targets), nv, secp)
        out.specials, out.n_rows_reduced = specials, len(red)
        return out
    except (ref2.BoundsError, ref2.DivideError, ref2.UndefVarError, ref2.JlKeyError, ref2.Watchdog, ref2.FormatError) as e:
        R.status = e.status
        return R


def test_oracle_and_second_reading_agree_on_callers_lists(sys_dir):
    from test_ref2 import differences
    n = without_one_changes = 0
    for p, kn, tg, secp in cases(sys_dir):
        o = orc.run(p, secp_solve=secp, known_variables=kn, target_variables=tg)
        assert differences(_ref2(p, kn, tg, secp), o) == [], (p, kn)
        n += 1
        if 1 not in kn and o.status == 0:
            assert not (o.flags[0] & 1) or o.summary.successful_steps > 0      # the wire is unique only if a rule made it so
            without_one_changes += int(tuple(o.counts()) != tuple(orc.run(p, secp_solve=secp, known_variables=kn + [1], target_variables=tg).counts()))
    assert n >= 800 and without_one_changes >= 50


TRUSTED = [("secp256k1.r1cs", ["bigmultmodp.r1cs", "biglessthan.r1cs"], ["BigMultModP", "BigLessThan"]),
           ("tornadocash_circuits/commitHasher.r1cs", fixtures.PED, fixtures.PED_NAMES)]


def _trusted_cases():
    for rel, trusted, names in TRUSTED:
        p = fixtures.path(rel)
        tp = [fixtures.path(t) for t in trusted]
        st, info = orc.read_info(p)
        assert st == 0
        rest = [v for v in info["knowns"] if v != 1]
        yield p, tp, names, rest, info["outputs"]
        yield p, tp, names, rest[:-1], info["outputs"]


def test_trusted_functions_without_the_constant_wire_oracle_vs_second_reading():
    from test_ref2 import differences
    n = 0
    for p, tp, names, kn, tg in _trusted_cases():
        o = orc.run(p, tp, names, True, known_variables=kn, target_variables=tg)
        assert differences(_ref2(p, kn, tg, True, tp, names), o) == [], (p, len(kn))
        n += 1
    assert n >= 4


@pytest.mark.gpu
def test_gpu_callers_lists(sys_dir):
    import ecneproject_amd as E
    from gpu_common import assert_bit_exact
    todo = list(cases(sys_dir))
    files = {}
    systems = []
    for p, kn, tg, secp in todo:
        f = files.get(p) or files.setdefault(p, E.R1CS(p))
        s = E.System(f)
        s.set_io(kn, tg)
        s.set_secp_solve(secp)
        systems.append(s)
    results = []
    for i in range(0, len(systems), 120):
        results += E.solve_batch(systems[i:i + 120])
    for (p, kn, tg, secp), g in zip(todo, results):
        o = orc.run(p, secp_solve=secp, known_variables=kn, target_variables=tg)
        assert_bit_exact("known_variables %s %r" % (p, kn), g, o)


@pytest.mark.gpu
@pytest.mark.parametrize("frontend", ["host", "device"])
def test_gpu_trusted_functions_without_the_constant_wire(frontend):
    """P1 (:718-747) with input lists of 9 / 6 / 12 entries while variable 1 is not unique: the padded fetch must not ask the wire"""
    import ecneproject_amd as E
    from gpu_common import assert_bit_exact, build_system
    prev = E.set_frontend(-1)
    E.set_frontend(E.FRONTEND_DEVICE if frontend == "device" else E.FRONTEND_HOST)
    try:
        n = 0
        for p, tp, names, kn, tg in _trusted_cases():
            s = build_system(None, [], [], path=p)
            fl = sorted([(nm, E.R1CS(t)) for t, nm in zip(tp, names)], key=lambda x: -len(x[1]))
            for nm, f in fl:
                s.abstract(f, nm)
            s.set_io(kn, tg)
            g = E.solve_batch([s], secp_solve=True)[0]
            assert_bit_exact("trusted, no constant wire: %s" % p, g, orc.run(p, tp, names, True, known_variables=kn, target_variables=tg))
            n += 1
        assert n >= 4
    finally:
        E.set_frontend(prev)
