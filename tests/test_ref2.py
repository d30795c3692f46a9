"""N-version check of the oracle: tests/ref2.py -- a second, independent reading of the reference's solver path in plain Python
(its own Julia Dict / Set model, objects with the reference's aliasing semantics) -- against oracle/ecne_oracle.cpp.
Compared: status, verdict, the four printed counts, successful_steps / num_unique / pops / outer iterations, the special
constraints abstraction produces, and the WHOLE per-variable state (unique, is_known, lb, ub, abz, values).
Any disagreement is a finding about /root/reference/src/R1CSConstraintSolver.jl."""
import json
import lzma
import os
import random

import fixtures
import fuzz_r1cs
import orc
import ref2

HERE = os.path.dirname(os.path.abspath(__file__))
FULL = os.environ.get("ECNE_REF2_FULL", "1") != "0"      # ECNE_REF2_FULL=0: a third of the seeds, fixtures up to 6 000 rows


def test_ref2_dict_and_set_order_against_the_reference_dumps():
    """ref2's own Julia 1.7 Dict / Set model on the 16 620 known-answer vectors extracted from the reference's equation dumps"""
    with open(os.path.join(HERE, "golden", "julia_order_kat.json.xz"), "rb") as f:
        vecs = json.loads(lzma.decompress(f.read()))
    assert len(vecs) == 16620
    bad = 0
    for kind, keys, expect in vecs:
        d = ref2.JDict()
        for k in keys:
            d[k] = 1
        ks = d.keys()
        if kind == 2:
            d2 = ref2.JDict()
            for k in ks:
                d2[k] = 1
            ks = d2.keys()
        s = ref2.JSet()
        for k in ks:
            s.push(k)
        bad += list(s) != expect
    assert bad == 0


def differences(r, o):
    """ref2 result r vs oracle result o -> list of what differs"""
    if r.status != o.status:
        return [("status", r.status, o.status)]
    if r.status != 0:
        return []
    d = []
    if r.verdict != o.verdict:
        d.append(("verdict", r.verdict, o.verdict))
    if (r.unique_nontrivial, r.n_nontrivial, r.unique_targets, r.n_targets) != tuple(o.counts()):
        d.append(("counts", (r.unique_nontrivial, r.n_nontrivial, r.unique_targets, r.n_targets), tuple(o.counts())))
    s = o.summary
    mine, theirs = (r.successful_steps, r.num_unique, r.pops, r.outer_iterations), (s.successful_steps, s.num_unique, s.pops, s.outer_iterations)
    if mine != theirs:
        d.append(("steps/num_unique/pops/outer", mine, theirs))
    if [(list(x[1]), list(x[2])) for x in r.specials] != [(list(x[1]), list(x[2])) for x in o.specials]:
        d.append(("specials",))
    if r.n_rows_reduced != s.n_rows_reduced:
        d.append(("reduced rows", r.n_rows_reduced, s.n_rows_reduced))
    bad = []
    for i, st in enumerate(r.states):
        fl = (1 if st.unique else 0) | (2 if st.is_known else 0)
        if (fl != int(o.flags[i]) or st.lb != orc.limbs_to_int(o.lb[i]) or st.ub != orc.limbs_to_int(o.ub[i]) or st.abz != int(o.abz[i]) or
                len(st.values) != int(o.nvalues[i]) or any(st.values[k] != orc.limbs_to_int(o.values[i][k]) for k in range(min(len(st.values), 2)))):
            bad.append(i + 1)
    if bad:
        d.append(("state of variables", bad[:10], len(bad)))
    return d


def check(path, trusted=(), names=(), secp=False):
    return differences(ref2.run(path, list(trusted), list(names), secp), orc.run(path, list(trusted), list(names), secp))


def test_reference_asserted_configurations():
    """the nine booleans of test/runtests.jl, the README transcript -- and full state agreement with the oracle on each"""
    for rel, trusted, names, secp, verdict in fixtures.REFERENCE_ASSERTED:
        tp = [fixtures.path(t) for t in trusted]
        r = ref2.run(fixtures.path(rel), tp, names, secp)
        assert r.status == 0 and r.verdict == verdict, rel
        assert differences(r, orc.run(fixtures.path(rel), tp, names, secp)) == [], rel
    r = ref2.run(fixtures.path("target/division.r1cs"))
    assert (r.unique_nontrivial, r.n_nontrivial, r.unique_targets, r.n_targets) == (5, 7, 0, 1)       # README.md:95-107
    # secp256k1 with BigMultModP x BigLessThan pairs but without secp_solve: UndefVarError `dsu` (:762)
    tp = [fixtures.path("bigmultmodp.r1cs"), fixtures.path("biglessthan.r1cs")]
    assert ref2.run(fixtures.path("secp256k1.r1cs"), tp, ["BigMultModP", "BigLessThan"], False).status == -4


def test_every_fixture_against_the_oracle():
    """all fixture files (up to 24 316 rows): whole-state agreement"""
    n = 0
    for rel in fixtures.all_r1cs():
        p = fixtures.path(rel)
        st, info = orc.read_info(p)
        if st != 0 or (not FULL and info["nConstraints"] > 6000):
            continue
        assert check(p) == [], rel
        n += 1
    assert n >= 60


def test_bits2point_strict_disagrees_with_its_own_example():
    """examples/Bits2Point_Strict.jl:4 asserts true; both readings of the reference's rules say false"""
    rel = [r for r in fixtures.all_r1cs() if "Bits2Point_Strict" in r]
    assert rel
    r = ref2.run(fixtures.path(rel[0]))
    assert r.status == 0 and r.verdict is False


def test_fuzz_families(tmp_path):
    """the seeded random systems of test_fuzz.py (degenerate rows, repeated ids, explicit zeros, error rows, contradictions) and of
    the abstraction fuzz (overlapping windows, near copies, KeyError)"""
    import r1cs_py
    import test_abstraction_fuzz as TA
    n = 0
    for seed in range(0, 400, 1 if FULL else 3):
        p = str(tmp_path / ("f%d.r1cs" % seed))
        fuzz_r1cs.write(p, fuzz_r1cs.make(seed))
        assert check(p) == [], seed
        n += 1
    for seed in range(0, TA.N_CASES, 1 if FULL else 3):
        rng = random.Random(77000 + seed)
        sub, tag = TA._rand_sub(rng)
        main = TA._rand_main(rng, sub, tag)
        sp, mp = str(tmp_path / ("sub%d.r1cs" % seed)), str(tmp_path / ("main%d.r1cs" % seed))
        r1cs_py.write(sp, sub["nwires"], sub["nout"], sub["npub"], sub["nprv"], sub["rows"])
        r1cs_py.write(mp, main["nwires"], main["nout"], main["npub"], main["nprv"], main["rows"])
        assert check(mp, [sp], ["T"]) == [], seed
        n += 1
    assert n > 200


def test_structured_families(tmp_path):
    """chains of 5 / 150 trusted-function instances and isZero pairs (cascading P1 / P5 sweeps, forward and reverse), P3 groups of
    4..7 unknowns (the odd-permutation sum), the long-row cases (R7 chains of 100 digits, R1 / R8 on 80..100-term rows, hub fan-out)"""
    import bigrow_cases
    import r1cs_py
    import test_p3_groups as TP
    import test_specials_chain as TC
    sub = str(tmp_path / "cube.r1cs")
    TC._write_sub(sub)
    for K in (5, 150):
        for rev in (False, True):
            p = str(tmp_path / ("chain_%d_%d.r1cs" % (K, rev)))
            TC._write_chain(p, K, rev)
            assert check(p, [sub], ["Cube"]) == [], ("chain", K, rev)
            p = str(tmp_path / ("iz_%d_%d.r1cs" % (K, rev)))
            TC._write_iszero_chain(p, K, rev)
            assert check(p) == [], ("iszero", K, rev)
    for name, (spec, status) in TP.CASES.items():
        if status != 0:
            continue                                  # (k = 11: the oracle's documented EDETSIZE deviation; the reference would run 11! * 11 steps)
        p = str(tmp_path / (name + ".r1cs"))
        r1cs_py.write(p, spec["nwires"], spec["nout"], spec["npub"], spec["nprv"], spec["rows"])
        assert check(p) == [], name
    for name, (spec, _rule, _fires) in bigrow_cases.CASES.items():
        p = str(tmp_path / (name + ".r1cs"))
        bigrow_cases.write(p, spec)
        assert check(p) == [], name


def test_ecdsa_like_small():
    """the synthetic ecdsa-scale generator at S = 1 (42 000 rows, 1 025-term decoder sums, one abstracted adder): the same state"""
    import ecdsa_like
    p = ecdsa_like.cached(1, 10)
    assert check(p, [fixtures.path("secp256k1.r1cs")], ["Secp256k1AddUnequal"]) == []
