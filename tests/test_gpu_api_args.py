"""SolveConstraintsSymbolic's list arguments through the shim (reference :583-592): known_variables, target_variables and
special_constraints given by the caller replace what the handle carries; num_variables must be the handle's. The oracle
only reads lists from files, so the I/O case compares with the oracle run on a second file that has the SAME rows and a
header whose counts produce the edited lists (ParseR1CS.jl:123)."""
import numpy as np
import pytest

import fixtures
import fuzz_r1cs
import orc
from gpu_common import assert_bit_exact, build_system

pytestmark = pytest.mark.gpu


def test_edited_io_lists_match_a_file_with_that_header(tmp_path, capsys):
    import ecneproject_amd as E
    for seed in (3, 11, 42, 77):
        spec = fuzz_r1cs.make(seed, allow_errors=False)
        a, b = str(tmp_path / ("a%d.r1cs" % seed)), str(tmp_path / ("b%d.r1cs" % seed))
        fuzz_r1cs.write(a, spec)
        # the same rows, one output more and one private input fewer
        spec2 = dict(spec, n_out=spec["n_out"] + 1, n_prv=max(spec["n_prv"] - 1, 0))
        fuzz_r1cs.write(b, spec2)
        fb, kn_b, out_b, nv_b = E.readR1CS(b)
        o = orc.run(b)
        s = E.System(E.R1CS(a))
        ok = None
        try:
            ok = E.SolveConstraintsSymbolic(s, None, kn_b, False, out_b, nv_b, "")
        except E.EcneError as e:
            assert e.status == o.status
        g = E.last_result
        assert_bit_exact("edited io seed %d" % seed, g, o)
        if ok is not None:
            assert ok == o.verdict
    capsys.readouterr()


def test_same_specials_passed_explicitly_and_removed(capsys):
    import ecneproject_amd as E
    rel, tr, nm = "secp256k1.r1cs", ["bigmultmodp.r1cs", "biglessthan.r1cs"], ["BigMultModP", "BigLessThan"]
    o = orc.run(fixtures.path(rel), [fixtures.path(t) for t in tr], nm, True)
    s = build_system(rel, tr, nm)
    sp = s.specials()
    kn, tg = s.io()
    assert E.SolveConstraintsSymbolic(s, sp, kn, False, tg, -1, "", True) is True            # test/runtests.jl:35
    assert_bit_exact("explicit specials", E.last_result, o)
    # without the special constraints the reduced rows alone do not determine the outputs
    assert E.SolveConstraintsSymbolic(s, [], kn, False, tg, -1, "", True) is False
    assert s.specials() == []
    # and back again
    assert E.SolveConstraintsSymbolic(s, sp, None, False, None, -1, "", True) is True
    assert_bit_exact("specials restored", E.last_result, o)
    with pytest.raises(ValueError):
        E.SolveConstraintsSymbolic(s, None, None, False, None, 5, "", True)
    capsys.readouterr()


def test_full_report_division(capsys):
    """the text SolveConstraintsSymbolic prints with input_sym set: counts, Bad Constraints (README.md:102-105), All Variables"""
    import ecneproject_amd as E
    ok = E.solveWithTrustedFunctions(fixtures.path("target/division.r1cs"), "division", input_sym=fixtures.path("target/division.sym"))
    out = capsys.readouterr().out
    assert ok is False
    assert "Solved for 5 variables out of 7 total variables" in out
    assert "Solved for 0 target variables out of 1 total target variables" in out
    i_bad, i_all = out.index("------ Bad Constraints ------"), out.index("------ All Variables ------")
    bad, allv = out[i_bad:i_all], out[i_all:]
    assert "constraint #2\n(-1 * main.y2) * (1 * main.x3) = (-1 * main.y1)\n" in bad
    assert "constraint #3\n0 * 0 = (-1 * main.x4 + -1 * main.out + 1 * main.y2)\n" in bad
    assert "constraint #1" not in bad
    # every non-trivial variable but the constant wire, once
    for name in ("main.out", "main.x2", "main.x1", "main.x3", "main.x4", "main.y1", "main.y2"):
        assert allv.count(name + "\n") == 1, name
    assert allv.count("Uniquely Determined: true") == 5 and allv.count("Uniquely Determined: false") == 2      # "5 out of 7"
    assert "R1CS function division has potentially unsound constraints" in out


def test_always_on_prints_debug_dump_and_missing_sym(capsys):
    """the lines the reference prints whatever the flags ("time to prep inputs" :551, "setup solver" :704), the printState dump of
    debug=true between the two count lines (:1573-1577), println(specials) of abstractionOnly (:547), and CSV.File on a missing
    .sym (:1603) -- "default.sym", SolveConstraintsSymbolic's own default, included"""
    import re
    import ecneproject_amd as E
    ok = E.solveWithTrustedFunctions(fixtures.path("target/division.r1cs"), "division", debug=True)
    out = capsys.readouterr().out
    assert ok is False
    assert re.search(r"^time to prep inputs \d+ milliseconds$", out, re.M) and re.search(r"^setup solver \d+ milliseconds$", out, re.M)
    assert out.index("time to prep inputs") < out.index("setup solver") < out.index("Solved for 5 variables")
    a, b = out.index("Solved for 5 variables out of 7 total variables"), out.index("Solved for 0 target variables")
    dump = out[a:b]
    assert dump.count("Uniquely Determined: true") == 5 and dump.count("Uniquely Determined: false") == 2
    assert "main." not in dump                                   # printState only, no signal names
    assert out.rstrip().endswith("R1CS function division has potentially unsound constraints")
    assert E.solveWithTrustedFunctions(fixtures.path("secp256k1.r1cs"), "secp", trusted_r1cs=[fixtures.path("biglessthan.r1cs")],
                                       trusted_r1cs_names=["BigLessThan"], abstractionOnly=True) is True
    out = capsys.readouterr().out
    assert out.startswith("called abstraction\nAny[(\"BigLessThan\", [") and out.count("BigLessThan") == 4
    s = E.System(E.R1CS(fixtures.path("target/division.r1cs")))
    with pytest.raises(FileNotFoundError):
        E.SolveConstraintsSymbolic(s)                             # input_sym defaults to "default.sym"
    with pytest.raises(FileNotFoundError):
        E.SolveConstraintsSymbolic(s, None, None, False, None, -1, "/nonexistent/x.sym")


def test_result_states_after_free_or_resolve_is_einval():
    """a result's per-variable state can only be fetched while its system is alive and has not been solved again"""
    import ctypes as C
    import ecneproject_amd as E
    from ecneproject_amd import _lib
    L = _lib.lib()
    s = E.System(E.R1CS(fixtures.path("target/division.r1cs")))
    outs = (C.c_void_p * 1)()
    hs = (C.c_void_p * 1)(s._h)
    assert L.ecne_solve_batch(hs, 1, None, outs) == 0
    r1 = C.c_void_p(outs[0])
    assert L.ecne_solve_batch(hs, 1, None, outs) == 0           # solved again: r1's state is gone
    r2 = C.c_void_p(outs[0])
    p = C.POINTER(C.c_uint8)()
    assert L.ecne_result_states(r1, C.byref(p), None, None, None, None, None) == -9
    assert L.ecne_result_states(r2, C.byref(p), None, None, None, None, None) == 0
    outs3 = (C.c_void_p * 1)()
    assert L.ecne_solve_batch(hs, 1, None, outs3) == 0
    r3 = C.c_void_p(outs3[0])
    h = s._h
    s._h = None
    L.ecne_system_free(h)
    assert L.ecne_result_states(r3, C.byref(p), None, None, None, None, None) == -9       # system freed first
    sm = _lib.Summary()
    assert L.ecne_result_summary(r3, C.byref(sm)) == 0 and sm.n_rows == 3
    for r in (r1, r2, r3):
        L.ecne_result_free(r)


def test_per_system_secp_solve_in_one_batch():
    """jobs that want different secp_solve values in ONE launch (ecne_system_set_secp_solve): config 3 needs it (:762), the secp256k1
    adder without trusted functions must not get it, and a system with BigMultModP x BigLessThan pairs but the flag off raises UndefVarError"""
    import ecneproject_amd as E
    tr, names = ["bigmultmodp.r1cs", "biglessthan.r1cs"], ["BigMultModP", "BigLessThan"]
    a, b, c = build_system("secp256k1.r1cs", tr, names), build_system("target/division.r1cs"), build_system("secp256k1.r1cs", tr, names)
    a.set_secp_solve(True); b.set_secp_solve(False); c.set_secp_solve(False)
    ra, rb, rc = E.solve_batch([a, b, c], secp_solve=False)
    assert_bit_exact("secp+trusted", ra, orc.run(fixtures.path("secp256k1.r1cs"), [fixtures.path(t) for t in tr], names, True))
    assert_bit_exact("division", rb, orc.run(fixtures.path("target/division.r1cs")))
    assert rc.status == -4                                      # UndefVarError: dsu
    for s in (a, b, c):
        s.set_secp_solve(None)
    ra2, rb2, rc2 = E.solve_batch([a, b, c], secp_solve=True)   # back to "what the launch says"
    assert ra2.status == 0 and ra2.function_good and rc2.status == 0 and rc2.function_good and rb2.status == 0


@pytest.mark.gpu
def test_warmup_is_optional_and_repeatable():
    """ecne_warmup (include/ecne.h): pays the cold-process costs up front; callable any number of times, leaves the current device alone,
    refuses a device that is not there"""
    import ecneproject_amd as E
    a = E.warmup(0)
    b = E.warmup(0)
    assert a > 0 and b > 0
    with pytest.raises(E.NoDeviceError):
        E.warmup(E.device_count() + 3)
    g = E.solve_batch([E.System(E.R1CS(fixtures.path("target/division.r1cs")))])[0]
    assert g.status == 0 and g.function_good is False
