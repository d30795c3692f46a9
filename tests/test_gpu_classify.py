"""k_classify_rows' shape words against a CPU classifier written from the reference's pattern tests (tests/classify_ref.py):
every fixture of the reference tree and the seeded fuzz systems (explicit zeros, repeated wire ids, un-reduced
coefficients, degenerate rows). SURVEY.md §7 S6 / VERDICT r1 item 3."""
import numpy as np
import pytest

import classify_ref
import fixtures
import fuzz_r1cs
import r1cs_py

pytestmark = pytest.mark.gpu


def _check(path, tag):
    import ecneproject_amd as E
    hdr, rows = r1cs_py.parse_file(path)
    s = E.System(E.R1CS(path))
    shape, ms, nbytes = E.classify(s)
    assert len(shape) == len(rows)
    want = np.array(classify_ref.classify(rows), dtype=np.uint32)
    got = shape & np.uint32(classify_ref.CHECKED)
    bad = np.nonzero(got != want)[0]
    assert len(bad) == 0, (tag, int(bad[0]), hex(int(got[bad[0]])), hex(int(want[bad[0]])), rows[int(bad[0])])


@pytest.mark.parametrize("rel", [r for r in fixtures.all_r1cs()])
def test_shape_words_every_fixture(rel):
    _check(fixtures.path(rel), rel)


def test_shape_words_fuzz(tmp_path):
    for seed in range(200):
        p = str(tmp_path / ("%d.r1cs" % seed))
        fuzz_r1cs.write(p, fuzz_r1cs.make(seed))
        _check(p, "fuzz %d" % seed)
    for seed in range(40):
        p = str(tmp_path / ("w%d.r1cs" % seed))
        fuzz_r1cs.write(p, fuzz_r1cs.make_wide(seed))
        _check(p, "wide %d" % seed)


def _non_unit_system(path, n, seed=5):
    """n single-variable rows whose divisor is neither 1 nor -1: R3 rows `c x + d = 0` and R2 rows `(a x + b)(c x + d) = 0` with random
    field coefficients -- what circom --O1 / --O2 output or hand-written R1CS presents; every one of them is DEFERRED by the streaming lanes"""
    import random
    rng = random.Random(seed)
    P = r1cs_py.P
    rows = []
    for i in range(n):
        x = 2 + i
        c, d = rng.randrange(2, P - 1), rng.randrange(0, P)
        if i % 3 == 2:
            a, b = rng.randrange(2, P - 1), rng.randrange(0, P)
            rows.append(([(x, a), (1, b)], [(x, c), (1, d)], []))
        else:
            rows.append(([], [], [(x, c), (1, d)]))
    r1cs_py.write(path, nwires=n + 1, nout=1, npub=0, nprv=1, rows=rows)
    return path


def test_non_unit_divisors_lane_fallback(tmp_path):
    """rows whose divisors need a real field inversion (deferred by the streaming pass) are classified one LANE each: shape words as the
    CPU classifier's, the solve -- which consumes the constants -num/den this pass stores -- bit-exact against the oracle, and the
    classification of 200 000 such rows stays a streaming-scale pass (one wavefront per row was the cliff)"""
    import ecneproject_amd as E
    import orc
    from gpu_common import assert_bit_exact
    p = _non_unit_system(str(tmp_path / "nonunit_small.r1cs"), 3000)
    _check(p, "non-unit small")
    g = E.solve_batch([E.System(E.R1CS(p))], device=0)[0]
    assert_bit_exact("non-unit divisors", g, orc.run(p))
    p = _non_unit_system(str(tmp_path / "nonunit_big.r1cs"), 200000, seed=9)
    s = E.System(E.R1CS(p))
    E.classify(s)
    ms = sorted(E.classify(s)[1] for _ in range(5))[2]
    print("classify 200 000 rows with non-unit divisors: %.3f ms" % ms)
    assert ms < 1.5, ms      # (0.51 ms on MI355X: one inversion per wavefront; one divergent EGCD per lane 5.3 ms)
    g = E.solve_batch([s], device=0)[0]
    assert_bit_exact("non-unit divisors, 200 000 rows", g, orc.run(p))
