"""The oracle's field arithmetic (oracle/u256.hpp) against Python big integers.

Reference: AbstractAlgebra.GF(bjj_p) — /root/reference/src/R1CSConstraintSolver.jl:21-24; call
sites listed in SURVEY.md §8a row T21 (divexact, unary -, ==, F(2)^i, *, +)."""
import random

import orc

P = orc.P


def test_constants():
    assert P.bit_length() == 254
    st, r = orc.fp_op(5, 1)          # F(-1) = p - 1 (reference :145 ub default)
    assert st == 0 and r == P - 1
    st, r = orc.fp_op(5, 0)
    assert r == 0


def test_random_ops():
    rng = random.Random(20260928)
    edge = [0, 1, 2, P - 1, P - 2, (P - 1) // 2, 1 << 253, (1 << 253) - 1, 0xFFFFFFFFFFFFFFFF, 1 << 64]
    vals = edge + [rng.randrange(P) for _ in range(300)]
    for i, a in enumerate(vals):
        b = vals[(i * 7 + 3) % len(vals)]
        assert orc.fp_op(0, a, b)[1] == (a + b) % P
        assert orc.fp_op(1, a, b)[1] == (a - b) % P
        assert orc.fp_op(2, a, b)[1] == (a * b) % P
        assert orc.fp_op(5, a)[1] == (-a) % P
        if a:
            assert orc.fp_op(3, a)[1] == pow(a, -1, P)
        if b:
            assert orc.fp_op(4, a, b)[1] == (a * pow(b, -1, P)) % P


def test_divide_by_zero_is_an_error():
    # AbstractAlgebra divexact by zero raises DivideError (reference :919-920, :1467)
    st, _ = orc.fp_op(3, 0)
    assert st == -3
    st, _ = orc.fp_op(4, 5, 0)
    assert st == -3
