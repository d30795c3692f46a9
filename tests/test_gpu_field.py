"""Device field / integer arithmetic (fp256.hpp as compiled for gfx950) against Python integers,
through the C ABI's ecne_fp_selftest.  Covers the operations the rules use: +, -, *, inverse, negation,
field division (R2/R3/P4, reference :919-920, :961-964, :1467), the integer quotient of R7
(:1267-1268, with its fast paths: 64-bit operands, power-of-two divisors, divisor 1) and the
`coefficient * (ub + 1) <= p` test (:1274)."""
import random

import numpy as np
import pytest

P = 21888242871839275222246405745257275088548364400416034343698204186575808495617


def _limbs(xs):
    a = np.zeros((len(xs), 4), np.uint64)
    for i, x in enumerate(xs):
        for w in range(4):
            a[i, w] = (x >> (64 * w)) & 0xFFFFFFFFFFFFFFFF
    return a


def _ints(a):
    return [sum(int(a[i, w]) << (64 * w) for w in range(4)) for i in range(a.shape[0])]


def _run(op, xs, ys):
    from ecneproject_amd import _lib
    L = _lib.lib()
    a, b = _limbs(xs), _limbs(ys)
    out = np.zeros_like(a)
    st = L.ecne_fp_selftest(0, op, len(xs), a.ctypes.data, b.ctypes.data, out.ctypes.data)
    assert st == 0
    return _ints(out)


def _field_vectors(n, seed):
    rnd = random.Random(seed)
    edge = [0, 1, 2, P - 1, P - 2, (P - 1) // 2, (P + 1) // 2, 1 << 64, (1 << 64) - 1, 1 << 128, 1 << 192, 1 << 253]
    xs = edge + [rnd.randrange(P) for _ in range(n)]
    ys = list(reversed(edge)) + [rnd.randrange(P) for _ in range(n)]
    return xs, ys


@pytest.mark.gpu
def test_field_ops_match_python():
    xs, ys = _field_vectors(2000, 1)
    assert _run(0, xs, ys) == [(x + y) % P for x, y in zip(xs, ys)]
    assert _run(1, xs, ys) == [(x - y) % P for x, y in zip(xs, ys)]
    assert _run(2, xs, ys) == [(x * y) % P for x, y in zip(xs, ys)]
    assert _run(4, xs, ys) == [(-x) % P for x in xs]
    assert _run(3, xs, ys) == [pow(x, -1, P) if x else 0 for x in xs]
    assert _run(5, xs, ys) == [(x * pow(y, -1, P)) % P if y else 0 for x, y in zip(xs, ys)]


@pytest.mark.gpu
def test_integer_quotient_and_product_bound():
    rnd = random.Random(2)
    xs, ys = [], []
    for _ in range(1500):
        la, lb = rnd.randrange(1, 255), rnd.randrange(1, 255)
        xs.append(rnd.randrange(1 << (la - 1), 1 << la))
        ys.append(rnd.randrange(1 << (lb - 1), 1 << lb))
    for i in range(254):                      # power-of-two divisors, divisor 1, equal operands, 64-bit operands
        a = rnd.randrange(1, P)
        xs += [a, a, a, rnd.randrange(1, 1 << 64), (1 << i) + (rnd.randrange(1 << i) if i else 0)]
        ys += [1 << i, 1, a, rnd.randrange(1, 1 << 64), 1 << i]
    assert _run(6, xs, ys) == [x // y for x, y in zip(xs, ys)]
    assert _run(7, xs, ys) == [1 if x * y > P else 0 for x, y in zip(xs, ys)]
