"""Chains of trusted-function instances (P1, :718-747).  Main circuit = K copies of a two-row sub-circuit
(out = in^3) wired in a chain; abstraction turns every copy into a special constraint.  With the copies
in chain order ALL specials fire in the first P1 sweep, one after the other (each firing makes the next
one's input unique) -- the engine tests 64 specials at a time and must re-test after every firing, across
batch boundaries; in reverse order only one special fires per outer iteration."""
import pytest

import orc
import r1cs_py

P = r1cs_py.P


def _write_sub(path):
    # variables: 1 one, 2 out, 3 in, 4 tmp
    rows = [([(3, 1)], [(3, 1)], [(4, 1)]), ([(4, 1)], [(3, 1)], [(2, 1)])]
    r1cs_py.write(path, 3, 1, 1, 0, rows)


def _write_chain(path, K, reverse):
    # variables: 1 one, 2 = v_K (output), 3 = v_0 (input), 4 .. K+2 = v_1 .. v_{K-1}, then t_0 .. t_{K-1}
    v = [3] + list(range(4, 3 + K)) + [2]
    t = list(range(3 + K, 3 + 2 * K))
    windows = [[([(v[i], 1)], [(v[i], 1)], [(t[i], 1)]), ([(t[i], 1)], [(v[i], 1)], [(v[i + 1], 1)])] for i in range(K)]
    if reverse:
        windows.reverse()
    r1cs_py.write(path, 2 * K + 1, 1, 1, 0, [r for w in windows for r in w])


@pytest.fixture(scope="module")
def chain_files(tmp_path_factory):
    d = tmp_path_factory.mktemp("chain")
    _write_sub(str(d / "cube.r1cs"))
    for K in (5, 150):
        for rev in (False, True):
            _write_chain(str(d / ("chain_%d_%d.r1cs" % (K, rev))), K, rev)
    return d


@pytest.mark.parametrize("K,rev", [(5, False), (5, True), (150, False), (150, True)])
def test_oracle_chain(chain_files, K, rev):
    o = orc.run(str(chain_files / ("chain_%d_%d.r1cs" % (K, rev))), [str(chain_files / "cube.r1cs")], ["Cube"])
    assert o.status == 0 and o.verdict is True
    assert len(o.specials) == K and o.summary.n_rows_reduced == 0
    assert o.summary.rule_hits[8] == K                                   # every special fired once
    assert o.summary.outer_iterations == (K + 1 if rev else 2)


@pytest.mark.gpu
@pytest.mark.parametrize("force_nwg", [0, 2])
def test_gpu_chain_parity(chain_files, force_nwg):
    import ecneproject_amd as E
    from gpu_common import assert_bit_exact
    sub = E.R1CS(str(chain_files / "cube.r1cs"))
    for K in (5, 150):
        for rev in (False, True):
            p = str(chain_files / ("chain_%d_%d.r1cs" % (K, rev)))
            s = E.System(E.R1CS(p))
            s.abstract(sub, "Cube")
            assert len(s.specials()) == K
            g = E.solve_batch([s], force_nwg=force_nwg)[0]
            assert_bit_exact("chain K=%d rev=%s" % (K, rev), g, orc.run(p, [str(chain_files / "cube.r1cs")], ["Cube"]))


# ---- the same for P5 (isZero pairs, :1492-1550): pair i is  a_i * z_i = 1 - y_i ;  a_i * y_i = 0  with
# a_i = y_{i-1} (a_0 = the input). Every firing makes the next pair's A unique.
def _write_iszero_chain(path, K, reverse):
    # variables: 1 one, 2 = input, then y_0 .. y_{K-1}, then z_0 .. z_{K-1}
    y = list(range(3, 3 + K))
    z = list(range(3 + K, 3 + 2 * K))
    a = [2] + y[:-1]
    pairs = [[([(a[i], 1)], [(z[i], 1)], [(1, 1), (y[i], (-1) % P)]), ([(a[i], 1)], [(y[i], 1)], [])] for i in range(K)]
    if reverse:
        pairs.reverse()
    r1cs_py.write(path, 2 * K + 1, 0, 1, 0, [r for pr in pairs for r in pr])


@pytest.fixture(scope="module")
def iszero_files(tmp_path_factory):
    d = tmp_path_factory.mktemp("iszero")
    for K in (5, 150):
        for rev in (False, True):
            _write_iszero_chain(str(d / ("iz_%d_%d.r1cs" % (K, rev))), K, rev)
    return d


@pytest.mark.parametrize("K,rev", [(5, False), (5, True), (150, False), (150, True)])
def test_oracle_iszero_chain(iszero_files, K, rev):
    o = orc.run(str(iszero_files / ("iz_%d_%d.r1cs" % (K, rev))))
    assert o.status == 0
    assert o.summary.rule_hits[12] == K                                  # P5 fired once per pair
    assert o.unique[2:2 + K].all()                                       # every y_i
    assert o.summary.outer_iterations == (K + 1 if rev else 2)


@pytest.mark.gpu
@pytest.mark.parametrize("force_nwg", [0, 2])
def test_gpu_iszero_chain_parity(iszero_files, force_nwg):
    import ecneproject_amd as E
    from gpu_common import assert_bit_exact
    for K in (5, 150):
        for rev in (False, True):
            p = str(iszero_files / ("iz_%d_%d.r1cs" % (K, rev)))
            g = E.solve_batch([E.System(E.R1CS(p))], force_nwg=force_nwg)[0]
            assert_bit_exact("iszero chain K=%d rev=%s" % (K, rev), g, orc.run(p))
