"""The fast wavefront round (ecneproject_amd/csrc/wave2.hip.hpp, fastrow.hip.hpp) on the rows it learnt in round 2:
constant rows (R3), short binary decompositions, empty pops of other shapes, events with long row lists popped solo,
and the long linear rows whose empty pops are settled from a WATCHED pair of variables kept in the row's record line.

The watched words are a cache that every solve's setup has to reset ("every term unique" would otherwise leak into the
next solve of the same system): each system is therefore solved several times in a row, every result bit-exact against
the oracle (counters included). Both as single-workgroup jobs (state in LDS) and with helpers forced (state in device
memory: the multi-workgroup schedule, where the fast round is the master's tool for narrow frontiers)."""
import os

import pytest

import fixtures
import orc
from gpu_common import assert_bit_exact, build_system

pytestmark = pytest.mark.gpu

# circuits with long rows: 1 025-/33-term multiplexer sums, 254-bit decompositions, constants, isZero / 1 = x + y rows
LONG_ROW_FIXTURES = [
    "ecne_circomlib_tests/Decoder@multiplexer.r1cs",
    "ecne_circomlib_tests/Multiplexer@multiplexer.r1cs",
    "ecne_circomlib_tests/Num2Bits_strict@bitify.r1cs",
    "ecne_circomlib_tests/Bits2Point_Strict@pointbits.r1cs",
    "ecne_circomlib_tests/Pedersen@pedersen.r1cs",
    "ecne_circomlib_tests/EdDSAPoseidonVerifier@eddsaposeidon.r1cs",
]


@pytest.mark.parametrize("force_nwg", [0, 3])
@pytest.mark.parametrize("rel", LONG_ROW_FIXTURES)
def test_same_system_solved_three_times(rel, force_nwg):
    import ecneproject_amd as E
    s = build_system(rel)
    o = orc.run(fixtures.path(rel))
    for rep in range(3):
        g = E.solve_batch([s], force_nwg=force_nwg)[0]
        assert_bit_exact("%s nwg=%d solve %d" % (rel, force_nwg, rep), g, o)


@pytest.mark.parametrize("force_nwg", [0, 4])
def test_ecdsa_like_small_solved_three_times(tmp_path, force_nwg):
    """ecdsa_like(2): multiplexer blocks (1 024 constants, products, a 1 025-term sum per output), the hasPrevNonZero chain of
    1 = x + y rows, two trusted adders -- the shapes the bench workload consists of, at a size the oracle takes a second for"""
    import ecneproject_amd as E
    import ecdsa_like
    p = ecdsa_like.cached(2, 10, directory=str(tmp_path))
    s = build_system(None, ["secp256k1.r1cs"], ["Secp256k1AddUnequal"], path=p)
    o = orc.run(p, [fixtures.path("secp256k1.r1cs")], ["Secp256k1AddUnequal"], True)
    for rep in range(3):
        g = E.solve_batch([s], secp_solve=True, force_nwg=force_nwg)[0]
        assert_bit_exact("ecdsa_like(2) nwg=%d solve %d" % (force_nwg, rep), g, o)
        if force_nwg:
            sd = list(g.summary.sched)
            assert sd[0] > 0, "the fast wavefront round never ran on the multi-workgroup schedule"


def test_general_rounds_only_same_answers(monkeypatch, tmp_path):
    """ECNE_LDS_BYTES=0: no fast-round tables, no chain executor -- the general rounds give the same answers"""
    import ecneproject_amd as E
    import ecdsa_like
    monkeypatch.setenv("ECNE_LDS_BYTES", "0")
    p = ecdsa_like.cached(2, 10, directory=str(tmp_path))
    s = build_system(None, ["secp256k1.r1cs"], ["Secp256k1AddUnequal"], path=p)
    o = orc.run(p, [fixtures.path("secp256k1.r1cs")], ["Secp256k1AddUnequal"], True)
    for force_nwg in (0, 4):
        g = E.solve_batch([s], secp_solve=True, force_nwg=force_nwg)[0]
        assert_bit_exact("ecdsa_like(2) general rounds nwg=%d" % force_nwg, g, o)
