"""The chain executor (ecneproject_amd/csrc/chain.hip.hpp): strictly sequential pops of single-workgroup systems with
the unique / is_known flags and the in_queue tags in LDS (queue_mode=2), against the oracle -- every fixture, the
trusted-function configurations, the seeded fuzz systems (degenerate rows, error statuses, long rows) -- and with
the LDS switched off (ECNE_LDS_BYTES=0: the same answers from the general path)."""
import os

import pytest

import fixtures
import fuzz_r1cs
import orc
from gpu_common import assert_bit_exact, build_system

pytestmark = pytest.mark.gpu

TRUSTED = [
    ("tornadocash_circuits/commitHasher.r1cs", fixtures.PED, fixtures.PED_NAMES, False),
    ("tornadocash_circuits/withdraw.r1cs", fixtures.PED, fixtures.PED_NAMES, False),
    ("secp256k1.r1cs", ["bigmultmodp.r1cs", "biglessthan.r1cs"], ["BigMultModP", "BigLessThan"], True),
]


@pytest.mark.parametrize("rel", fixtures.all_r1cs())
def test_chain_mode_every_fixture(rel):
    import ecneproject_amd as E
    g = E.solve_batch([build_system(rel)], queue_mode=2)[0]
    assert_bit_exact(rel, g, orc.run(fixtures.path(rel)))


@pytest.mark.parametrize("rel,trusted,names,secp", TRUSTED, ids=[c[0] for c in TRUSTED])
def test_chain_mode_trusted(rel, trusted, names, secp):
    import ecneproject_amd as E
    g = E.solve_batch([build_system(rel, trusted, names)], secp_solve=secp, queue_mode=2)[0]
    assert_bit_exact(rel, g, orc.run(fixtures.path(rel), [fixtures.path(t) for t in trusted], names, secp))


def test_chain_mode_fuzz(tmp_path):
    import ecneproject_amd as E
    paths = []
    for seed in range(300):
        p = str(tmp_path / ("%d.r1cs" % seed))
        fuzz_r1cs.write(p, fuzz_r1cs.make(seed))
        paths.append(p)
    for seed in range(60):
        p = str(tmp_path / ("w%d.r1cs" % seed))
        fuzz_r1cs.write(p, fuzz_r1cs.make_wide(seed))
        paths.append(p)
    systems = [E.System(E.R1CS(p)) for p in paths]
    res = []
    for i in range(0, len(systems), 120):
        res += E.solve_batch(systems[i:i + 120], queue_mode=2)
    for p, g in zip(paths, res):
        assert_bit_exact(os.path.basename(p), g, orc.run(p))


def test_without_lds_same_answers(monkeypatch):
    """ECNE_LDS_BYTES=0 keeps all state in device memory (no chain executor): same results"""
    import ecneproject_amd as E
    monkeypatch.setenv("ECNE_LDS_BYTES", "0")
    for rel in ("ecne_circomlib_tests/Poseidon@poseidon.r1cs", "ecne_circomlib_tests/BabyPbk@babyjub.r1cs", "target/division.r1cs"):
        for mode in (0, 2):
            g = E.solve_batch([build_system(rel)], queue_mode=mode)[0]
            assert_bit_exact(rel, g, orc.run(fixtures.path(rel)))
