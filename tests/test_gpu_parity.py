"""GPU parity tests proper: the HIP engine (through the C ABI) against the CPU oracle, bit-exact on
the per-variable state (unique, is_known, lb, ub, abz, values), the printed counts, the verdict,
the bad-constraint list and every schedule counter (pops, successful_steps, per-rule hits).

Covers BASELINE.json configs 1-4 and every other .r1cs of the reference tree; config 5 is in
test_gpu_ecdsa_like.py."""
import hashlib
import json
import os

import numpy as np
import pytest

import ecneproject_amd as E
import fixtures
import orc
from gpu_common import assert_bit_exact, build_system

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))

TRUSTED_CASES = [
    ("tornadocash_circuits/commitHasher.r1cs", fixtures.PED, fixtures.PED_NAMES, False),
    ("tornadocash_circuits/withdraw.r1cs", fixtures.PED, fixtures.PED_NAMES, False),
    ("secp256k1.r1cs", ["bigmultmodp.r1cs", "biglessthan.r1cs"], ["BigMultModP", "BigLessThan"], True),   # config 3
]


def test_library_loaded_is_in_tree():
    from ecneproject_amd import _lib
    assert os.path.dirname(_lib.SO) == os.path.dirname(os.path.abspath(E.__file__))
    assert E.device_count() >= 1


@pytest.mark.parametrize("rel", fixtures.all_r1cs())
def test_every_reference_fixture(rel):
    """configs 1, 2 and each file of config 4, one solve per file."""
    g = E.solve_batch([build_system(rel)])[0]
    o = orc.run(fixtures.path(rel))
    assert_bit_exact(rel, g, o)


@pytest.mark.parametrize("rel,trusted,names,secp", TRUSTED_CASES, ids=[c[0] for c in TRUSTED_CASES])
def test_trusted_function_configs(rel, trusted, names, secp):
    g = E.solve_batch([build_system(rel, trusted, names)], secp_solve=secp)[0]
    o = orc.run(fixtures.path(rel), [fixtures.path(t) for t in trusted], names, secp)
    assert o.status == 0 and o.verdict is True      # test/runtests.jl:25,30,35
    assert_bit_exact(rel, g, o)


def test_secp_without_secp_solve_is_undefvar():
    """reference :762 reads `dsu`, defined only under secp_solve=true"""
    s = build_system("secp256k1.r1cs", ["bigmultmodp.r1cs", "biglessthan.r1cs"], ["BigMultModP", "BigLessThan"])
    g = E.solve_batch([s], secp_solve=False)[0]
    assert g.status == -4
    with pytest.raises(E.UndefVarError):
        g.raise_for_status()
    # same with helper workgroups: an error raised before the queue phase must not strand them at a
    # barrier (regression: a one-barrier mismatch on this path made the helpers spin until their bound)
    import time
    t = time.time()
    g = E.solve_batch([s], secp_solve=False, force_nwg=4)[0]
    assert g.status == -4 and time.time() - t < 5.0


@pytest.mark.parametrize("force_nwg", [2, 5])
def test_helper_workgroups_do_not_change_results(force_nwg):
    """multi-workgroup sweeps and multi-workgroup queue rounds on mid-size circuits"""
    for rel, trusted, names, secp in [("secp256k1.r1cs", [], [], False), TRUSTED_CASES[2],
                                      ("ecne_circomlib_tests/EdDSAMiMCSpongeVerifier@eddsamimcsponge.r1cs", [], [], False),
                                      ("bigmultmodp.r1cs", [], [], False), ("poseidon.r1cs", [], [], False)]:
        g = E.solve_batch([build_system(rel, trusted, names)], secp_solve=secp, force_nwg=force_nwg)[0]
        o = orc.run(fixtures.path(rel), [fixtures.path(t) for t in trusted], names, secp)
        assert_bit_exact("%s nwg=%d" % (rel, force_nwg), g, o)


def test_config4_suite_as_one_batch():
    """The 67 circomlib files as ONE launch (one workgroup per file): identical to solving them
    one at a time, and to the oracle."""
    rels = fixtures.circomlib_suite()
    assert len(rels) == 67
    systems = [build_system(r) for r in rels]
    res = E.solve_batch(systems)
    for rel, g in zip(rels, res):
        assert_bit_exact(rel, g, orc.run(fixtures.path(rel)))


def test_resolve_is_idempotent():
    """solving the same resident system twice gives the same answer (state is re-initialised)"""
    s = build_system("ecne_circomlib_tests/EdDSAPoseidonVerifier@eddsaposeidon.r1cs")
    a = E.solve_batch([s])[0]
    b = E.solve_batch([s])[0]
    assert np.array_equal(a.flags, b.flags) and np.array_equal(a.ub, b.ub)
    assert a.summary.pops == b.summary.pops


def test_committed_goldens():
    """tests/golden/fixture_goldens.json (written by tests/golden/make_fixture_goldens.py from the
    oracle): verdict, counts and SHA-256 of the per-variable arrays."""
    with open(os.path.join(HERE, "golden", "fixture_goldens.json")) as f:
        gold = json.load(f)
    for key, exp in gold.items():
        rel, trusted, names, secp = exp["case"]
        g = E.solve_batch([build_system(rel, trusted, names)], secp_solve=secp)[0]
        assert g.status == exp["status"], key
        if exp["status"] != 0:
            continue
        assert g.function_good == exp["verdict"], key
        assert list(g.counts()) == exp["counts"], key
        assert hashlib.sha256(np.ascontiguousarray(g.flags).tobytes()).hexdigest() == exp["sha_flags"], key
        state = np.ascontiguousarray(g.lb).tobytes() + np.ascontiguousarray(g.ub).tobytes() + \
            np.ascontiguousarray(g.abz.astype(np.int64)).tobytes()
        assert hashlib.sha256(state).hexdigest() == exp["sha_bounds_abz"], key


def test_readr1cs_csr_view():
    f, kn, out, nv = E.readR1CS(fixtures.path("target/division.r1cs"))
    assert (kn, out, nv) == ([1, 3, 4, 5, 6], [2], 8)
    rp, col, coef = f.csr(2)
    assert rp.tolist() == [0, 3, 4, 7] and sorted(col[:3].tolist()) == [3, 4, 7]
