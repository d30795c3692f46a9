"""BASELINE.json config 5 on the GPU: the synthetic ecdsa_like(S) circuit (tests/ecdsa_like.py) with
secp256k1.r1cs trusted.  Small S: bit-exact against the oracle.  Full size (S = 26, 1.09 M rows):
size-independent properties and the whole per-variable state against the oracle (~20 s of CPU; ECNE_FULL_ORACLE=0
skips that part; ECNE_FULL_ORACLE=2 adds ecdsa_like(104), 4.4 M rows, minutes of oracle)."""
import os

import numpy as np
import pytest

import ecneproject_amd as E
import ecdsa_like
import fixtures
import orc
from gpu_common import assert_bit_exact, build_system

pytestmark = pytest.mark.gpu
TRUSTED = (["secp256k1.r1cs"], ["Secp256k1AddUnequal"])


@pytest.mark.parametrize("S,stride", [(2, 2), (3, 3), (4, 4), (5, 6)])
def test_small_ecdsa_like_bit_exact(S, stride):
    path = ecdsa_like.cached(S, stride)
    s = build_system(None, *TRUSTED, path=path)
    assert len(s.specials()) == S - 1
    g = E.solve_batch([s])[0]
    o = orc.run(path, [fixtures.path("secp256k1.r1cs")], TRUSTED[1])
    assert o.verdict is True and o.summary.outer_iterations == S + 2
    assert_bit_exact("ecdsa_like(%d,%d)" % (S, stride), g, o)


@pytest.mark.parametrize("S,stride", [(3, 7), (4, 8), (3, 9)])
@pytest.mark.parametrize("force_nwg", [0, 3, 16])
def test_long_rows_ride_along_bit_exact(S, stride, force_nwg):
    """Sum rows of 129 / 257 / 513 terms (8 per stride): they are marked, checked and executed inside the
    queue rounds by whole workgroups, on one workgroup and on forced teams of 3 and 16."""
    path = ecdsa_like.cached(S, stride)
    s = build_system(None, *TRUSTED, path=path)
    g = E.solve_batch([s], force_nwg=force_nwg)[0]
    o = orc.run(path, [fixtures.path("secp256k1.r1cs")], TRUSTED[1])
    assert o.verdict is True
    assert_bit_exact("ecdsa_like(%d,%d) nwg=%d" % (S, stride, force_nwg), g, o)


def test_full_size_properties():
    """ecdsa_like(26): 25 adders abstracted, chained one per outer iteration (P1 fires one special per
    iteration), every variable the circuit mentions resolved, idempotent."""
    path = ecdsa_like.cached(26, 10)
    s = build_system(None, *TRUSTED, path=path)
    info = s.info
    assert info.n_rows_main == 1092639 and info.n_rows == 694264 and info.n_specials == 25
    g = E.solve_batch([s])[0]
    assert g.status == 0 and g.function_good
    assert g.summary.outer_iterations == 28
    assert list(g.counts()) == [694285, 694311, 6, 6]
    assert g.summary.rule_hits[8] == 25            # P1: each special fired exactly once
    assert g.summary.rule_hits[7] == 52            # R8: one all-but-one-zero group per decoder
    assert len(g.bad_rows) == 26                   # the 26 IsZero rows holding the `inv` hint (never determined)
    # every unique variable is also known; bounds are ordered
    assert not np.any(g.unique & ~g.is_known)
    g2 = E.solve_batch([s])[0]
    assert np.array_equal(g.flags, g2.flags) and g.summary.pops == g2.summary.pops
    if os.environ.get("ECNE_FULL_ORACLE", "1") != "0":      # the whole state against the oracle, ~20 s of CPU (ECNE_FULL_ORACLE=0 skips it)
        o = orc.run(path, [fixtures.path("secp256k1.r1cs")], TRUSTED[1])
        assert_bit_exact("ecdsa_like(26,10)", g, o)
        # (pins tests/golden/scale_goldens.json -- what bench.py checks its own state against -- to the oracle on every run of the suite)
        from state_digest import numpy_digest
        assert ["%016x" % x for x in numpy_digest(o)] == _golden("ecdsa_like(26,10)+Secp256k1AddUnequal")["digest"]
        gd = E.solve_batch([s], fetch_states="digest")[0]
        _assert_matches_golden("ecdsa_like(26,10)", gd, _golden("ecdsa_like(26,10)+Secp256k1AddUnequal"))


@pytest.mark.parametrize("force_nwg", [0, 43])
def test_mid_size_bit_exact(force_nwg):
    """ecdsa_like(6, 10): 160 k reduced rows with the full-size 1 025-term rows, on the default and on the
    bench's workgroup count -- the largest case the suite compares state by state (oracle ~2 s)."""
    path = ecdsa_like.cached(6, 10)
    s = build_system(None, *TRUSTED, path=path)
    g = E.solve_batch([s], force_nwg=force_nwg)[0]
    o = orc.run(path, [fixtures.path("secp256k1.r1cs")], TRUSTED[1])
    assert_bit_exact("ecdsa_like(6,10) nwg=%d" % force_nwg, g, o)


def _golden(key):
    import json
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "scale_goldens.json")) as f:
        return json.load(f)[key]


def _assert_matches_golden(tag, g, gold):
    """g: a SolveResult fetched with the device-side digest; gold: the oracle's entry of tests/golden/scale_goldens.json"""
    s = g.summary
    assert g.status == gold["status"] == 0 and g.function_good == gold["verdict"], tag
    assert [int(x) for x in g.counts()] == gold["counts"], tag
    assert (int(s.pops), int(s.successful_steps), int(s.num_unique), int(s.outer_iterations)) == (gold["pops"], gold["steps"], gold["num_unique"], gold["outer"]), tag
    assert [int(x) for x in list(s.rule_hits)[:13]] == gold["rule_hits"], tag
    assert ["%016x" % g.digest[0], "%016x" % g.digest[1]] == gold["digest"], tag      # flags, bounds, tags, values of every variable


def test_scale_out_416_full_state_parity():
    """ecdsa_like(416): 17.6 M rows, a 2.3 GB file -- the last point of SURVEY.md 8(d)'s scale-out series {26, 104, 416} -- through the device
    front-end: counters (418 outer iterations, 25.6 M pops) and the digest of the WHOLE per-variable state against the oracle's, committed by
    tests/golden/make_scale_goldens.py (90 minutes and 45 GB of oracle: vectors, not a live oracle run)."""
    gold = _golden("ecdsa_like(416,10)+Secp256k1AddUnequal")
    path = ecdsa_like.cached(416, 10)
    s = build_system(None, *TRUSTED, path=path)
    g = E.solve_batch([s], fetch_states="digest")[0]
    _assert_matches_golden("ecdsa_like(416,10)", g, gold)
    assert int(g.summary.outer_iterations) == 418 and int(g.summary.pops) == 25633135
    del s, g


def test_scale_out_full_state_parity():
    """ecdsa_like(104): 4.4 M rows -- the only case beyond the 256 MiB Infinity Cache -- through the device front-end, and one million-row
    input of a different shape (1 400 Poseidon copies side by side): counters and the WHOLE per-variable state against the oracle's, through
    the digest the engine computes on the device (ecne_result_digest; tests/test_gpu_soak.py checks the digest against its numpy
    restatement) and the oracle's digest committed by tests/golden/make_scale_goldens.py. ECNE_FULL_ORACLE=2 also runs the oracle itself
    (3-7 minutes) and compares array by array."""
    import multi_copy
    from state_digest import numpy_digest
    path = ecdsa_like.cached(104, 10)
    s = build_system(None, *TRUSTED, path=path)
    assert E.frontend_stats()["layout_device"] in (0.0, 1.0)
    full = os.environ.get("ECNE_FULL_ORACLE", "1") == "2"
    g = E.solve_batch([s], fetch_states="both" if full else "digest")[0]
    _assert_matches_golden("ecdsa_like(104,10)", g, _golden("ecdsa_like(104,10)+Secp256k1AddUnequal"))
    p2 = multi_copy.cached("ecne_circomlib_tests/Poseidon@poseidon.r1cs", 1400)
    s2 = build_system(None, path=p2)
    g2 = E.solve_batch([s2], fetch_states="both" if full else "digest")[0]
    _assert_matches_golden("1400 x Poseidon", g2, _golden("1400xPoseidon@poseidon"))
    if full:
        o = orc.run(path, [fixtures.path("secp256k1.r1cs")], TRUSTED[1])
        assert o.verdict is True
        assert_bit_exact("ecdsa_like(104,10)", g, o)
        assert ["%016x" % x for x in numpy_digest(o)] == _golden("ecdsa_like(104,10)+Secp256k1AddUnequal")["digest"]
        assert_bit_exact("1400 x Poseidon", g2, orc.run(p2))
