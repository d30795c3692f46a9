"""The committed goldens (tests/golden/fixture_goldens.json) are what the oracle produces today —
guards against silent drift of the checker — and the result is invariant under randomised Set
iteration and initial-queue order on every fixture, so the GPU schedule
question is not masked by a lucky order."""
import hashlib
import json
import os

import numpy as np
import pytest

import fixtures
import orc

HERE = os.path.dirname(os.path.abspath(__file__))


def test_goldens_reproduce():
    with open(os.path.join(HERE, "golden", "fixture_goldens.json")) as f:
        gold = json.load(f)
    assert len(gold) == 93
    for key, exp in gold.items():
        rel, trusted, names, secp = exp["case"]
        o = orc.run(fixtures.path(rel), [fixtures.path(t) for t in trusted], names, secp)
        assert o.status == exp["status"], key
        if o.status:
            continue
        assert o.verdict == exp["verdict"] and list(o.counts()) == exp["counts"], key
        assert hashlib.sha256(np.ascontiguousarray(o.flags).tobytes()).hexdigest() == exp["sha_flags"], key


@pytest.mark.parametrize("rel", [r for r in fixtures.all_r1cs()])
def test_unique_set_is_order_invariant(rel):
    base = orc.run(fixtures.path(rel))
    for seed in (1, 2):
        r = orc.run(fixtures.path(rel), policy=orc.ORDER_RANDOM, seed=seed, shuffle_queue=True)
        assert r.verdict == base.verdict and np.array_equal(r.unique, base.unique), rel
    r = orc.run(fixtures.path(rel), policy=orc.ORDER_ASCENDING)
    assert r.verdict == base.verdict and np.array_equal(r.unique, base.unique), rel


def test_circomlib_suite_totals():
    rels = fixtures.circomlib_suite()
    tot_rows = tot_nnz = 0
    for rel in rels:
        st, d = orc.read_info(fixtures.path(rel))
        tot_rows += d["nConstraints"]
        tot_nnz += sum(d["nnz"])
    assert (len(rels), tot_rows, tot_nnz) == (67, 88599, 289890)
