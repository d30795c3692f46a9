"""abstraction's candidate scan on the device (ecneproject_amd/csrc/abstract.hip.hpp: row fingerprints, weighted prefix scan,
window test) against the host scan and the oracle: specials, reduced rows and statuses must be identical whichever finds the
candidate windows -- on the 300 seeded abstraction-fuzz cases (overlapping windows, near-copies, shared variables, KeyError),
the reference's trusted-function configurations and ecdsa_like(26) (1.09 M rows x a 15 935-row pattern)."""
import numpy as np
import pytest

import fixtures
import orc
from test_abstraction_fuzz import N_CASES, abs_dir      # noqa: F401  (the module-scoped fixture with the generated files)

pytestmark = pytest.mark.gpu


def _abstract(E, monkeypatch, main, subs, names, device):
    monkeypatch.setenv("ECNE_ABSTRACT_DEVICE", "1" if device else "0")
    s = E.System(E.R1CS(main))
    fl = sorted(((n, E.R1CS(p)) for p, n in zip(subs, names)), key=lambda x: -len(x[1]))
    st, used = 0, []
    try:
        for n, f in fl:
            s.abstract(f, n)
            used.append(E.System.last_abstract_stats()["device"])
    except E.EcneError as e:
        st = e.status
    return s, st, used


def _same_rows(a, b):
    for part in range(3):
        ra, rb = a.rows(part), b.rows(part)
        for x, y in zip(ra, rb):
            if not np.array_equal(x, y):
                return False
    return True


def test_device_scan_on_abstraction_fuzz(abs_dir, monkeypatch):      # noqa: F811
    import ecneproject_amd as E
    n_inst = 0
    for seed in range(N_CASES):
        mp, sp = str(abs_dir / ("main%d.r1cs" % seed)), str(abs_dir / ("sub%d.r1cs" % seed))
        o = orc.run(mp, [sp], ["T"], want_states=False)
        d, st_d, used = _abstract(E, monkeypatch, mp, [sp], ["T"], True)
        h, st_h, _ = _abstract(E, monkeypatch, mp, [sp], ["T"], False)
        assert used == [True] or st_d != 0, seed
        assert st_d == st_h == (o.status if o.status == -5 else 0), (seed, st_d, st_h, o.status)
        if st_d:
            continue
        assert d.specials() == h.specials() == o.specials, seed
        assert len(d) == len(h) == o.summary.n_rows_reduced and _same_rows(d, h), seed
        n_inst += len(o.specials)
    assert n_inst > 150


CASES = [
    ("secp256k1.r1cs", ["bigmultmodp.r1cs", "biglessthan.r1cs"], ["BigMultModP", "BigLessThan"]),
    ("tornadocash_circuits/withdraw.r1cs", fixtures.PED, fixtures.PED_NAMES),
    ("tornadocash_circuits/commitHasher.r1cs", fixtures.PED, fixtures.PED_NAMES),
]


@pytest.mark.parametrize("rel,trusted,names", CASES, ids=[c[0] for c in CASES])
def test_device_scan_reference_configs(rel, trusted, names, monkeypatch):
    import ecneproject_amd as E
    paths = [fixtures.path(t) for t in trusted]
    o = orc.run(fixtures.path(rel), paths, names, True, want_states=False)
    d, st_d, used = _abstract(E, monkeypatch, fixtures.path(rel), paths, names, True)
    h, st_h, _ = _abstract(E, monkeypatch, fixtures.path(rel), paths, names, False)
    assert st_d == st_h == 0 and all(used)
    assert d.specials() == h.specials() == o.specials and len(d) == len(h) == o.summary.n_rows_reduced and _same_rows(d, h)


def test_device_scan_ecdsa_like_full_size(monkeypatch):
    """BASELINE.json config 5: 25 copies of the 15 935-row adder in 1.09 M rows; the default picks the device here"""
    import ecdsa_like
    import ecneproject_amd as E
    monkeypatch.delenv("ECNE_ABSTRACT_DEVICE", raising=False)
    path = ecdsa_like.cached(26, 10)
    s = E.System(E.R1CS(path))
    s.abstract(E.R1CS(fixtures.path("secp256k1.r1cs")), "Secp256k1AddUnequal")
    st = E.System.last_abstract_stats()
    assert st["device"] and st["candidates"] >= 25 and st["fingerprint_ms"] > 0
    h, st_h, _ = _abstract(E, monkeypatch, path, [fixtures.path("secp256k1.r1cs")], ["Secp256k1AddUnequal"], False)
    assert st_h == 0 and len(s) == len(h) == 694264 and s.specials() == h.specials() and len(s.specials()) == 25
    assert _same_rows(s, h)
    print("abstraction on device: fingerprint kernel %.3f ms over %.1f MB = %.0f GB/s, scan + window test %.3f ms, upload %.1f ms, %d candidates"
          % (st["fingerprint_ms"], st["bytes"] / 1e6, st["bytes"] / max(st["fingerprint_ms"], 1e-9) / 1e6, st["scan_ms"], st["upload_ms"], st["candidates"]))
