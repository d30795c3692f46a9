"""Run-to-run determinism inside `-m gpu`: the schedule decides WHEN a row is popped, never what the pops leave behind, so the same
resident system solved again and again must reproduce the same per-variable state and the same counters -- on the configuration
the bench runs (ecdsa_like(26) on its default team of 170 workgroups) and on the stress batch that once exposed a 1-in-8
cross-workgroup race (tests/test_gpu_drain.py::test_sweeps_that_end_early_agree_across_workgroups). The state is compared through
ecne_result_digest (a digest computed on the device: no 150 MB download per solve); the digest itself is checked against a numpy
restatement on fetched states, and the first solve of every soak against the oracle."""
import os

import numpy as np
import pytest

import ecneproject_amd as E
import fixtures
import fuzz_r1cs
import orc
from gpu_common import assert_bit_exact, build_system

pytestmark = pytest.mark.gpu


from state_digest import numpy_digest  # noqa: E402  (include/ecne.h, ecne_result_digest restated on fetched arrays)


def _counters(g):
    s = g.summary
    return (g.status, g.function_good, tuple(g.counts()), s.pops, s.successful_steps, s.num_unique, s.outer_iterations, tuple(s.rule_hits[:13]))


def test_digest_is_the_state():
    """the device digest equals the numpy restatement on real states, and moves when one flag / bound / tag / value moves"""
    cases = [("secp256k1.r1cs", ["bigmultmodp.r1cs", "biglessthan.r1cs"], ["BigMultModP", "BigLessThan"], True),
             ("ecne_circomlib_tests/Poseidon@poseidon.r1cs", [], [], False), ("target/division.r1cs", [], [], False),
             ("ecne_circomlib_tests/Decoder@multiplexer.r1cs", [], [], False), ("ecne_circomlib_tests/Num2Bits@bitify.r1cs", [], [], False)]
    seen = set()
    for rel, trusted, names, secp in cases:
        s = build_system(rel, trusted, names)
        g = E.solve_batch([s], secp_solve=secp, fetch_states="both")[0]
        assert g.digest == numpy_digest(g), rel
        seen.add(g.digest)
        for field in ("flags", "abz", "lb", "ub"):       # the restatement is sensitive to every array (so the kernel's sum is)
            a = getattr(g, field)
            if a.size == 0:
                continue
            keep = a.copy()
            a.flat[a.size // 2] ^= 1
            assert numpy_digest(g) != g.digest, (rel, field)
            a[...] = keep
    assert len(seen) == len(cases)


def test_soak_ecdsa_like_on_the_default_team():
    """30 solves of BASELINE config 5 on the team the bench uses: one digest, one set of counters"""
    import ecdsa_like
    s = build_system(None, ["secp256k1.r1cs"], ["Secp256k1AddUnequal"], path=ecdsa_like.cached(26, 10))
    first = E.solve_batch([s], fetch_states="both")[0]
    assert first.status == 0 and first.function_good and first.digest == numpy_digest(first)
    assert list(first.counts()) == [694285, 694311, 6, 6] and first.summary.outer_iterations == 28      # (the oracle's, test_gpu_ecdsa_like.py)
    ref = _counters(first)
    for rep in range(30):
        g = E.solve_batch([s], fetch_states="digest")[0]
        assert g.digest == first.digest, "solve %d left another state" % rep
        assert _counters(g) == ref, "solve %d: other counters" % rep


def test_soak_stress_batch_on_teams(tmp_path):
    """the 92000-seed batch (wide random systems with long rows, P3 firing in one of them), 30 passes on teams of 5 and 8 workgroups
    and with every frontier drained: the first pass against the oracle, every other pass against the first by digest"""
    paths = []
    for seed in range(92000, 92016):
        p = str(tmp_path / ("%d.r1cs" % seed))
        fuzz_r1cs.write(p, fuzz_r1cs.make_wide(seed, 4))
        paths.append(p)
    systems = [E.System(E.R1CS(p)) for p in paths]
    firsts = E.solve_batch(systems, force_nwg=5, fetch_states="both")
    for p, g in zip(paths, firsts):
        assert_bit_exact("soak batch %s" % os.path.basename(p), g, orc.run(p))
        assert g.digest == numpy_digest(g)
    want = [(g.digest, _counters(g)) for g in firsts]
    for rep in range(30):
        nwg, mode = (5, 8)[rep % 2], (0, 4)[(rep // 2) % 2]
        res = E.solve_batch(systems, force_nwg=nwg, queue_mode=mode, fetch_states="digest")
        got = [(g.digest, _counters(g)) for g in res]
        assert got == want, ("pass %d (nwg %d, mode %d)" % (rep, nwg, mode), [os.path.basename(p) for p, a, b in zip(paths, got, want) if a != b])


def test_long_stretch_without_a_barrier_is_not_a_timeout(monkeypatch):
    """The job barrier's bound is on time WITHOUT PROGRESS (job_heartbeat): with the bound cut to 3 ms, solves whose master works
    alone for far longer than that -- the chain executor's circuit on a forced team (EdDSAMiMCSponge: ~20 ms of sequential pops while
    the helpers wait), drain rounds over a window of dependent rows -- still end normally, bit-exact."""
    monkeypatch.setenv("ECNE_BARRIER_TIMEOUT_MS", "3")
    for rel, mode in (("ecne_circomlib_tests/EdDSAMiMCSpongeVerifier@eddsamimcsponge.r1cs", 0), ("ecne_circomlib_tests/EdDSAMiMCSpongeVerifier@eddsamimcsponge.r1cs", 4),
                      ("ecne_circomlib_tests/Poseidon@poseidon.r1cs", 4)):
        s = build_system(rel)
        g = E.solve_batch([s], force_nwg=6, queue_mode=mode)[0]
        assert g.summary.device_ms > 3.0 or "Poseidon" in rel
        assert_bit_exact("%s with a 3 ms barrier bound" % rel, g, orc.run(fixtures.path(rel)))
