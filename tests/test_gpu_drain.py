"""Drain rounds (csrc/drain.hip.hpp): a window of the queue executed in dataflow order by all workgroups of a job, pushes resolved
once per window -- the schedule large systems take for their wide frontiers. The tests force several workgroups on small systems
and run the three schedules side by side: queue_mode 0 (default: drain rounds from 256 queued rows on), 3 (prefix rounds, round 2's
schedule) and 4 (test hook: every frontier of two rows and more is drained, so the level logic meets dependency chains, narrow
windows, long rows at every rank, error pops). All of them must give the oracle's state bit for bit."""
import os

import numpy as np
import pytest

import ecneproject_amd as E
import ecdsa_like
import fixtures
import fuzz_r1cs
import orc
from gpu_common import assert_bit_exact, build_system

pytestmark = pytest.mark.gpu
MODES = (0, 3, 4)


@pytest.fixture(scope="module")
def fuzz_paths(tmp_path_factory):
    d = tmp_path_factory.mktemp("drainfuzz")
    paths = []
    for seed in range(7000, 7160):
        p = str(d / ("%d.r1cs" % seed))
        fuzz_r1cs.write(p, fuzz_r1cs.make_wide(seed, 1 + seed % 3) if seed % 4 else fuzz_r1cs.make(seed))
        paths.append(p)
    return paths


@pytest.mark.parametrize("nwg", [2, 7, 24])      # 24: narrow frontiers go to a sub-team of 8 workgroups (rounds.hip.hpp multi_chain), error pops included
def test_fuzz_families_all_schedules(fuzz_paths, nwg):
    oracles = [orc.run(p) for p in fuzz_paths]
    systems = [E.System(E.R1CS(p)) for p in fuzz_paths]
    for mode in MODES:
        res = []
        for i in range(0, len(systems), 80):
            res += E.solve_batch(systems[i:i + 80], force_nwg=nwg, queue_mode=mode)
        for p, g, o in zip(fuzz_paths, res, oracles):
            assert_bit_exact("%s nwg=%d mode=%d" % (os.path.basename(p), nwg, mode), g, o)


@pytest.mark.parametrize("S,stride", [(2, 10), (3, 7), (4, 8)])
def test_ecdsa_like_all_schedules(S, stride):
    """multiplexer blocks one behind the other in the FIFO (decoder sum, its dependents, next decoder sum ...): what drain
    rounds are for; 513- and 1 025-term rows at every rank of a window"""
    path = ecdsa_like.cached(S, stride)
    s = build_system(None, ["secp256k1.r1cs"], ["Secp256k1AddUnequal"], path=path)
    o = orc.run(path, [fixtures.path("secp256k1.r1cs")], ["Secp256k1AddUnequal"])
    for nwg in (3, 16, 40):
        for mode in MODES:
            g = E.solve_batch([s], force_nwg=nwg, queue_mode=mode)[0]
            assert_bit_exact("ecdsa_like(%d,%d) nwg=%d mode=%d" % (S, stride, nwg, mode), g, o)
            if mode != 3:
                assert g.summary.multi_ms[7] > 0, "no drain round ran"


@pytest.mark.parametrize("rel,trusted,names", [
    ("secp256k1.r1cs", ["bigmultmodp.r1cs", "biglessthan.r1cs"], ["BigMultModP", "BigLessThan"]),
    ("tornadocash_circuits/withdraw.r1cs", fixtures.PED, fixtures.PED_NAMES),
    ("ecne_circomlib_tests/EdDSAMiMCSpongeVerifier@eddsamimcsponge.r1cs", [], []),
])
def test_reference_configs_all_schedules(rel, trusted, names):
    s = build_system(rel, trusted, names)
    o = orc.run(fixtures.path(rel), [fixtures.path(t) for t in trusted], names, True)
    for mode in MODES:
        g = E.solve_batch([s], secp_solve=True, force_nwg=5, queue_mode=mode)[0]
        assert_bit_exact("%s mode=%d" % (rel, mode), g, o)


def test_drain_is_the_default_for_large_systems():
    """ecdsa_like(6): 160 k reduced rows get a team of workgroups and drain rounds without being asked; far fewer rounds than
    the prefix schedule needs for the same pops"""
    path = ecdsa_like.cached(6, 10)
    s = build_system(None, ["secp256k1.r1cs"], ["Secp256k1AddUnequal"], path=path)
    g0 = E.solve_batch([s], fetch_states=False)[0]
    g3 = E.solve_batch([s], fetch_states=False, queue_mode=3)[0]
    assert g0.summary.multi_ms[7] > 0 and g3.summary.multi_ms[7] == 0
    assert g0.summary.pops == g3.summary.pops and list(g0.counts()) == list(g3.counts())
    assert g0.summary.rule_hits[13] < g3.summary.rule_hits[13]


@pytest.mark.parametrize("mode", [0, 4])
def test_every_reference_fixture_on_teams(mode):
    """all 89 .r1cs files of the reference, each on a team of six workgroups (so that rounds on teams, sub-team logic and -- mode 4
    -- drain rounds on every frontier meet real circuits: long rows of every shape, error statuses, empty files), batched"""
    rels = fixtures.all_r1cs()
    systems = [build_system(r) for r in rels]
    res = []
    for i in range(0, len(systems), 30):
        res += E.solve_batch(systems[i:i + 30], force_nwg=6, queue_mode=mode)
    for r, g in zip(rels, res):
        assert_bit_exact("%s nwg=6 mode=%d" % (r, mode), g, orc.run(fixtures.path(r)))


def test_sweeps_that_end_early_agree_across_workgroups(tmp_path):
    """P3 ends at its first barrier when nobody reported a candidate, P4 when no candidate is untagged (k_solve.hip.hpp): every
    workgroup decides that by itself from words published before the barrier, so nobody may change those words before the next one.
    The first version let the master re-arm P3's candidate word while slower workgroups were still reading it -- they left the pass,
    the others went on to its second barrier: one wrong state in eight runs of this batch (tests/tools/stress_fuzz.py 92000 60 4,
    seed 92002 on five workgroups, the only system of the batch whose P3 fires). Repeated, on teams of 5 and 8, every frontier drained."""
    paths = []
    for seed in range(92000, 92024):
        p = str(tmp_path / ("%d.r1cs" % seed))
        fuzz_r1cs.write(p, fuzz_r1cs.make_wide(seed, 4))
        paths.append(p)
    oracles = [orc.run(p) for p in paths]
    assert any(o.summary.rule_hits[10] for o in oracles)          # P3 fires somewhere in the batch
    systems = [E.System(E.R1CS(p)) for p in paths]
    for rep in range(6):
        for nwg in (5, 8):
            res = E.solve_batch(systems, force_nwg=nwg, queue_mode=4 if rep % 2 else 0)
            for p, g, o in zip(paths, res, oracles):
                assert_bit_exact("%s nwg=%d rep %d" % (os.path.basename(p), nwg, rep), g, o)
