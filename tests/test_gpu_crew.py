"""Crew rounds (ecneproject_amd/csrc/crew.hip.hpp): the narrow dependency levels of a single-workgroup job with one wavefront per queued
row. The schedule decides WHEN a row is popped, never what the pops leave behind -- so the chained circuits the crew rounds were built
for must come out bit for bit as the oracle's sequential pops (/root/reference/src/R1CSConstraintSolver.jl:805-1349) leave them, with
the same counters, (a) with crew rounds (the default), (b) in a second process with ECNE_CREW=0 (level rounds only), (c) with the long
decompositions' "every term unique" shortcut off (ECNE_R4DONE=0), and (d) again and again on the resident systems, alone and as one batch
launch (a soak for cross-wavefront races in the crew's LDS protocol: write marks by round parity, published read sets, candidate lists)."""
import json
import os
import subprocess
import sys

import pytest

import ecneproject_amd as E
import fixtures
import orc
from gpu_common import assert_bit_exact, build_system

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))

CASES = [("ecne_circomlib_tests/Poseidon@poseidon.r1cs", [], []),
         ("ecne_circomlib_tests/EdDSAMiMCSpongeVerifier@eddsamimcsponge.r1cs", [], []),
         ("ecne_circomlib_tests/EdDSAPoseidonVerifier@eddsaposeidon.r1cs", [], []),
         ("ecne_circomlib_tests/BabyPbk@babyjub.r1cs", [], []),
         ("ecne_circomlib_tests/MiMCSponge@mimcsponge.r1cs", [], []),
         ("ecne_circomlib_tests/Pedersen@pedersen_old.r1cs", [], []),
         ("secp256k1.r1cs", ["bigmultmodp.r1cs", "biglessthan.r1cs"], ["BigMultModP", "BigLessThan"]),
         ("tornadocash_circuits/withdraw.r1cs", fixtures.PED, fixtures.PED_NAMES)]


def _counters(g):
    s = g.summary
    return [g.status, bool(g.function_good), list(g.counts()), s.pops, s.successful_steps, s.num_unique, s.outer_iterations, list(s.rule_hits[:13])]


def _solve_all():
    systems = [build_system(rel, tr, nm) for rel, tr, nm in CASES]
    return systems, E.solve_batch(systems, secp_solve=True, fetch_states="both")


def test_chained_circuits_on_crew_rounds_match_the_oracle():
    systems, res = _solve_all()
    for (rel, tr, nm), g in zip(CASES, res):
        o = orc.run(fixtures.path(rel), [fixtures.path(t) for t in tr], nm, True)
        assert_bit_exact(rel, g, o)
    # the rounds were really taken level by level (rule_hits[13] counts rounds of every kind): far fewer rounds than pops on the chains
    pos = res[0].summary
    assert 300 <= pos.rule_hits[13] <= 600 and pos.pops == 1774, (pos.rule_hits[13], pos.pops)


_CHILD = r"""
import json, os, sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import test_gpu_crew as T
systems, res = T._solve_all()
print("RESULT " + json.dumps([[list(g.digest), T._counters(g)] for g in res]))
"""


@pytest.mark.parametrize("env", [{"ECNE_CREW": "0"}, {"ECNE_R4DONE": "0"}, {"ECNE_CREW": "0", "ECNE_LEVEL": "0"}])
def test_same_state_and_counters_without_crew_rounds(env):
    """the switches are read once per process: a second process solves the same circuits without crew rounds / without the shortcut for
    long decompositions / without level rounds at all -- same device-side digest of the state, same counters"""
    _, res = _solve_all()
    want = [[list(g.digest), _counters(g)] for g in res]
    e = dict(os.environ)
    e.update(env)
    out = subprocess.run([sys.executable, "-c", _CHILD % (os.path.dirname(HERE), HERE)], env=e, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT ")][-1]
    got = json.loads(line[len("RESULT "):])
    for (rel, _, _), w, g in zip(CASES, want, got):
        assert w == g, (rel, env)


def test_crew_rounds_soak():
    """12 more passes over the resident systems, alone and as one batch launch: one digest and one set of counters per circuit"""
    systems, first = _solve_all()
    want = [(g.digest, _counters(g)) for g in first]
    for it in range(12):
        res = E.solve_batch(systems, secp_solve=True, fetch_states="digest") if it % 2 else [E.solve_batch([s], secp_solve=True, fetch_states="digest")[0] for s in systems]
        for (rel, _, _), w, g in zip(CASES, want, res):
            assert (g.digest, _counters(g)) == w, (rel, it)
