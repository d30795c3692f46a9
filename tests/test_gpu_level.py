"""Level rounds (csrc/level.hip.hpp): the narrow dependency levels of a deep circuit executed round after round by one wavefront --
on a single-workgroup job (flags / in_queue tags in LDS) and on the master of a team (device memory). Bit-exact against the oracle,
whole state and counters, on the circuits they were built for, on a medium system made of chains (3 x EdDSAMiMCSponge in ONE file:
72 948 rows, too large for one workgroup's LDS -- 114 ms before the master of a team had level rounds, 22 ms with them), on forced
teams, and the same results with the level rounds switched off (ECNE_LEVEL=0 is read once per process: a subprocess)."""
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
ROOT = os.path.dirname(HERE)
SPONGE = "ecne_circomlib_tests/EdDSAMiMCSpongeVerifier@eddsamimcsponge.r1cs"
DEEP = [SPONGE, "ecne_circomlib_tests/EdDSAPoseidonVerifier@eddsaposeidon.r1cs", "ecne_circomlib_tests/EdDSAMiMCVerifier@eddsamimc.r1cs",
        "ecne_circomlib_tests/Poseidon@poseidon.r1cs", "ecne_circomlib_tests/BabyPbk@babyjub.r1cs", "ecne_circomlib_tests/MiMCSponge@mimcsponge.r1cs"]


@pytest.mark.parametrize("force_nwg", [0, 2, 6])
def test_deep_circuits_bit_exact(force_nwg):
    rels = [r for r in DEEP if r in fixtures.all_r1cs()]
    assert len(rels) >= 5
    systems = [build_system(r) for r in rels]
    res = E.solve_batch(systems, force_nwg=force_nwg)
    for r, g in zip(rels, res):
        assert_bit_exact("%s nwg=%d" % (r, force_nwg), g, orc.run(fixtures.path(r)))
        if force_nwg == 0 and "Poseidon@" in r:
            assert g.summary.sched[0] >= 300          # (schedule diagnostics: the levels ran as level rounds, ~408 of them)


def test_medium_system_of_chains_on_a_team():
    """3 x EdDSAMiMCSponge side by side in one file: a team job whose master runs the chains as level rounds on device-memory state"""
    import multi_copy
    p = multi_copy.cached(SPONGE, 3)
    s = build_system(None, path=p)
    assert len(s) == 3 * 24316
    g = E.solve_batch([s])[0]
    o = orc.run(p)
    assert_bit_exact("3 x EdDSAMiMCSponge", g, o)
    assert g.summary.sched[0] >= 2000 and g.summary.pops == 3 * 28073
    g2 = E.solve_batch([s])[0]                         # the same resident system again: same state, same counters
    assert_bit_exact("3 x EdDSAMiMCSponge, second solve", g2, o)


def test_same_results_without_level_rounds(tmp_path):
    """ECNE_LEVEL=0 (the schedule of round 3: chain bursts and fast wavefront rounds): identical state digests and counters"""
    code = r'''
import json, sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import ecneproject_amd as E
from gpu_common import build_system
out = {}
for rel in %r:
    g = E.solve_batch([build_system(rel)], fetch_states="digest")[0]
    s = g.summary
    out[rel] = [g.status, list(g.digest), s.pops, s.successful_steps, s.num_unique, s.outer_iterations, list(s.rule_hits[:13]), int(s.sched[0])]
print("RESULT " + json.dumps(out))
''' % (ROOT, HERE, DEEP[:4])
    res = []
    for lv in ("1", "0"):
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=dict(os.environ, ECNE_LEVEL=lv))
        assert out.returncode == 0, out.stderr[-2000:]
        res.append(json.loads([l for l in out.stdout.splitlines() if l.startswith("RESULT ")][0][7:]))
    for rel in DEEP[:4]:
        assert res[0][rel][:7] == res[1][rel][:7], rel
    assert res[0][SPONGE][7] > 1000           # level rounds ran in the first process ...
