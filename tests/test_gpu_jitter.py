"""The team protocols under scheduling noise (round 6). libecne_hip_jitter.so is the product's sources built with -DECNE_JITTER: thread 0 of
every workgroup sleeps a pseudo-random time (mostly nothing, one call in eight up to ~60 us, one in a thousand 0.2-0.4 ms) in front of every
barrier arrival and behind every release, workgroups start at random times and every other helper of a team is held back past the master's
first commands (csrc/job_barrier.hip.hpp). Results must not move; a barrier that does not complete ends in ECNE_ETIMEOUT.

The regression this file pins: a helper that was NOT on the commanded team went from the command barrier straight back to the next one,
and nothing made it read the command words before the master -- done with a short chain of rounds -- wrote the next command into the same
words; held up behind the release it read the next command's team size (and ran a chain nobody else ran) or its "queue phase over" (and left
the loop one barrier early). 3 % of the solves of these five systems on 24 workgroups ended in ECNE_ETIMEOUT under the jitter build; the
command block is per barrier parity now (Counters.q_cmd[2][12])."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JIT = os.path.join(ROOT, "ecneproject_amd", "libecne_hip_jitter.so")
pytestmark = pytest.mark.gpu


def _repro(args, seed, env=None):
    e = dict(os.environ, ECNE_LIB="libecne_hip_jitter.so", ECNE_JITTER_SEED=str(seed), PYTHONPATH=ROOT)
    e.update(env or {})
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "tools", "jitter_repro.py")] + [str(a) for a in args], capture_output=True, text=True, timeout=900, env=e, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    m = re.search(r"(\d+) tries x (\d+) systems, (\d+) differing", out.stdout)
    assert m, out.stdout[-2000:]
    return int(m.group(3)), out.stdout


@pytest.mark.skipif(not os.path.exists(JIT), reason="libecne_hip_jitter.so not built (__graft_entry__.build() builds it)")
@pytest.mark.parametrize("seed,env", [(1, {"ECNE_DRAIN": "2"}), (4, {})])
def test_commands_survive_late_readers(seed, env):
    """fuzz systems 341041-341045 (wide, long rows: chains of rounds on changing sub-teams) x 120 solves on 24 workgroups: the committed reproducer
    of the command race (4 of 750 solves timed out before the fix, 0 of 3 750 after)"""
    bad, log = _repro([341043, 4, 24, 120, 2], seed, env)
    assert bad == 0, log[-3000:]
    assert "ECNE TIMEOUT" not in log


@pytest.mark.skipif(not os.path.exists(JIT), reason="libecne_hip_jitter.so not built")
def test_long_decompositions_under_jitter():
    bad, log = _repro([320010, 0, 8, 40, 3], 2)
    assert bad == 0, log[-3000:]
