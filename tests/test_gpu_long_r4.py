"""Long binary decompositions (R4, /root/reference/src/R1CSConstraintSolver.jl:991-1076) in the states the engine's shortcuts tell apart
(csrc/fastrow.hip.hpp): `long_r4_idle` -- pivot and lowest bit not unique, the pivot's bounds cut already: the pop is settled from two flag
bytes and the pivot's bounds, by the pop at the queue head, the declined loop and the level rounds --, `exec_long_r4` -- the two pops that
do something, through R4 alone --, `long_r4_done` and the watched pair. tests/fuzz_r1cs.py: make_decomp -- both orientations (the reference
negates a row of the second one, :1001-1011: the pivot is then the term stored with -1), bits with and without bounds, pivots pinned at
once / late / never / shared by two decompositions. Oracle = second reading on the CPU; engine = oracle bit for bit on the GPU, as
single-workgroup jobs, with every executor switched off in turn and on forced teams."""
import os

import pytest

import fuzz_r1cs
import orc

N = 240


@pytest.fixture(scope="module")
def dec_dir(tmp_path_factory):
    d = tmp_path_factory.mktemp("decomp")
    for seed in range(N):
        fuzz_r1cs.write(str(d / ("%d.r1cs" % seed)), fuzz_r1cs.make_decomp(seed))
    return d


def test_oracle_and_second_reading_agree_and_r4_fires(dec_dir):
    import ref2
    from test_ref2 import differences
    r4 = cuts_only = 0
    for seed in range(0, N, 6):
        p = str(dec_dir / ("%d.r1cs" % seed))
        o = orc.run(p)
        assert o.status == 0
        assert differences(ref2.run(p), o) == [], seed
        r4 += o.summary.rule_hits[3] > 0
    assert r4 >= N // 6 * 3 // 4


def test_both_orientations_and_all_states_occur(dec_dir):
    """what the generator is for: rows stored with the pivot at -1, rows whose bits lack bounds, pivots never pinned"""
    neg = unpinned = 0
    for seed in range(N):
        spec = fuzz_r1cs.make_decomp(seed)
        for A, B, C in spec["rows"]:
            if not A and not B and len(C) >= 16:
                cs = sorted(c % orc.P for _, c in C)
                neg += cs[0] == 1 and cs[1] == 2          # {-1 (pivot), 1, 2, 4, ...}: smallest residues 1, 2
        o = orc.run(str(dec_dir / ("%d.r1cs" % seed)))
        unpinned += o.summary.num_unique < spec["n_wires"]
    assert neg >= N // 2 and unpinned >= N // 4


@pytest.mark.gpu
@pytest.mark.parametrize("force_nwg", [0, 2, 5])
def test_gpu_decompositions_bit_exact(dec_dir, force_nwg):
    import ecneproject_amd as E
    from gpu_common import assert_bit_exact
    systems = [E.System(E.R1CS(str(dec_dir / ("%d.r1cs" % seed)))) for seed in range(N)]
    for seed, g in enumerate(E.solve_batch(systems, force_nwg=force_nwg)):
        assert_bit_exact("decomp %d nwg=%d" % (seed, force_nwg), g, orc.run(str(dec_dir / ("%d.r1cs" % seed))))


@pytest.mark.gpu
@pytest.mark.parametrize("switch", ["ECNE_LEVEL", "ECNE_CREW", "ECNE_R4DONE"])
def test_gpu_decompositions_with_an_executor_off(dec_dir, switch):
    """the same systems with the level rounds / the crew rounds / the finished-row shortcuts off: other executors meet the same rows"""
    import subprocess
    import sys
    code = (
        "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import ecneproject_amd as E, orc\n"
        "from gpu_common import assert_bit_exact\n"
        "d = %r\n"
        "ss = [E.System(E.R1CS('%%s/%%d.r1cs' %% (d, s))) for s in range(0, %d, 2)]\n"
        "for k, g in enumerate(E.solve_batch(ss)): assert_bit_exact('decomp %%d' %% (2 * k), g, orc.run('%%s/%%d.r1cs' %% (d, 2 * k)))\n"
        "print('ok')\n"
    ) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)), str(dec_dir), N)
    env = dict(os.environ)
    env[switch] = "0"
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.stdout[-2000:], r.stderr[-2000:])
