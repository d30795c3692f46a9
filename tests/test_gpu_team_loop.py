"""The outer loop of a system on several workgroups (round 5, csrc/k_solve.hip.hpp): an outer iteration in which the master workgroup popped a
few rows by itself is finished by the master alone -- P3 (:1357-1417) and P4 (:1425-1483) from the rows popped since the last pass, the
group table and the per-row caches kept from the last pass that fired nothing -- and anything that could fire hands the iteration to the
full sweeps with every workgroup. These systems make that hand-over happen LATE: a reverse chain of isZero pairs (P5, one pair per outer
iteration, :1492-1550) delays the moment a variable z becomes unique by K iterations; z sits in every row of a k-row group over k other
unknowns, so the group has k + 1 unknowns (and k rows: nothing fires) until z is unique -- then the incremental pass finds the group
complete and the full pass has to fire it in row order. Compared with the oracle bit for bit on forced teams; `ecne_summary.team` says
that both kinds of iteration ran."""
import random

import pytest

import orc
import r1cs_py

P = r1cs_py.P


def delayed_group(K, k, seed, groups=1, singular=False):
    """variables: 1 one, 2 = input, y_0..y_{K-1} (isZero chain outputs), zc_0..zc_{K-1} (the pairs' hints), then per group: z, x_1..x_k.
    Rows: the K isZero pairs in REVERSE order (pair i needs y_{i-1}: one pair per outer iteration), z_g = y_{K-1} + g (made unique by R1 the
    iteration after the last pair fired), and k rows  sum_j c_rj x_j + d_r z + const = 0  per group."""
    rng = random.Random(seed)
    y = list(range(3, 3 + K))
    zc = list(range(3 + K, 3 + 2 * K))
    a = [2] + y[:-1]
    nxt = 3 + 2 * K
    pairs = [[([(a[i], 1)], [(zc[i], 1)], [(1, 1), (y[i], (-1) % P)]), ([(a[i], 1)], [(y[i], 1)], [])] for i in range(K)]
    pairs.reverse()
    rows = [r for pr in pairs for r in pr]
    for g in range(groups):
        z = nxt
        xs = list(range(nxt + 1, nxt + 1 + k))
        nxt += 1 + k
        rows.append(([], [], [(z, 1), (y[K - 1], (-1) % P), (1, (-(g + 1)) % P)]))       # z = y_{K-1} + g + 1
        first = None
        for r in range(k):
            cs = [rng.randrange(1, 50) for _ in xs]
            if singular and r == k - 1:
                cs = list(first)
            if r == 0:
                first = cs
            rows.append(([], [], [(x, c) for x, c in zip(xs, cs)] + [(z, rng.randrange(1, 9)), (1, rng.randrange(1, 1000))]))
    # a few rows that keep the queue busy in the late iterations: w_i = y_i * y_i (behind everything: a pair's two rows stay neighbours, :1493)
    for i in range(K):
        rows.append(([(y[i], 1)], [(y[i], 1)], [(nxt, 1)]))
        nxt += 1
    return dict(nwires=nxt - 1, nout=0, npub=1, nprv=0, rows=rows)


CASES = {
    "k2_after_3": delayed_group(3, 2, 1),
    "k3_after_4": delayed_group(4, 3, 2),
    "k4_after_6": delayed_group(6, 4, 3),
    "two_groups": delayed_group(5, 3, 4, groups=2),
    "singular_group": delayed_group(4, 3, 5, singular=True),
    "k5_after_9": delayed_group(9, 5, 6),
}


@pytest.fixture(scope="module")
def loop_dir(tmp_path_factory):
    d = tmp_path_factory.mktemp("team_loop")
    for name, spec in CASES.items():
        r1cs_py.write(str(d / (name + ".r1cs")), spec["nwires"], spec["nout"], spec["npub"], spec["nprv"], spec["rows"])
    return d


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_fires_the_group_late(loop_dir, name):
    import ref2
    from test_ref2 import differences
    p = str(loop_dir / (name + ".r1cs"))
    o = orc.run(p)
    assert o.status == 0
    assert differences(ref2.run(p), o) == []
    K = {"k2_after_3": 3, "k3_after_4": 4, "k4_after_6": 6, "two_groups": 5, "singular_group": 4, "k5_after_9": 9}[name]
    assert o.summary.rule_hits[12] == K                                       # one isZero pair per outer iteration
    assert o.summary.outer_iterations >= K + 2
    assert o.summary.rule_hits[10] == {"two_groups": 2}.get(name, 1) or name == "singular_group"      # the group(s) fired, after the chain


@pytest.mark.gpu
@pytest.mark.parametrize("force_nwg", [0, 2, 3, 8])
def test_gpu_late_groups_on_teams(loop_dir, force_nwg):
    import ecneproject_amd as E
    from gpu_common import assert_bit_exact
    names = sorted(CASES)
    systems = [E.System(E.R1CS(str(loop_dir / (n + ".r1cs")))) for n in names]
    for n, g in zip(names, E.solve_batch(systems, force_nwg=force_nwg)):
        o = orc.run(str(loop_dir / (n + ".r1cs")))
        assert_bit_exact("late group %s nwg=%d" % (n, force_nwg), g, o)
        t = list(g.summary.team)
        if force_nwg >= 2:
            # iterations on the master alone AND iterations with full sweeps (the first one, and the one in which the group fires)
            assert t[0] >= 2 and t[2] >= 2 and t[0] + t[2] == g.summary.outer_iterations, (n, t)
        else:
            assert t[0] == 0 and t[2] == 0


@pytest.mark.gpu
def test_gpu_bench_system_runs_most_iterations_on_the_master_alone():
    """ecdsa_like(6, 10) on its default team: one adder fires per outer iteration, a few dozen rows are popped -- those iterations need no helper"""
    import ecdsa_like
    import fixtures
    import ecneproject_amd as E
    from gpu_common import assert_bit_exact, build_system
    path = ecdsa_like.cached(6, 10)
    s = build_system(None, ["secp256k1.r1cs"], ["Secp256k1AddUnequal"], path=path)
    g = E.solve_batch([s])[0]
    assert_bit_exact("ecdsa_like(6,10)", g, orc.run(path, [fixtures.path("secp256k1.r1cs")], ["Secp256k1AddUnequal"]))
    t = list(g.summary.team)
    assert t[0] + t[2] == g.summary.outer_iterations and t[0] >= g.summary.outer_iterations - 4 and t[1] > 0
