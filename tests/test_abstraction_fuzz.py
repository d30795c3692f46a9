"""Random main circuits with embedded copies of random trusted sub-circuits: the product's host-side
abstraction (ecneproject_amd/csrc/host_model.hpp, reference :237-395) must find the same instances, map
the same variables and keep the same rows as the oracle's restatement -- including self-similar patterns
whose candidate windows overlap (the greedy replacement with the reference's stuck cursor, :368-388),
near-copies that differ in one coefficient, copies that share variables and permuted term order."""
import random

import pytest

import orc
import r1cs_py

P = r1cs_py.P


def _rand_sub(rng):
    """a small sub-circuit: n_in inputs, 1-2 outputs, a few internal signals, 2-6 rows"""
    kind = rng.random()
    if kind < 0.35:                       # self-similar chain x_{i+1} = x_i^2 (+ c): windows overlap in a longer chain
        L = rng.randint(2, 4)
        c = rng.choice([0, 0, 3])
        # variables: 1 one, 2 out, 3 in, 4.. internals
        chain = [3] + list(range(4, 3 + L)) + [2]
        rows = [([(chain[i], 1)], [(chain[i], 1)], [(chain[i + 1], 1)] + ([(1, (-c) % P)] if c else [])) for i in range(L)]
        return dict(nwires=L + 1, nout=1, npub=1, nprv=0, rows=rows), ("chain", L, c)
    n_in, n_out, n_int = rng.randint(1, 3), rng.randint(1, 2), rng.randint(0, 3)
    nv = 1 + n_out + n_in + n_int
    vs = list(range(2, nv + 1))
    rows = []
    for _ in range(rng.randint(2, 6)):
        def terms(n):
            return [(v, rng.choice([1, 2, 3, 5, P - 1, P - 2, 1 << 20])) for v in rng.sample(vs + [1], min(n, len(vs) + 1))]
        if rng.random() < 0.5:
            rows.append((terms(rng.randint(1, 2)), terms(rng.randint(1, 2)), terms(rng.randint(0, 3))))
        else:
            rows.append(([], [], terms(rng.randint(2, 4))))
    return dict(nwires=nv - 1, nout=n_out, npub=n_in, nprv=0, rows=rows), ("random",)


def _rand_main(rng, sub, tag):
    """main circuit: renamed copies of `sub` (fresh variables, or sharing some with the previous copy),
    random rows in between, the odd near-copy; for chains also one long chain (overlapping windows)"""
    nv = [4]            # next fresh variable id (1 one, 2 output, 3 input are the main file's own)
    def fresh():
        nv[0] += 1
        return nv[0] - 1
    rows = []
    sub_vars = sorted({v for r in sub["rows"] for part in r for v, _ in part} - {1})
    prev_map = None
    for _ in range(rng.randint(1, 5)):
        r = rng.random()
        if r < 0.55:
            m = {1: 1}
            for v in sub_vars:
                m[v] = prev_map[v] if (prev_map and rng.random() < 0.15) else fresh()
            block = [tuple([(m[v], c) for v, c in (rng.sample(part, len(part)) if rng.random() < 0.5 else part)] for part in row) for row in sub["rows"]]
            if rng.random() < 0.15 and block:      # near-copy: one coefficient off
                i = rng.randrange(len(block))
                parts = [list(p) for p in block[i]]
                nonempty = [p for p in parts if p]
                if nonempty:
                    p = rng.choice(nonempty)
                    v, c = p[0]
                    p[0] = (v, (c + 1) % P)
                block[i] = tuple(parts)
            rows += block
            prev_map = m
        elif r < 0.8:
            for _ in range(rng.randint(1, 3)):
                a, b, c = fresh(), fresh(), fresh()
                rows.append(([(a, 1)], [(b, rng.choice([1, 2]))], [(c, 1)]))
        elif tag[0] == "chain":
            L, c = tag[1], tag[2]
            M = L + rng.randint(1, 2 * L + 1)
            chain = [fresh() for _ in range(M + 1)]
            rows += [([(chain[i], 1)], [(chain[i], 1)], [(chain[i + 1], 1)] + ([(1, (-c) % P)] if c else [])) for i in range(M)]
    if not rows:
        rows.append(([], [], [(2, 1), (3, P - 1)]))
    return dict(nwires=nv[0] - 2, nout=1, npub=1, nprv=0, rows=rows)


N_CASES = 300


@pytest.fixture(scope="module")
def abs_dir(tmp_path_factory):
    d = tmp_path_factory.mktemp("absfuzz")
    for seed in range(N_CASES):
        rng = random.Random(77000 + seed)
        sub, tag = _rand_sub(rng)
        main = _rand_main(rng, sub, tag)
        r1cs_py.write(str(d / ("sub%d.r1cs" % seed)), sub["nwires"], sub["nout"], sub["npub"], sub["nprv"], sub["rows"])
        r1cs_py.write(str(d / ("main%d.r1cs" % seed)), main["nwires"], main["nout"], main["npub"], main["nprv"], main["rows"])
    return d


def test_host_abstraction_matches_oracle(abs_dir):
    from ecneproject_amd import build
    build.build()
    import ecneproject_amd as E
    n_inst = n_stuck = 0
    for seed in range(N_CASES):
        mp, sp = str(abs_dir / ("main%d.r1cs" % seed)), str(abs_dir / ("sub%d.r1cs" % seed))
        o = orc.run(mp, [sp], ["T"], want_states=False)
        s = E.System(E.R1CS(mp))
        try:
            s.abstract(E.R1CS(sp), "T")
            st = 0
        except E.EcneError as e:
            st = e.status
        assert st == (o.status if o.status == -5 else 0), (seed, st, o.status)
        if st != 0:
            continue
        assert s.specials() == o.specials, seed
        assert len(s) == o.summary.n_rows_reduced, seed
        n_inst += len(o.specials)
    assert n_inst > 150          # the generator really embeds instances


@pytest.mark.gpu
def test_gpu_solve_after_abstraction_parity(abs_dir):
    import ecneproject_amd as E
    from gpu_common import assert_bit_exact
    systems, oracles = [], []
    for seed in range(N_CASES):
        mp, sp = str(abs_dir / ("main%d.r1cs" % seed)), str(abs_dir / ("sub%d.r1cs" % seed))
        o = orc.run(mp, [sp], ["T"])
        if o.status == -5:
            continue
        s = E.System(E.R1CS(mp))
        s.abstract(E.R1CS(sp), "T")
        systems.append(s)
        oracles.append((seed, o))
    for (seed, o), g in zip(oracles, E.solve_batch(systems)):
        assert_bit_exact("abstraction fuzz %d" % seed, g, o)
