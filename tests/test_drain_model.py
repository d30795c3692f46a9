"""The hazard logic of the drain round (ecneproject_amd/csrc/drain.hip.hpp) as a small executable model, checked against
sequential execution on random dependency structures -- the scheduling argument of DESIGN.md 4.2 ("Round 3: drain rounds") without
any solver rule in it.

A row has a static access set (what it may read or write in any state), a conservative write set (a subset of it) and an action:
a deterministic function of the state of its access set that writes some of its conservative set -- WHICH variables it writes
depends on what it reads, like a product row that writes its output only once its inputs are unique. The reference pops the rows in
queue order. The model drains the same window level by level with the three mark planes:

    X  exact writes on the state as it is (every pending row)          A  contested accesses + every access of an unstable row
    C  conservative writes of unstable rows (a lower X mark on something they access)
    stable rows: a lower C mark on an access demotes (lowest demoted rank = the level's cut), a lower A mark on a write waits

and checks (1) the rows that run in one level are pairwise conflict-free on what they actually do, (2) every row does exactly what it
does in the sequential order -- same writes, same values -- and (3) the final state is the sequential one."""
import random

import pytest


def make_case(rng, n_rows, n_vars, chainy):
    rows = []
    for i in range(n_rows):
        k = min(rng.randint(1, 4), n_vars)
        if chainy and i and rng.random() < 0.5:      # depend on a neighbour: chains and diamonds inside the window
            acc = set(rng.sample(sorted(rows[rng.randrange(max(0, i - 6), i)]["acc"]), 1))
        else:
            acc = set()
        while len(acc) < k:
            acc.add(rng.randrange(n_vars))
        acc = sorted(acc)
        wc = sorted(rng.sample(acc, rng.randint(0, len(acc))))
        rows.append(dict(acc=acc, wc=wc, salt=rng.randrange(1 << 30), thr=rng.randrange(4)))
    return rows


def action(row, state):
    """What the row would do on `state`: {variable: new value} for a subset of its conservative write set. A row fires on a
    variable only if the sum of what it reads passes a threshold (state-dependent write set), and writes a value that depends on
    everything it reads (so a stale read shows)."""
    s = sum(state[v] for v in row["acc"])
    out = {}
    for j, v in enumerate(row["wc"]):
        if (s + row["salt"] + j) % 4 >= row["thr"]:
            nv = (s * 31 + row["salt"] + 7 * j) % 1000003
            if nv != state[v]:
                out[v] = nv
    return out


def run_sequential(rows, state):
    state = list(state)
    did = []
    for row in rows:
        w = action(row, state)
        did.append(w)
        for v, x in w.items():
            state[v] = x
    return state, did


INF = 1 << 60


def run_drain(rows, state, rng):
    state = list(state)
    n = len(rows)
    pending = set(range(n))
    did = [None] * n
    levels = 0
    while pending:
        levels += 1
        # P1: exact writes on the state as it is
        wx = {r: action(rows[r], state) for r in pending}
        X = {}
        for r in pending:
            for v in wx[r]:
                X[v] = min(X.get(v, INF), r)
        # P2: unstable rows (a lower exact writer on something they access) mark C and A; contested accesses (a higher exact
        # writer) mark A
        unstable = set()
        A, C = {}, {}
        for r in pending:
            if any(X.get(v, INF) < r for v in rows[r]["acc"]):
                unstable.add(r)
        for r in pending:
            if r in unstable:
                for v in rows[r]["wc"]:
                    C[v] = min(C.get(v, INF), r)
                for v in rows[r]["acc"]:
                    A[v] = min(A.get(v, INF), r)
            else:
                for v in rows[r]["acc"]:
                    m = X.get(v, INF)
                    if m != INF and m > r:
                        A[v] = min(A.get(v, INF), r)
        # P3: demoted / waiting
        demoted = {r for r in pending - unstable if any(C.get(v, INF) < r for v in rows[r]["acc"])}
        waiting = {r for r in pending - unstable if any(A.get(v, INF) < r for v in wx[r])}
        dcut = min(demoted) if demoted else INF
        ready = [r for r in pending if r not in unstable and r not in demoted and r not in waiting and r < dcut]
        assert ready and min(pending) in ready, "the lowest pending row always runs"
        # (1) what runs together is conflict-free: nobody writes what another one of them accesses
        for r in ready:
            for q in ready:
                if q != r:
                    assert not (set(wx[r]) & set(rows[q]["acc"])), (r, q)
        rng.shuffle(ready)                      # any order, in particular all at once
        for r in ready:
            w = action(rows[r], state)           # decided again at run time, as the kernel does: must be what P1 saw
            assert w == wx[r]
            did[r] = w
        for r in ready:
            for v, x in did[r].items():
                state[v] = x
        pending -= set(ready)
    return state, did, levels


@pytest.mark.parametrize("chainy", [False, True])
def test_drain_levels_equal_sequential_pops(chainy):
    rng = random.Random(20260928 + chainy)
    total_levels = total_rows = 0
    for case in range(400):
        n_rows = rng.randint(1, 60)
        n_vars = rng.randint(2, 40)
        rows = make_case(rng, n_rows, n_vars, chainy)
        state0 = [rng.randrange(5) for _ in range(n_vars)]
        s_seq, did_seq = run_sequential(rows, state0)
        s_dr, did_dr, levels = run_drain(rows, state0, rng)
        assert did_dr == did_seq, case            # (2) every row does what it does in queue order
        assert s_dr == s_seq, case                # (3) and the state is the sequential one
        assert levels <= n_rows
        total_levels += levels
        total_rows += n_rows
    assert total_levels < total_rows              # it is a parallel schedule: fewer levels than rows


def test_independent_blocks_behind_each_other_drain_together():
    """the shape drain rounds were built for: block i = one row that writes k variables + k rows that read one each (a decoder sum
    and its dependents), blocks one behind the other in the queue, and in front of every writer a row that READS one of its targets
    (the prefix schedule ends there: one block per round). Two or three levels for any number of blocks."""
    rows, nv = [], 0
    for b in range(12):
        k = 6
        targets = list(range(nv, nv + k)); nv += k
        src = nv; nv += 1
        outs = list(range(nv, nv + k)); nv += k
        rows.append(dict(acc=[targets[0], outs[0]], wc=[outs[0]], salt=5, thr=0))               # the reader in front of the writer
        rows.append(dict(acc=[src] + targets, wc=targets, salt=3 + b, thr=0))                    # the block's writer
        for j in range(k):
            rows.append(dict(acc=[targets[j], outs[j]], wc=[outs[j]], salt=11 * j + b, thr=0))   # its dependents
    state0 = [1] * nv
    s_seq, did_seq = run_sequential(rows, state0)
    s_dr, did_dr, levels = run_drain(rows, state0, random.Random(1))
    assert did_dr == did_seq and s_dr == s_seq
    assert levels <= 3
