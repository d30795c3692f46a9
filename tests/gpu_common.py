"""Shared helpers of the `-m gpu` parity tests: every call goes through the C ABI (ctypes)."""
import numpy as np

import ecneproject_amd as E
import fixtures
import orc


def build_system(rel, trusted=(), names=(), path=None):
    main = E.R1CS(path or fixtures.path(rel))
    fl = [(n, E.R1CS(fixtures.path(t))) for t, n in zip(trusted, names)]
    fl.sort(key=lambda x: -len(x[1]))          # solveWithTrustedFunctions :527
    s = E.System(main)
    for n, f in fl:
        s.abstract(f, n)
    return s


def assert_bit_exact(tag, g, o, counters=True):
    """GPU result g (ecneproject_amd.SolveResult) vs oracle result o (orc.OracleResult)."""
    assert g.status == o.status, (tag, g.status, o.status)
    if o.status != 0:
        return
    assert g.function_good == o.verdict, tag
    assert tuple(g.counts()) == tuple(o.counts()), (tag, g.counts(), o.counts())
    assert np.array_equal(g.flags, o.flags), tag                    # unique + is_known per variable
    assert np.array_equal(g.lb, o.lb), tag
    assert np.array_equal(g.ub, o.ub), tag
    assert np.array_equal(g.abz.astype(np.int64), o.abz), tag
    assert np.array_equal(g.nvalues, o.nvalues), tag
    assert np.array_equal(g.values, o.values), tag
    assert g.bad_rows.tolist() == o.bad_rows.tolist(), tag
    if counters:
        gs, os_ = g.summary, o.summary
        assert gs.successful_steps == os_.successful_steps, tag
        assert gs.outer_iterations == os_.outer_iterations, tag
        assert gs.num_unique == os_.num_unique, tag
        assert gs.pops == os_.pops, tag
        assert list(gs.rule_hits[:13]) == list(os_.rule_hits[:13]), tag
