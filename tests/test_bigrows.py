"""Long rows (> 64 terms): the oracle must hit the intended rule, and the HIP engine's
workgroup-cooperative long-row path must agree with it bit for bit."""
import pytest

import bigrow_cases
import orc


@pytest.fixture(scope="module")
def big_dir(tmp_path_factory):
    d = tmp_path_factory.mktemp("bigrows")
    for name, (spec, _, _) in bigrow_cases.CASES.items():
        bigrow_cases.write(str(d / (name + ".r1cs")), spec)
    return d


@pytest.mark.parametrize("name", sorted(bigrow_cases.CASES))
def test_oracle_hits_the_intended_rule(big_dir, name):
    _, rule, fires = bigrow_cases.CASES[name]
    o = orc.run(str(big_dir / (name + ".r1cs")))
    assert o.status == 0
    assert (o.summary.rule_hits[rule] > 0) == fires, (name, list(o.summary.rule_hits[:13]))
    if name.startswith("r7_chain"):
        assert o.unique[2:-1].all()        # every digit became unique (the last id, nWires + 1, is unused)
    if name.startswith("r7_broken"):
        assert not o.unique[2:-1].any()


@pytest.mark.gpu
@pytest.mark.parametrize("force_nwg", [0, 3])
def test_gpu_bigrow_parity(big_dir, force_nwg):
    import ecneproject_amd as E
    from gpu_common import assert_bit_exact
    names = sorted(bigrow_cases.CASES)
    systems = [E.System(E.R1CS(str(big_dir / (n + ".r1cs")))) for n in names]
    for n, g in zip(names, E.solve_batch(systems, force_nwg=force_nwg)):
        assert_bit_exact("bigrow " + n, g, orc.run(str(big_dir / (n + ".r1cs"))))
        if n == "hub_fanout":
            # the over-long candidate list was replayed sequentially (general rounds), or the fast wavefront round handed
            # the high-fan-out events in a round of that one pop (sched[12]: events with > 3 target rows, walked by the wavefront)
            assert (g.summary.rule_hits[15] & 0xFF) or g.summary.sched[12] > 0, "high-fan-out path not exercised"


@pytest.mark.gpu
def test_gpu_bigrow_parity_general_rounds_only(big_dir, monkeypatch):
    """ECNE_LDS_BYTES=0 switches the fast wavefront round and the chain executor off: the general rounds alone, and the
    sequential replay of an over-long candidate list (resolve_pushes / queue_round_multi fallback) really runs."""
    import ecneproject_amd as E
    from gpu_common import assert_bit_exact
    monkeypatch.setenv("ECNE_LDS_BYTES", "0")
    names = sorted(bigrow_cases.CASES)
    systems = [E.System(E.R1CS(str(big_dir / (n + ".r1cs")))) for n in names]
    for force_nwg in (0, 3):
        for n, g in zip(names, E.solve_batch(systems, force_nwg=force_nwg)):
            assert_bit_exact("bigrow " + n, g, orc.run(str(big_dir / (n + ".r1cs"))))
            if n == "hub_fanout":
                assert g.summary.rule_hits[15] & 0xFF, "candidate-buffer fallback not exercised"
