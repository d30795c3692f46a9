"""Degenerate inputs the reader, the layout and the engine must take exactly as the reference does:
no constraints at all, constraints with empty parts only (ParseR1CS.jl:113-115 stores {1 => 0}), explicit
zero coefficients only, no outputs, no inputs, a wire id equal to nWires (accepted),
one row repeated many times.  CPU part: reader and flat layout against the oracle's reader; GPU part:
bit-exact solve parity through the C ABI, alone and with helper workgroups forced on."""
import pytest

import fuzz_r1cs
import orc

P = orc.P

# name -> (nwires, nout, npub, nprv, rows)
CASES = {
    "no_constraints": (4, 1, 1, 2, []),
    "no_constraints_no_outputs": (3, 0, 2, 1, []),
    "one_wire_only": (1, 0, 0, 0, []),
    "empty_parts_only": (4, 1, 1, 2, [([], [], []), ([], [], [])]),
    "explicit_zeros_only": (4, 1, 1, 2, [([(2, 0)], [(3, 0)], [(4, 0), (1, 0)])]),
    "no_outputs": (4, 0, 2, 2, [([(2, 1)], [(3, 1)], [(4, 1)])]),
    "no_inputs": (3, 2, 0, 1, [([], [], [(2, 1), (1, P - 5)]), ([(2, 1)], [(2, 1)], [(3, 1)])]),
    "wire_id_equals_nwires": (3, 1, 1, 1, [([], [], [(4, 1), (3, P - 1)]), ([], [], [(2, 1), (4, P - 1)])]),
    "same_row_many_times": (4, 1, 1, 2, [([(3, 1)], [(4, 1)], [(2, 1)])] * 40),
    "constant_only_rows": (3, 1, 1, 1, [([], [], [(1, 0)]), ([(1, 2)], [(1, 3)], [(1, 6)]), ([], [], [(2, 1), (3, P - 1)])]),
}


@pytest.fixture(scope="module")
def edge_dir(tmp_path_factory):
    d = tmp_path_factory.mktemp("edge")
    for name, (nw, no, npub, nprv, rows) in CASES.items():
        fuzz_r1cs.write_raw(str(d / (name + ".r1cs")), nw, no, npub, nprv, rows)
    return d


@pytest.mark.parametrize("name", sorted(CASES))
def test_reader_and_layout(edge_dir, name):
    from ecneproject_amd import build
    build.build()
    import ecneproject_amd as E
    p = str(edge_dir / (name + ".r1cs"))
    f, kn, out, nv = E.readR1CS(p)
    st, d = orc.read_info(p)
    assert st == 0
    assert list(f.info.nnz) == d["nnz"] and nv == d["nVars"] and kn == d["knowns"] and out == d["outputs"]
    s = E.System(f)
    assert len(s) == len(CASES[name][4])
    for part in range(3):
        rp, col, cf = s.rows(part)
        assert len(rp) == len(s) + 1 and int(rp[-1]) == d["nnz"][part] == len(col)
    assert orc.run(p).status in (0, -2, -3)       # the oracle terminates (possibly with the reference's exception)


@pytest.mark.gpu
@pytest.mark.parametrize("force_nwg", [0, 3])
def test_gpu_edge_parity(edge_dir, force_nwg):
    import ecneproject_amd as E
    from gpu_common import assert_bit_exact
    names = sorted(CASES)
    systems = [E.System(E.R1CS(str(edge_dir / (n + ".r1cs")))) for n in names]
    for n, g in zip(names, E.solve_batch(systems, force_nwg=force_nwg)):
        assert_bit_exact("edge case %s nwg=%d" % (n, force_nwg), g, orc.run(str(edge_dir / (n + ".r1cs"))))
