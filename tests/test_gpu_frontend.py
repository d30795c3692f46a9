"""Device front-end (ecne_frontend.hip: parse, abstraction and layout as gfx950 kernels) against the host front-end
(host_model.hpp + build_layout) and the oracle, through the C ABI.

* rows in DICTIONARY order after the device parse == after the host parse (ParseR1CS.jl:50-124), on every fixture, on the fuzz
  files (repeated wire ids, explicit zeros, un-reduced coefficients, all three section orders) and on damaged files (same accept /
  reject as the oracle's reader);
* after abstraction on the device: the same reduced rows and the same special constraints as the host path
  (R1CSConstraintSolver.jl:237-395) on the reference configurations and on the abstraction fuzz family (overlapping windows,
  near-copies, KeyError);
* every static array of the device image (CSR in nonzeroKeys order, row descriptors, fan-out, P4 / P5 / long-row lists, row
  records) byte for byte equal between a host-laid and a device-laid system;
* solves through the device front-end bit-exact against the oracle.
"""
import os
import random

import numpy as np
import pytest

import fixtures
import fuzz_r1cs
import orc

pytestmark = pytest.mark.gpu

ARRAYS = ["rpA", "rpB", "rpC", "colA", "colB", "colC", "coefA", "coefB", "coefC", "rinfo", "fo_ptr", "fo_rows", "nontrivial",
          "p4_list", "p4_b", "p4_s", "cls_list", "tbig", "bigrows", "long_list", "p5_rows", "p5_y", "rec", "foi", "scalars"]


@pytest.fixture(autouse=True)
def _restore_frontend():
    import ecneproject_amd as E
    before = E.set_frontend()
    yield
    E.set_frontend(before)


def _system(E, mode, rel=None, trusted=(), names=(), path=None):
    E.set_frontend(mode)
    main = E.R1CS(path or fixtures.path(rel))
    fl = [(n, E.R1CS(fixtures.path(t) if not os.path.isabs(t) else t)) for t, n in zip(trusted, names)]
    fl.sort(key=lambda x: -len(x[1]))
    s = E.System(main)
    for n, f in fl:
        s.abstract(f, n)
    return main, s


def _same_dict_rows(tag, a, b):
    for part in range(3):
        ra, rb = a.dict_rows(part), b.dict_rows(part)
        for x, y, what in zip(ra, rb, ("ptr", "var", "coef")):
            assert np.array_equal(x, y), (tag, part, what)


def _same_static_arrays(tag, a, b):
    sa, sb = a.static_array(24), b.static_array(24)
    va, vb = np.frombuffer(sa, np.int64), np.frombuffer(sb, np.int64)
    assert va[8] == 0 and vb[8] == 1, (tag, "which front-end laid the system out", va[8], vb[8])
    assert va[:8].tolist() == vb[:8].tolist(), (tag, "scalars", va.tolist(), vb.tolist())
    for which, name in enumerate(ARRAYS[:-1]):
        xa, xb = a.static_array(which), b.static_array(which)
        if xa != xb:
            w = 4
            ia, ib = np.frombuffer(xa[:len(xa) // w * w], np.uint32), np.frombuffer(xb[:len(xb) // w * w], np.uint32)
            first = int(np.argmax(ia != ib)) if len(ia) == len(ib) else -1
            raise AssertionError((tag, name, len(xa), len(xb), first, ia[first:first + 8].tolist() if first >= 0 else None,
                                  ib[first:first + 8].tolist() if first >= 0 else None))


def test_parse_and_layout_all_fixtures():
    import ecneproject_amd as E
    for rel in fixtures.all_r1cs():
        mh, sh = _system(E, E.FRONTEND_HOST, rel)
        md, sd = _system(E, E.FRONTEND_DEVICE, rel)
        assert E.frontend_stats()["parse_device"] == 1.0, rel
        assert list(mh.info.nnz) == list(md.info.nnz) and mh.io() == md.io() and int(mh.info.n_vars) == int(md.info.n_vars), rel
        _same_dict_rows(rel, sh, sd)
        _same_static_arrays(rel, sh, sd)


def _rows_against_second_reading(tag, sysm, path):
    """dictionary-order rows of a system (host or device reader) against tests/ref2.py's reader -- an independent parser with its
    own model of Julia's Dict: same keys in the same iteration order, same values (reduced mod p, last value of a repeated wire id
    at the first occurrence's position, explicit zeros kept, {1 => 0} for an empty part; ParseR1CS.jl:96-115)"""
    import ref2
    eqs, _kn, _out, _nv = ref2.read_r1cs(path)
    parts = [sysm.dict_rows(p) for p in range(3)]
    assert all(len(parts[p][0]) == len(eqs) + 1 for p in range(3)), (tag, "row count")
    for p in range(3):
        ptr, var, coef = parts[p]
        ints = orc.limbs_to_int(coef) if len(coef) else []
        for i, eq in enumerate(eqs):
            d = (eq.a, eq.b, eq.c)[p]
            keys = list(d.keys())
            a, b = int(ptr[i]), int(ptr[i + 1])
            assert var[a:b].tolist() == keys, (tag, "keys of row %d part %d" % (i + 1, p), var[a:b].tolist(), keys)
            assert ints[a:b] == [d[k] for k in keys], (tag, "values of row %d part %d" % (i + 1, p))


def test_both_readers_against_the_second_reading(tmp_path):
    """every fixture, the fuzz family (values >= p, repeated ids, explicit zeros), ids above nVars and the three section orders:
    rows of the DEVICE reader and of the host reader, each row by row against an independent parser (not only against each other)"""
    import ecneproject_amd as E
    n_rows = 0
    for rel in fixtures.all_r1cs():
        _md, sd = _system(E, E.FRONTEND_DEVICE, rel)
        assert E.frontend_stats()["parse_device"] == 1.0, rel
        _rows_against_second_reading(rel, sd, fixtures.path(rel))
        if len(sd) <= 4000:
            _mh, sh = _system(E, E.FRONTEND_HOST, rel)
            _rows_against_second_reading(rel + " (host)", sh, fixtures.path(rel))
        n_rows += len(sd)
    assert n_rows > 150000
    for seed in range(0, 120):
        p = str(tmp_path / ("f%d.r1cs" % seed))
        fuzz_r1cs.write(p, fuzz_r1cs.make(seed) if seed % 4 else fuzz_r1cs.make_oob(seed))
        for mode in (E.FRONTEND_DEVICE, E.FRONTEND_HOST):
            _m, s = _system(E, mode, path=p)
            _rows_against_second_reading((seed, mode), s, p)
    rows = [([(2, 0), (2, 5), (4, orc.P + 2)], [], [(3, 5), (2, 1), (3, 7), (1, orc.P - 9), (2, 0)])]
    for order in ((2, 1, 3), (1, 2, 3), (3, 1, 2), (3, 2, 1)):
        p = str(tmp_path / ("o%d%d%d.r1cs" % order))
        fuzz_r1cs.write_raw(p, 3, 1, 0, 1, rows, section_order=order)
        for mode in (E.FRONTEND_DEVICE, E.FRONTEND_HOST):
            _m, s = _system(E, mode, path=p)
            _rows_against_second_reading((order, mode), s, p)
    # long parts in every tier of the device reader (tables in LDS / HBM), with repeats
    from test_fuzz import _many_block_rows
    rng = random.Random(7)
    rows = _many_block_rows(600, 300, 99)
    for i, n in enumerate((12, 43, 90, 171, 700, 1100, 3500)):
        terms = [(rng.randint(1, 5000), rng.choice([0, 1, 2, orc.P - 1, orc.P + 3, rng.getrandbits(250)])) for _ in range(n)]
        terms[rng.randrange(1, n)] = (terms[0][0], 7)
        a = list(rows[40 * i + 3])
        a[i % 3] = terms
        rows[40 * i + 3] = tuple(a)
    p = str(tmp_path / "tiers.r1cs")
    fuzz_r1cs.write_raw(p, 4999, 1, 1, 4997, rows)
    for mode in (E.FRONTEND_DEVICE, E.FRONTEND_HOST):
        _m, s = _system(E, mode, path=p)
        _rows_against_second_reading(("tiers", mode), s, p)


def test_abstraction_reference_configs():
    import ecneproject_amd as E
    for rel, trusted, names, _secp, _verdict in fixtures.REFERENCE_ASSERTED:
        if not trusted:
            continue
        _mh, sh = _system(E, E.FRONTEND_HOST, rel, trusted, names)
        _md, sd = _system(E, E.FRONTEND_DEVICE, rel, trusted, names)
        assert E.frontend_stats()["abstract_device"] == 1.0, rel
        assert sh.specials() == sd.specials(), rel
        assert len(sh) == len(sd), rel
        _same_dict_rows(rel, sh, sd)
        _same_static_arrays(rel, sh, sd)


def test_fuzz_files_section_orders_and_damage(tmp_path):
    import ecneproject_amd as E
    # the fuzz family of test_fuzz.py: degenerate rows, duplicates, zeros, values >= p
    for seed in range(0, 120, 3):
        p = str(tmp_path / ("f%d.r1cs" % seed))
        fuzz_r1cs.write(p, fuzz_r1cs.make(seed))
        _mh, sh = _system(E, E.FRONTEND_HOST, path=p)
        md, sd = _system(E, E.FRONTEND_DEVICE, path=p)
        st, d = orc.read_info(p)
        assert st == 0 and list(md.info.nnz) == d["nnz"], seed
        _same_dict_rows(seed, sh, sd)
        _same_static_arrays(seed, sh, sd)
    rows = [([], [], [(3, 5), (2, 1), (3, 7), (1, orc.P - 9)])]
    for order in ((2, 1, 3), (1, 2, 3), (3, 1, 2)):
        p = str(tmp_path / ("s%d%d%d.r1cs" % order))
        fuzz_r1cs.write_raw(p, 3, 1, 0, 1, rows, section_order=order)
        _mh, sh = _system(E, E.FRONTEND_HOST, path=p)
        md, sd = _system(E, E.FRONTEND_DEVICE, path=p)
        assert list(md.info.nnz) == [0, 0, 3]
        _same_dict_rows(order, sh, sd)
    # damaged files: accept / reject like the oracle's reader
    rng = random.Random(5)
    src = str(tmp_path / "good.r1cs")
    fuzz_r1cs.write(src, fuzz_r1cs.make(3))
    data = open(src, "rb").read()
    E.set_frontend(E.FRONTEND_DEVICE)
    n_rej = 0
    for case in range(150):
        d = bytearray(data)
        if case % 3 == 0:
            d = d[:rng.randrange(0, len(d))]
        elif case % 3 == 1:
            pos = rng.randrange(0, min(len(d), 200))
            d[pos] ^= 1 << rng.randrange(8)
        else:
            pos = rng.randrange(0, len(d) - 4)
            d[pos:pos + 4] = rng.choice([bytes([255, 255, 255, 127]), bytes(4), bytes([16, 0, 0, 0])])
        p = str(tmp_path / ("bad%d.r1cs" % case))
        open(p, "wb").write(bytes(d))
        st_o, info = orc.read_info(p)
        try:
            f = E.R1CS(p)
            st_n = 0
        except E.EcneError as e:
            st_n = e.status
        assert (st_n == 0) == (st_o == 0), (case, st_n, st_o)
        if st_n == 0:
            assert list(f.info.nnz) == info["nnz"] and int(f.info.n_constraints) == info["nConstraints"], case
        n_rej += st_n != 0
    assert n_rej > 30


def test_many_rows_with_repeated_wire_ids(tmp_path):
    """7 000 rows, every ~8th part repeating a wire id (the device parse closes the gaps in a second pass), long parts woven in"""
    import ecneproject_amd as E
    from test_fuzz import _many_block_rows
    rng = random.Random(99)
    n_vars = 9000
    rows = _many_block_rows(7000, 400, 4242)
    for i in range(0, 7000, 500):      # parts of 11..170 terms, 171..2730 (tables in LDS) and beyond (tables in HBM), some with repeats
        for n in (12, 43, 90, 171, 700, 1100, 3500):
            terms = [(rng.randint(1, n_vars), rng.choice([0, 1, 2, orc.P - 1, orc.P + 3, rng.getrandbits(250)])) for _ in range(n)]
            if rng.random() < 0.5:
                terms[rng.randrange(1, n)] = (terms[0][0], 7)
            a = list(rows[i + (n % 17)])
            a[rng.randrange(3)] = terms
            rows[i + (n % 17)] = tuple(a)
    p = str(tmp_path / "blocks.r1cs")
    fuzz_r1cs.write_raw(p, n_vars - 1, 1, 1, n_vars - 3, rows)
    _mh, sh = _system(E, E.FRONTEND_HOST, path=p)
    md, sd = _system(E, E.FRONTEND_DEVICE, path=p)
    st, d = orc.read_info(p)
    assert st == 0 and list(md.info.nnz) == d["nnz"]
    _same_dict_rows("blocks", sh, sd)
    _same_static_arrays("blocks", sh, sd)


def test_abstraction_fuzz_family(tmp_path):
    import ecneproject_amd as E
    import r1cs_py
    import test_abstraction_fuzz as TA
    n = n_inst = 0
    for seed in range(0, TA.N_CASES):
        rng = random.Random(77000 + seed)
        sub, tag = TA._rand_sub(rng)
        main = TA._rand_main(rng, sub, tag)
        sp, mp = str(tmp_path / ("sub%d.r1cs" % seed)), str(tmp_path / ("main%d.r1cs" % seed))
        r1cs_py.write(sp, sub["nwires"], sub["nout"], sub["npub"], sub["nprv"], sub["rows"])
        r1cs_py.write(mp, main["nwires"], main["nout"], main["npub"], main["nprv"], main["rows"])
        outs = []
        for mode in (E.FRONTEND_HOST, E.FRONTEND_DEVICE):
            E.set_frontend(mode)
            try:
                s = E.System(E.R1CS(mp))
                s.abstract(E.R1CS(sp), "T")
                outs.append((0, s.specials(), len(s), s))
            except E.EcneError as e:
                outs.append((e.status, None, None, None))
        assert outs[0][:3] == outs[1][:3], (seed, outs[0][:3], outs[1][:3])
        if outs[0][0] == 0:
            o = orc.run(mp, [sp], ["T"], want_states=False)
            assert outs[1][1] == o.specials and outs[1][2] == o.summary.n_rows_reduced, seed
            _same_dict_rows(seed, outs[0][3], outs[1][3])
            n += 1
            n_inst += len(outs[1][1])
    assert n > 100 and n_inst > 150


@pytest.mark.parametrize("force_nwg", [0, 3])
def test_solves_through_the_device_frontend(force_nwg):
    import ecneproject_amd as E
    from gpu_common import assert_bit_exact
    E.set_frontend(E.FRONTEND_DEVICE)
    for rel, trusted, names, secp, verdict in fixtures.REFERENCE_ASSERTED:
        _m, s = _system(E, E.FRONTEND_DEVICE, rel, trusted, names)
        g = E.solve_batch([s], secp_solve=secp, force_nwg=force_nwg)[0]
        o = orc.run(fixtures.path(rel), [fixtures.path(t) for t in trusted], names, secp)
        assert_bit_exact(rel, g, o)
        assert g.function_good == verdict, rel
    suite = fixtures.circomlib_suite()
    systems = [_system(E, E.FRONTEND_DEVICE, rel)[1] for rel in suite]
    for rel, g in zip(suite, E.solve_batch(systems, force_nwg=force_nwg)):
        assert_bit_exact(rel, g, orc.run(fixtures.path(rel)))


def test_ecdsa_like_small_through_both_frontends():
    import ecneproject_amd as E
    import ecdsa_like
    from gpu_common import assert_bit_exact
    p = ecdsa_like.cached(3, 10)
    tr = fixtures.path("secp256k1.r1cs")
    _mh, sh = _system(E, E.FRONTEND_HOST, path=p, trusted=[tr], names=["Secp256k1AddUnequal"])
    _md, sd = _system(E, E.FRONTEND_DEVICE, path=p, trusted=[tr], names=["Secp256k1AddUnequal"])
    st = E.frontend_stats()
    assert st["abstract_device"] == 1.0 and st["matched"] == 2.0, st
    assert sh.specials() == sd.specials()
    _same_dict_rows("ecdsa_like(3)", sh, sd)
    _same_static_arrays("ecdsa_like(3)", sh, sd)
    g = E.solve_batch([sd])[0]
    o = orc.run(p, [tr], ["Secp256k1AddUnequal"], False)
    assert_bit_exact("ecdsa_like(3)", g, o)


def test_paths_the_device_hands_back(tmp_path, monkeypatch):
    """(1) a trusted function whose mapped inputs share their appearance signature (out = a + b, t = (a + b)^2: a and b tie; the reference breaks the
    tie by hash-table order of the window's variables) goes through the host path as a whole; (2) with ECNE_FE_FORCE_HOST_VERIFY every
    window the device accepts is verified again by the host code on a downloaded copy (the path a 128-bit signature collision would
    take); (3) a file whose constraint section does not start on a word boundary (field size 33) is uploaded from the odd address;
    (4) a file without constraints. Results: the host front-end's and the oracle's."""
    import struct
    import ecneproject_amd as E
    import r1cs_py
    P = orc.P
    # (1) ties: sub = { out = a + b ; t = (a + b)^2 } x main with three renamed copies
    sub = str(tmp_path / "tie_sub.r1cs")
    r1cs_py.write(sub, 4, 1, 2, 0, [([], [], [(2, 1), (3, P - 1), (4, P - 1)]), ([(3, 1), (4, 1)], [(3, 1), (4, 1)], [(5, 1)])])
    rows = []
    for c in range(3):
        o_, a_, b_, t_ = 10 + 4 * c, 11 + 4 * c, 12 + 4 * c, 13 + 4 * c
        rows += [([], [], [(o_, 1), (a_, P - 1), (b_, P - 1)]), ([(a_, 1), (b_, 1)], [(a_, 1), (b_, 1)], [(t_, 1)])]
        rows.append(([(2, 1)], [(3, 1)], [(4 + c, 1)]))
    main = str(tmp_path / "tie_main.r1cs")
    r1cs_py.write(main, 24, 1, 2, 0, rows)
    outs = []
    for mode in (E.FRONTEND_HOST, E.FRONTEND_DEVICE):
        E.set_frontend(mode)
        s = E.System(E.R1CS(main))
        s.abstract(E.R1CS(sub), "T")
        outs.append((s.specials(), len(s), s))
    assert E.frontend_stats()["abstract_device"] == 0.0          # the tie sent it to the host path
    o = orc.run(main, [sub], ["T"], want_states=False)
    assert outs[0][:2] == outs[1][:2] and outs[1][0] == o.specials and len(o.specials) == 3
    _same_dict_rows("ties", outs[0][2], outs[1][2])
    # (2) forced host verification of device-accepted windows
    monkeypatch.setenv("ECNE_FE_FORCE_HOST_VERIFY", "1")
    _mh, sh = _system(E, E.FRONTEND_HOST, "secp256k1.r1cs", ["bigmultmodp.r1cs", "biglessthan.r1cs"], ["BigMultModP", "BigLessThan"])
    _md, sd = _system(E, E.FRONTEND_DEVICE, "secp256k1.r1cs", ["bigmultmodp.r1cs", "biglessthan.r1cs"], ["BigMultModP", "BigLessThan"])
    st = E.frontend_stats()
    assert st["abstract_device"] == 1.0 and st["matched"] >= 1e6          # (+ 1e6 per window re-verified on the host)
    assert sh.specials() == sd.specials() and len(sh) == len(sd)
    _same_dict_rows("forced host verify", sh, sd)
    monkeypatch.delenv("ECNE_FE_FORCE_HOST_VERIFY")
    # (3) field size 33, section 1 first: the constraints start at an odd address
    rows3 = [([(2, 3)], [(3, 1)], [(4, 1), (1, 5)]), ([], [], [(4, 1), (2, P - 1)]), ([(3, 1), (1, P - 1)], [(3, 1)], [])]
    body2 = bytearray()
    for parts in rows3:
        for terms in parts:
            body2 += struct.pack("<I", len(terms))
            for v, c in terms:
                body2 += struct.pack("<I", v - 1) + int(c).to_bytes(32, "little")
    body1 = struct.pack("<I", 33) + P.to_bytes(33, "little") + struct.pack("<IIII", 3, 1, 1, 1) + struct.pack("<QI", 3, len(rows3))
    body3 = b"".join(struct.pack("<Q", i) for i in range(3))
    odd = str(tmp_path / "odd.r1cs")
    with open(odd, "wb") as f:
        f.write(b"r1cs" + struct.pack("<II", 1, 3) + struct.pack("<IQ", 1, len(body1)) + body1 + struct.pack("<IQ", 2, len(body2)) + bytes(body2) +
                struct.pack("<IQ", 3, len(body3)) + body3)
    assert orc.read_info(odd)[0] == 0
    mh, sh = _system(E, E.FRONTEND_HOST, path=odd)
    md, sd = _system(E, E.FRONTEND_DEVICE, path=odd)
    assert E.frontend_stats()["parse_device"] == 1.0 and list(mh.info.nnz) == list(md.info.nnz)
    _same_dict_rows("odd", sh, sd)
    _same_static_arrays("odd", sh, sd)
    # (4) no constraints at all
    empty = str(tmp_path / "empty.r1cs")
    r1cs_py.write(empty, 3, 1, 1, 1, [])
    mh, sh = _system(E, E.FRONTEND_HOST, path=empty)
    md, sd = _system(E, E.FRONTEND_DEVICE, path=empty)
    assert len(md) == 0 and len(sd) == 0
    from gpu_common import assert_bit_exact
    assert_bit_exact("empty", E.solve_batch([sd])[0], orc.run(empty))


def test_scratch_pool_reuse_across_files(tmp_path):
    """(round 6) the device front-end's scratch blocks are pooled between files (a block is handed out again for any request of a quarter of its
    size or more): files of very different sizes through the device front-end one after the other, back and forth, each time the same rows in
    dictionary order and the same static arrays as the host front-end's -- a stale word of somebody else's scratch would show"""
    import ecneproject_amd as E
    import ecdsa_like
    big = ecdsa_like.cached(3, 10)
    rels = ["ecne_circomlib_tests/EdDSAMiMCSpongeVerifier@eddsamimcsponge.r1cs", "target/division.r1cs", "secp256k1.r1cs",
            "ecne_circomlib_tests/Poseidon@poseidon.r1cs"]
    order = [big, rels[1], rels[0], big, rels[3], rels[2], rels[1], big, rels[0]]
    host = {}
    for it in order:
        path = it if os.path.isabs(it) else fixtures.path(it)
        if path not in host:
            host[path] = _system(E, 0, path=path)
        md, sd = _system(E, 1, path=path)
        assert E.frontend_stats()["parse_device"] == 1.0, path
        _same_dict_rows("pool " + os.path.basename(path), host[path][1], sd)
        _same_static_arrays("pool " + os.path.basename(path), host[path][1], sd)
