"""CPU-side checks of the product: the C-ABI library loads and exports every symbol include/ecne.h
declares, the native reader and the host abstraction agree with the oracle, and every solve entry
point refuses loudly when there is no GPU (no CPU fallback).  No compute calls."""
import os
import re

import numpy as np
import pytest

import fixtures
import orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def E():
    from ecneproject_amd import build
    build.build()
    import ecneproject_amd
    return ecneproject_amd


def test_exports_match_header(E):
    from ecneproject_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "ecne.h")).read()
    declared = set(re.findall(r"\b(ecne_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_lib.EXPORTS)
    lib = _lib.lib()
    for sym in declared:
        assert getattr(lib, sym) is not None
    assert lib.ecne_strerror(-8).decode().startswith("no usable HIP device")


def test_julia_binding_names_only_exported_symbols(E):
    """julia/EcneHIP.jl (the redirect a maintainer of the reference loads; never executed here: no Julia in the image) binds by NAME:
    every `ccall((:name, LIB), ...)` has to be a symbol include/ecne.h declares and the library exports, and the structs it mirrors have to
    have the library's sizes"""
    import ctypes
    import re
    from ecneproject_amd import _lib
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "julia", "EcneHIP.jl")).read()
    names = set(re.findall(r"ccall\(\(:(ecne_[a-z_0-9]+), LIB\)", src))
    assert len(names) >= 20
    assert names <= set(_lib.EXPORTS), sorted(names - set(_lib.EXPORTS))
    lib = _lib.lib()
    for n in names:
        assert hasattr(lib, n), n
    m = re.search(r"team::NTuple\{4,Int64\}", src)
    assert m, "EcneSummary in the Julia binding has to end with ecne_summary's team[4]"
    assert ctypes.sizeof(E.Summary) % 8 == 0


def test_reader_matches_oracle_on_every_fixture(E):
    for rel in fixtures.all_r1cs():
        f, kn, out, nv = E.readR1CS(fixtures.path(rel))
        st, d = orc.read_info(fixtures.path(rel))
        assert st == 0
        i = f.info
        assert (i.n_wires, i.n_pub_out, i.n_pub_in, i.n_prv_in, i.n_constraints, i.n_vars) == \
            (d["nWires"], d["nPubOut"], d["nPubIn"], d["nPrvIn"], d["nConstraints"], d["nVars"])
        assert list(i.nnz) == d["nnz"]
        assert len(kn) == d["n_knowns"] and len(out) == d["n_outputs"]
        assert kn[:len(d["knowns"])] == d["knowns"] and out[:len(d["outputs"])] == d["outputs"]


def test_csr_view_matches_python_parser(E):
    import r1cs_py
    rel = "ecne_circomlib_tests/Multiplexer@multiplexer.r1cs"
    f = E.R1CS(fixtures.path(rel))
    _hdr, rows = r1cs_py.parse_file(fixtures.path(rel))
    for part in range(3):
        rp, col, coef = f.csr(part)
        k = 0
        for r, parts in enumerate(rows):
            terms = [(v, c) for v, c in parts[part] if c]
            assert rp[r + 1] - rp[r] == len(terms)
            for v, c in terms:
                assert col[k] == v and orc.limbs_to_int(coef[k]) == c
                k += 1


def test_format_errors(E, tmp_path):
    p = tmp_path / "bad.r1cs"
    p.write_bytes(b"r1cs" + (2).to_bytes(4, "little") + (3).to_bytes(4, "little"))   # version 2 (:58)
    with pytest.raises(AssertionError):
        E.R1CS(str(p))
    assert orc.read_info(str(p))[0] == -1
    p.write_bytes(b"r1cs" + (1).to_bytes(4, "little") + (4).to_bytes(4, "little"))   # 4 sections (:62)
    with pytest.raises(AssertionError):
        E.R1CS(str(p))
    with pytest.raises(OSError):
        E.R1CS(str(tmp_path / "missing.r1cs"))


ABSTRACTION_CASES = [
    ("secp256k1.r1cs", ["bigmultmodp.r1cs", "biglessthan.r1cs"], ["BigMultModP", "BigLessThan"]),
    ("tornadocash_circuits/commitHasher.r1cs", fixtures.PED, fixtures.PED_NAMES),
    ("tornadocash_circuits/withdraw.r1cs", fixtures.PED, fixtures.PED_NAMES),
    ("bigmultmodp86_3.r1cs", ["bigmultshortlong86_3.r1cs"], ["BigMultShortLong"]),   # bench/bench_abstraction.jl:13-18
    ("secp256k1.r1cs", ["biglessthan.r1cs"], ["BigLessThan"]),
]


@pytest.mark.parametrize("rel,trusted,names", ABSTRACTION_CASES, ids=[c[0] + "<-" + "+".join(c[2]) for c in ABSTRACTION_CASES])
def test_host_abstraction_matches_oracle(E, rel, trusted, names):
    main = E.R1CS(fixtures.path(rel))
    fl = sorted([(n, E.R1CS(fixtures.path(t))) for t, n in zip(trusted, names)], key=lambda x: -len(x[1]))
    s = E.System(main)
    for n, f in fl:
        s.abstract(f, n)
    o = orc.run(fixtures.path(rel), [fixtures.path(t) for t in trusted], names, True, want_states=False)
    assert len(s) == o.summary.n_rows_reduced
    assert s.specials() == o.specials
    assert s.info.n_rows_main == o.summary.n_rows_main


def test_ecdsa_like_abstraction(E):
    import ecdsa_like
    path = ecdsa_like.cached(3, 3)
    s = E.System(E.R1CS(path))
    s.abstract(E.R1CS(fixtures.path("secp256k1.r1cs")), "Secp256k1AddUnequal")
    o = orc.run(path, [fixtures.path("secp256k1.r1cs")], ["Secp256k1AddUnequal"], want_states=False)
    assert s.specials() == o.specials and len(s.specials()) == 2
    assert len(s) == o.summary.n_rows_reduced == 1056


def test_no_gpu_means_loud_failure(E):
    if E.device_count() > 0:
        pytest.skip("a GPU is visible")
    s = E.System(E.R1CS(fixtures.path("target/division.r1cs")))
    with pytest.raises(E.NoDeviceError):
        E.solve_batch([s])
    with pytest.raises(E.NoDeviceError):
        E.solveWithTrustedFunctions(fixtures.path("target/division.r1cs"), "division", printRes=False)
    with pytest.raises(E.NoDeviceError):
        E.classify(s)


def test_product_does_not_reference_the_oracle():
    """the shipped package must not import, link or execute anything under oracle/"""
    pkg = os.path.join(ROOT, "ecneproject_amd")
    for root, _d, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hpp", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(root, fn), errors="ignore").read()
                assert "libecne_oracle" not in txt and "import orc" not in txt and "oracle/" not in txt.replace("test oracle (oracle/jldict.hpp)", ""), fn
    # the developer aids under tools/ are not checkers either (the ones that are live in tests/tools/)
    for root, _d, files in os.walk(os.path.join(ROOT, "tools")):
        for fn in files:
            if fn.endswith((".py", ".sh", ".hip", ".cpp")):
                txt = open(os.path.join(root, fn), errors="ignore").read()
                assert "libecne_oracle" not in txt and "import orc" not in txt and "orc." not in txt, fn


def test_field_sqrt_utility(E):
    """src/Math.jl is dead code in the reference; self-consistency only (v*v == a)."""
    import ctypes as C
    from ecneproject_amd import _lib
    lib = _lib.lib()
    P = orc.P
    for a in [4, 9, 2, 3, 5, P - 1, 123456789 ** 2 % P]:
        x = orc.int_to_limbs(a)
        r = np.zeros(4, np.uint64)
        ok = lib.ecne_fp_sqrt(x.ctypes.data, r.ctypes.data)
        is_res = pow(a, (P - 1) // 2, P) == 1
        assert bool(ok) == is_res
        if ok:
            assert orc.limbs_to_int(r) ** 2 % P == a


def test_field_solve_quadratic_utility(E):
    """solveQuadratic (src/Math.jl:62-90, dead code upstream): the kinds of answer, the true roots by substitution, and the
    reference's two-value formula as written (the discriminant where its square root was meant, :83-84) against Python."""
    import random
    from ecneproject_amd import _lib
    lib = _lib.lib()
    P = orc.P
    rng = random.Random(11)

    def solve(a, b, c, literal):
        roots, n = np.zeros(8, np.uint64), C.c_int(0)
        la, lb, lc = orc.int_to_limbs(a), orc.int_to_limbs(b), orc.int_to_limbs(c)       # (kept alive across the call)
        kind = lib.ecne_fp_solve_quadratic(la.ctypes.data, lb.ctypes.data, lc.ctypes.data, literal, roots.ctypes.data, C.byref(n))
        return kind, [orc.limbs_to_int(roots[4 * i:4 * i + 4]) for i in range(n.value)]
    import ctypes as C
    assert solve(0, 0, 0, 0) == (0, []) and solve(0, 0, 5, 0) == (1, [])                 # "YES" / "NO"
    assert solve(0, 3, 6, 0) == (2, [(-6 * pow(3, -1, P)) % P])
    seen = set()
    for _ in range(60):
        a, r1, r2 = rng.randrange(1, P), rng.randrange(P), rng.randrange(P)
        if rng.random() < 0.2:
            r2 = r1                                                                        # double root
        b, c = (-a * (r1 + r2)) % P, (a * r1 * r2) % P
        kind, roots = solve(a, b, c, 0)
        seen.add(kind)
        if r1 == r2:
            assert kind == 4 and roots == [r1]
        else:
            assert kind == 3 and sorted(roots) == sorted([r1, r2])
            assert all((a * x * x + b * x + c) % P == 0 for x in roots)
            disc = (b * b - 4 * a * c) % P
            inv2a = pow(2 * a, -1, P)
            assert solve(a, b, c, 1) == (3, [((-b + disc) * inv2a) % P, ((-b - disc) * inv2a) % P])
    # a discriminant without a square root: the reference's squareRoot would not terminate
    n_none = 0
    for _ in range(20):
        a, b, c = rng.randrange(1, P), rng.randrange(P), rng.randrange(P)
        disc = (b * b - 4 * a * c) % P
        if disc and pow(disc, (P - 1) // 2, P) != 1:
            assert solve(a, b, c, 0)[0] == 5
            n_none += 1
    assert seen == {3, 4} and n_none > 3


_DIGEST_SCRIPT = r"""
import hashlib, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[2])
import ecneproject_amd as E, fixtures, ecdsa_like
s = E.System(E.R1CS(ecdsa_like.cached(4, 10)))
h = hashlib.sha256()
for stage in range(2):
    for part in range(3):
        for a in s.rows(part):
            h.update(a.tobytes())
    h.update(repr(s.specials()).encode())
    if stage == 0:
        s.abstract(E.R1CS(fixtures.path("secp256k1.r1cs")), "Secp256k1AddUnequal")
print(h.hexdigest(), len(s), len(s.specials()))
"""


def test_host_worker_threads_do_not_change_results():
    """reader, abstraction and copy run on ECNE_HOST_THREADS workers; rows, their order and the specials of
    ecdsa_like(4) (168 k rows, three Secp256k1AddUnequal instances) are the same with 1, 3 and 8 of them"""
    import subprocess
    import sys
    from ecneproject_amd import build
    build.build()
    here = os.path.dirname(os.path.abspath(__file__))
    outs = []
    for n in ("1", "3", "8"):
        env = dict(os.environ, ECNE_HOST_THREADS=n)
        outs.append(subprocess.check_output([sys.executable, "-c", _DIGEST_SCRIPT, os.path.dirname(here), here], env=env, timeout=600).decode().split())
    assert outs[0] == outs[1] == outs[2], outs
    assert int(outs[0][2]) == 3
