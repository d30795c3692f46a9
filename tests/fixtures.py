"""Fixture access for the test-suite and bench.py.

The reference's `.r1cs` inputs (circom compiler outputs — binary data, not code) travel with the
repo as tests/data/**/*.r1cs.xz (27 MB raw -> 1.2 MB) because /root/reference does not exist on
the GPU box.  `path(rel)` materialises one of them in a scratch directory and returns its path.
"""
import lzma
import os
import tempfile

_HERE = os.path.dirname(os.path.abspath(__file__))
DATA = os.path.join(_HERE, "data")
_CACHE = os.path.join(tempfile.gettempdir(), "ecne_fixtures_%d" % os.getuid())


def path(rel):
    """rel is the path inside the reference tree, e.g. 'target/division.r1cs'."""
    src = os.path.join(DATA, rel)
    if os.path.exists(src):
        return src
    src_xz = src + ".xz"
    if not os.path.exists(src_xz):
        raise FileNotFoundError(rel)
    dst = os.path.join(_CACHE, rel)
    if not os.path.exists(dst) or os.path.getmtime(dst) < os.path.getmtime(src_xz):
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        with open(src_xz, "rb") as f:
            data = lzma.decompress(f.read())
        tmp = dst + ".tmp%d" % os.getpid()
        with open(tmp, "wb") as f:
            f.write(data)
        os.replace(tmp, dst)
    return dst


def all_r1cs():
    out = []
    for root, _d, files in os.walk(DATA):
        for fn in files:
            if fn.endswith(".r1cs.xz"):
                out.append(os.path.relpath(os.path.join(root, fn), DATA)[:-3])
    return sorted(out)


def circomlib_suite():
    """The 67 files of BASELINE.json config 4."""
    return [r for r in all_r1cs() if r.startswith("ecne_circomlib_tests/")]


# The reference's own asserted results (test/runtests.jl:4-36, README.md:95-107,
# examples/commitHasherTornadoCash.jl:4-6).  (relpath, trusted files, trusted names, secp_solve, verdict)
PED = ["tornadocash_circuits/Pedersen248@pedersen.r1cs", "tornadocash_circuits/Pedersen496@pedersen.r1cs"]
PED_NAMES = ["Pedersen248", "Pedersen496"]
REFERENCE_ASSERTED = [
    ("straightforward.r1cs", [], [], False, True),                       # runtests.jl:5
    ("trivial_mult.r1cs", [], [], False, True),                          # runtests.jl:9
    ("bigmult86_3.r1cs", [], [], False, True),                           # runtests.jl:13
    ("poseidon.r1cs", [], [], False, True),                              # runtests.jl:17
    ("multiplexer_33.r1cs", [], [], False, True),                        # runtests.jl:21
    ("tornadocash_circuits/commitHasher.r1cs", PED, PED_NAMES, False, True),   # runtests.jl:25
    ("tornadocash_circuits/merkleTree.r1cs", [], [], False, True),       # runtests.jl:26
    ("tornadocash_circuits/withdraw.r1cs", PED, PED_NAMES, False, True),  # runtests.jl:30
    ("secp256k1.r1cs", ["bigmultmodp.r1cs", "biglessthan.r1cs"], ["BigMultModP", "BigLessThan"],
     True, True),                                                       # runtests.jl:35
    ("target/division.r1cs", [], [], False, False),                      # README.md:95-107
]
