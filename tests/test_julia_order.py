"""Known-answer test for the Julia 1.7 Dict/Set iteration-order emulation (oracle/jldict.hpp).

Vectors come from the reference's own equation dumps and README transcript — see
tests/golden/make_julia_order_kat.py (16 620 vectors, rows of up to 90 terms, table growth
16 -> 64 -> 256 and real collision chains)."""
import json
import lzma
import os

import orc

HERE = os.path.dirname(os.path.abspath(__file__))


def _vectors():
    with open(os.path.join(HERE, "golden", "julia_order_kat.json.xz"), "rb") as f:
        return json.loads(lzma.decompress(f.read()))


def test_hash_64_64_spot_values():
    # Set order of a handful of small ints, cross-checked with README.md:103-105 (keys 6,2,8)
    assert orc.julia_order([2, 6, 8], 1) == [6, 2, 8]
    assert orc.julia_order([8, 6, 2], 1) == [6, 2, 8]


def test_reference_dumps():
    vecs = _vectors()
    assert len(vecs) == 16620
    bad = 0
    for kind, keys, expect in vecs:
        if kind == 0:
            got = orc.julia_order(keys, 1)                       # file order -> Dict -> Set
        elif kind == 1:
            got = orc.julia_order(orc.julia_order(keys, 2), 0)   # Dict(+macro var) -> Set
        else:
            got = orc.julia_order(orc.julia_order(orc.julia_order(keys, 2), 2), 0)  # Dict -> Dict -> Set
        bad += got != expect
    assert bad == 0
