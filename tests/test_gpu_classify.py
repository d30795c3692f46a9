"""k_classify_rows' shape words against a CPU classifier written from the reference's pattern tests (tests/classify_ref.py):
every fixture of the reference tree and the seeded fuzz systems (explicit zeros, repeated wire ids, un-reduced
coefficients, degenerate rows). SURVEY.md §7 S6 / VERDICT r1 item 3."""
import numpy as np
import pytest

import classify_ref
import fixtures
import fuzz_r1cs
import r1cs_py

pytestmark = pytest.mark.gpu


def _check(path, tag):
    import ecneproject_amd as E
    hdr, rows = r1cs_py.parse_file(path)
    s = E.System(E.R1CS(path))
    shape, ms, nbytes = E.classify(s)
    assert len(shape) == len(rows)
    want = np.array(classify_ref.classify(rows), dtype=np.uint32)
    got = shape & np.uint32(classify_ref.CHECKED)
    bad = np.nonzero(got != want)[0]
    assert len(bad) == 0, (tag, int(bad[0]), hex(int(got[bad[0]])), hex(int(want[bad[0]])), rows[int(bad[0])])


@pytest.mark.parametrize("rel", [r for r in fixtures.all_r1cs()])
def test_shape_words_every_fixture(rel):
    _check(fixtures.path(rel), rel)


def test_shape_words_fuzz(tmp_path):
    for seed in range(200):
        p = str(tmp_path / ("%d.r1cs" % seed))
        fuzz_r1cs.write(p, fuzz_r1cs.make(seed))
        _check(p, "fuzz %d" % seed)
    for seed in range(40):
        p = str(tmp_path / ("w%d.r1cs" % seed))
        fuzz_r1cs.write(p, fuzz_r1cs.make_wide(seed))
        _check(p, "wide %d" % seed)
