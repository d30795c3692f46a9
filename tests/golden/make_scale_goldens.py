"""Writes tests/golden/scale_goldens.json: the CPU oracle's result on the million-row cases the `-m gpu` suite cannot afford to run the
oracle on every time -- ecdsa_like(104, 10) (4.4 M rows, the one configuration beyond the 256 MiB Infinity Cache; 3-7 minutes of oracle)
and 1 400 copies of Poseidon side by side -- as counters plus the digest of the WHOLE per-variable state (tests/state_digest.py, the numpy
restatement of ecne_result_digest). The inputs are generated (tests/ecdsa_like.py, tests/multi_copy.py; no RNG), so the vectors
reproduce anywhere:
    python tests/golden/make_scale_goldens.py            # ~10 minutes, 6 GB of memory, no GPU, no /root/reference
tests/test_gpu_ecdsa_like.py::test_scale_out_full_state_parity compares the engine's device-side digest and counters with them;
ECNE_FULL_ORACLE=2 still runs the oracle itself next to it (and checks this file against it)."""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import numpy as np  # noqa: E402

import ecdsa_like  # noqa: E402
import fixtures  # noqa: E402
import multi_copy  # noqa: E402
import orc  # noqa: E402
from state_digest import numpy_digest  # noqa: E402

CASES = {
    "ecdsa_like(104,10)+Secp256k1AddUnequal": lambda: (ecdsa_like.cached(104, 10), [fixtures.path("secp256k1.r1cs")], ["Secp256k1AddUnequal"]),
    # S = 416 (17.6 M rows, 2.3 GB file): the last point of SURVEY.md 8(d)'s scale-out series {26, 104, 416}; ~25 GB of memory for the oracle
    "ecdsa_like(416,10)+Secp256k1AddUnequal": lambda: (ecdsa_like.cached(416, 10), [fixtures.path("secp256k1.r1cs")], ["Secp256k1AddUnequal"]),
    "ecdsa_like(26,10)+Secp256k1AddUnequal": lambda: (ecdsa_like.cached(26, 10), [fixtures.path("secp256k1.r1cs")], ["Secp256k1AddUnequal"]),
    "1400xPoseidon@poseidon": lambda: (multi_copy.cached("ecne_circomlib_tests/Poseidon@poseidon.r1cs", 1400), [], []),
}


def entry(o):
    s = o.summary
    d = numpy_digest(o)
    return dict(status=int(o.status), verdict=bool(o.verdict), counts=[int(x) for x in o.counts()], n_vars=int(s.n_vars),
                rows_reduced=int(s.n_rows_reduced), steps=int(s.successful_steps), num_unique=int(s.num_unique), outer=int(s.outer_iterations),
                pops=int(s.pops), rule_hits=[int(x) for x in s.rule_hits[:13]], n_bad_rows=int(len(o.bad_rows)),
                sha_bad_rows=hashlib.sha256(np.ascontiguousarray(o.bad_rows, dtype=np.int64).tobytes()).hexdigest(),
                digest=["%016x" % d[0], "%016x" % d[1]])


def main():
    only = sys.argv[1:]
    path = os.path.join(HERE, "scale_goldens.json")
    out = json.load(open(path)) if os.path.exists(path) else {}
    for key, mk in CASES.items():
        if only and key not in only:
            continue
        p, tp, names = mk()
        o = orc.run(p, tp, names)
        assert o.status == 0
        out[key] = entry(o)
        print(key, out[key]["digest"], out[key]["pops"], flush=True)
        with open(path, "w") as f:
            json.dump(out, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
