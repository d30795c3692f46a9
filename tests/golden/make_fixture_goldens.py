"""Writes tests/golden/fixture_goldens.json from the CPU oracle (runs anywhere; no GPU, no
/root/reference needed — inputs are tests/data/*.r1cs.xz).  The oracle itself is pinned to the
reference by tests/test_oracle_reference_results.py and tests/test_julia_order.py.
    python tests/golden/make_fixture_goldens.py
"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import numpy as np  # noqa: E402

import fixtures  # noqa: E402
import orc  # noqa: E402


def main():
    cases = [(rel, [], [], False) for rel in fixtures.all_r1cs()]
    cases += [("tornadocash_circuits/commitHasher.r1cs", fixtures.PED, fixtures.PED_NAMES, False),
              ("tornadocash_circuits/withdraw.r1cs", fixtures.PED, fixtures.PED_NAMES, False),
              ("secp256k1.r1cs", ["bigmultmodp.r1cs", "biglessthan.r1cs"], ["BigMultModP", "BigLessThan"], True),
              ("secp256k1.r1cs", ["bigmultmodp.r1cs", "biglessthan.r1cs"], ["BigMultModP", "BigLessThan"], False)]
    out = {}
    for rel, trusted, names, secp in cases:
        o = orc.run(fixtures.path(rel), [fixtures.path(t) for t in trusted], names, secp)
        key = rel + ("+" + "+".join(names) if names else "") + ("+secp" if secp else "")
        e = {"case": [rel, trusted, names, secp], "status": int(o.status)}
        if o.status == 0:
            s = o.summary
            e.update(verdict=bool(o.verdict), counts=[int(x) for x in o.counts()],
                     steps=int(s.successful_steps), outer=int(s.outer_iterations), pops=int(s.pops),
                     sha_flags=hashlib.sha256(np.ascontiguousarray(o.flags).tobytes()).hexdigest(),
                     sha_bounds_abz=hashlib.sha256(np.ascontiguousarray(o.lb).tobytes() + np.ascontiguousarray(o.ub).tobytes() +
                                                   np.ascontiguousarray(o.abz).tobytes()).hexdigest())
        out[key] = e
    with open(os.path.join(HERE, "fixture_goldens.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print(len(out), "goldens")


if __name__ == "__main__":
    main()
