"""Generates tests/golden/julia_order_kat.json.xz from the reference's own equation dumps.

Run in the build container (needs /root/reference):  python tests/golden/make_julia_order_kat.py

Source of truth: Circom_Functions/benchmarks/{bigmod_5_2,bigmod_10_2,bigmod_86_3,bigmult_86_3}.txt
are printEquation dumps written by the reference itself (scripts/create_files.sh:1-7,
src/util/gen_benchmark.jl:22-24 via R1CSUnOptimize, src/Auxiliary.jl:187-250).  Each printed row
lists its terms in the iteration order of a Julia `Set{Any}` built by nonzeroKeys(); the input is
the row's keys in .r1cs file order.  Three vector kinds are extracted:
  kind 0  linear row, passes through unchanged:     file order -> Dict -> nonzeroKeys Set
  kind 1  new_eq_1 of a non-linear row's part:      file order + appended macro var -> Dict -> Set
  kind 2  new_eq_2 (flip_keys copy of that Dict):   ... -> Dict -> Dict -> Set
plus README.md:103-105 (row #3 of target/division.r1cs printed as x4, out, y2 = keys 6, 2, 8).
Only integer key lists are stored (data, no reference text).
"""
import json
import lzma
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import r1cs_py  # noqa: E402

REF = "/root/reference/"
B = REF + "Circom_Functions/benchmarks/"


def main():
    vecs = []
    for name in ["bigmod_5_2", "bigmod_10_2", "bigmod_86_3", "bigmult_86_3"]:
        _hdr, rows = r1cs_py.parse_file(B + name + ".r1cs")
        lines = open(B + name + ".txt").read().split("\n")[4:]
        li = 0
        for A, Bp, Cp in rows:
            if not A and not Bp:
                got = [int(x) for x in re.findall(r"x_\{(\d+)\}", lines[li].split(" = ", 1)[1])]
                li += 1
                vecs.append([0, [v for v, _c in Cp], got])
                continue
            macro = []
            while lines[li].startswith("0 * 0 = "):
                macro.append(lines[li])
                li += 1
            li += 1
            for j in range(0, len(macro), 2):
                k1 = [int(x) for x in re.findall(r"x_\{(\d+)\}", macro[j])]
                k2 = [int(x) for x in re.findall(r"x_\{(\d+)\}", macro[j + 1])]
                cv = max(k1)
                for part in (A, Bp):
                    if set(v for v, _c in part) == set(k1) - {cv}:
                        base = [v for v, _c in part] + [cv]
                        vecs.append([1, base, k1])
                        vecs.append([2, base, k2])
                        break
    _hdr, rows = r1cs_py.parse_file(REF + "target/division.r1cs")
    vecs.append([0, [v for v, _c in rows[2][2]], [6, 2, 8]])   # README.md:103-105
    out = os.path.join(HERE, "julia_order_kat.json.xz")
    with open(out, "wb") as f:
        f.write(lzma.compress(json.dumps(vecs, separators=(",", ":")).encode(), preset=9))
    print(len(vecs), "vectors ->", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
