"""A CPU classifier of rank-1 rows written from the reference's pattern tests, for checking k_classify_rows' shape words
directly (test_gpu_classify.py). Pure Python on the raw rows of tests/r1cs_py.parse (file order, explicit zeros and
repeated wire ids kept), so it shares nothing with the engine's host layout or with the oracle.

Bit names as in ecneproject_amd/csrc/engine_types.hpp. Reference lines (src/R1CSConstraintSolver.jl): nonzeroKeys :26-34,
R2 :875-927, R3 :949-960, R4 :991-1013, R5 :1078-1085, R6 :1148-1162, P4 :1427-1466; ParseR1CS.jl:108-115 for the maps."""
P = 21888242871839275222246405745257275088548364400416034343698204186575808495617

SH = dict(HAS_AB=1 << 0, C_EMPTY=1 << 1, R2=1 << 2, R2_BOUNDSERR=1 << 3, R2_DIV0=1 << 4, R2_IS01=1 << 5, R3=1 << 6, R4_T=1 << 7,
          R4_T2=1 << 8, R5=1 << 9, R6=1 << 10, P4=1 << 12, P4_DIV0=1 << 13, CZERO=1 << 14, C_HAS1=1 << 16, BIG=1 << 18)
CHECKED = sum(SH.values())
SMALL_ROW = 64


def part_map(terms):
    """the DefaultDict readR1CS builds for one part: file order, a repeated wire id overwrites the value in place; an
    empty part is stored as {1 => 0} (ParseR1CS.jl:113-115)"""
    d = {}
    for v, c in terms:
        d[v] = c % P
    if not terms:
        d[1] = 0
    return d


def classify_row(parts):
    a, b, c = (part_map(t) for t in parts)
    nz = [[k for k, v in d.items() if v != 0] for d in (a, b, c)]
    s = 0
    if nz[0] or nz[1]:
        s |= SH["HAS_AB"]
    if 1 in nz[2]:
        s |= SH["C_HAS1"]
    if len(nz[0]) + len(nz[1]) + len(nz[2]) > SMALL_ROW:
        s |= SH["BIG"]
    if not nz[2]:
        s |= SH["C_EMPTY"]
        others = sorted(set(nz[0]) | set(nz[1]) - {1})
        others = [k for k in others if k != 1]
        if len(others) == 0:
            s |= SH["R2_BOUNDSERR"]                     # variable_states[-1] (:916)
        elif len(others) == 1:
            x = others[0]
            s |= SH["R2"]
            if a.get(x, 0) == 0 or b.get(x, 0) == 0:
                s |= SH["R2_DIV0"]                      # divexact(-a[1], a[x]) with a[x] == 0 (:919-920)
            else:
                r1 = (-a.get(1, 0)) * pow(a[x], -1, P) % P
                r2 = (-b.get(1, 0)) * pow(b[x], -1, P) % P
                if sorted((r1, r2)) == [0, 1]:
                    s |= SH["R2_IS01"]                  # (:923-927)
        if len(nz[1]) == 1 and len(nz[0]) <= 2:       # P4's static tests (:1427-1453)
            s |= SH["P4"]
            if not [k for k in nz[0] if k != 1]:
                s |= SH["P4_DIV0"]                      # slope stays F(0) (:1467)
    if not (s & SH["HAS_AB"]) and nz[2]:
        non_one = [k for k in nz[2] if k != 1]
        if len(non_one) == 1:
            s |= SH["R3"]                               # (:949-960); reading c[1] inserts 1 => 0 when absent (:962)
        values = sorted(c.values())
        if (s & SH["R3"]) and 1 not in c:
            values = sorted(values + [0])
        if 0 in values:
            s |= SH["CZERO"]
        l = len(nz[2])
        if values == sorted([1] + [(-pow(2, i, P)) % P for i in range(l - 1)]):
            s |= SH["R4_T"]                             # (:999, :1013)
        if values == sorted([P - 1] + [pow(2, i, P) for i in range(l - 1)]):
            s |= SH["R4_T2"]                            # (:1000-1011)
        if l < 3 and values == [1, P - 1]:
            s |= SH["R5"]                               # (:1079-1085)
        if l < 4 and values == [1, P - 1, P - 1] and all(k == 1 for k, v in c.items() if v == 1):
            s |= SH["R6"]                               # (:1149-1162)
    return s


def classify(rows):
    return [classify_row(r) for r in rows]
