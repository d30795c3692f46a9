"""Developer aid (GPU box): histogram of row shapes (k_classify_rows output) and row lengths per system."""
import os, sys, collections
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests"))
import numpy as np
import ecneproject_amd as E, fixtures
BITS = ["HAS_AB", "C_EMPTY", "R2", "R2_BERR", "R2_DIV0", "R2_IS01", "R3", "R4_T", "R4_T2", "R5", "R6", "SWAP", "P4", "P4_DIV0", "CZERO", "R7S", "C_HAS1", "TOUCH1", "BIG"]
IGN = (1 << 11) | (1 << 15) | (1 << 16) | (1 << 17) | (1 << 14) | (1 << 5)
def name(s):
    return "|".join(b for i, b in enumerate(BITS) if (s >> i) & 1) or "plain-linear"
for rel in sys.argv[1:]:
    tr, nm = ((["bigmultmodp.r1cs", "biglessthan.r1cs"], ["BigMultModP", "BigLessThan"]) if rel == "secp256k1.r1cs" else ([], []))
    s = E.System(E.R1CS(fixtures.path(rel)))
    for t, n in zip(tr, nm):
        s.abstract(E.R1CS(fixtures.path(t)), n)
    shape, ms, by = E.classify(s)
    lens = [np.diff(s.rows(p)[0].astype(np.int64)) for p in range(3)]
    tot = lens[0] + lens[1] + lens[2]
    h = collections.Counter((int(x) & ~IGN) for x in shape)
    print(rel, "rows", len(shape), "len total: mean %.2f max %d; >15: %d" % (tot.mean(), tot.max(), int((tot > 15).sum())))
    for k, c in h.most_common(12):
        m = (shape & ~np.uint32(IGN)) == k
        print("   %7d  %-40s meanlen %.1f maxlen %d" % (c, name(k), tot[m].mean(), tot[m].max()))
