"""Developer aid (GPU box): one file through the device front-end, stats and comparison with the host front-end."""
import os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests"))
import numpy as np
import ecneproject_amd as E, fixtures
rel = sys.argv[1] if len(sys.argv) > 1 else "ecne_circomlib_tests/AliasCheck@aliascheck.r1cs"
p = rel if os.path.isabs(rel) else fixtures.path(rel)
print("devices", E.device_count(), "mode", E.set_frontend(1))
try:
    f = E.R1CS(p)
except Exception as e:
    print("load failed:", repr(e)); sys.exit(1)
print(E.frontend_stats())
s = E.System(f)
E.set_frontend(0)
fh = E.R1CS(p); sh = E.System(fh)
E.set_frontend(1)
print("nnz", list(f.info.nnz), list(fh.info.nnz))
for part in range(3):
    a, b = s.dict_rows(part), sh.dict_rows(part)
    for x, y, w in zip(a, b, ("ptr", "var", "coef")):
        ok = x.shape == y.shape and np.array_equal(x, y)
        print(part, w, "OK" if ok else "DIFF", x.shape, y.shape)
        if not ok and x.shape == y.shape:
            i = int(np.argmax((x != y).reshape(len(x), -1).any(axis=1))); print("  first diff at", i, x[i], y[i])
