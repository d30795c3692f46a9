"""Developer aid (CPU only): ideal dataflow depth of a solve, from the sequential oracle's pop trace.
  python tests/tools/dataflow_depth.py [S [stride]]        (ecdsa_like(S, stride) + trusted secp256k1.r1cs)
  python tests/tools/dataflow_depth.py <fixture relpath> [trusted.r1cs:Name ...]
Every pop gets a level = 1 + max(level of the pop that pushed its row, level of the last writer of any variable it mentions
(RAW), level of the last reader of any variable it writes (WAR)); writes = the variables the pop re-queued. The number of
distinct levels is what a perfect level-synchronous schedule would need; the engine's rounds are prefixes of the FIFO order."""
import collections, os, sys, tempfile
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE))); sys.path.insert(0, os.path.dirname(HERE))
import numpy as np
import ecneproject_amd as E, ecdsa_like, fixtures, orc

E.set_frontend(E.FRONTEND_HOST)
if len(sys.argv) > 1 and not sys.argv[1].isdigit():      # a fixture (relative path) [+ trusted fixture:Name ...]
    path = fixtures.path(sys.argv[1])
    TR = [(fixtures.path(a.split(":")[0]), a.split(":")[1]) for a in sys.argv[2:]]
else:
    S_ = int(sys.argv[1]) if len(sys.argv) > 1 else 26
    stride = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    path = ecdsa_like.cached(S_, stride)
    TR = [(fixtures.path("secp256k1.r1cs"), "Secp256k1AddUnequal")]
s = E.System(E.R1CS(path))
for tp, tn in sorted(TR, key=lambda x: -len(E.R1CS(x[0]))):
    s.abstract(E.R1CS(tp), tn)
parts = [s.rows(p) for p in range(3)]
n = len(s)
tr = os.path.join(tempfile.gettempdir(), "ecne_trace_%d.bin" % os.getpid())
os.environ["ECNE_ORACLE_TRACE"] = tr
o = orc.run(path, [t[0] for t in TR], [t[1] for t in TR], secp_solve=True, want_states=False)
del os.environ["ECNE_ORACLE_TRACE"]
T = np.fromfile(tr, dtype=np.int64).reshape(-1, 2)
os.unlink(tr)
print("rows", n, "pops", o.summary.pops, "trace records", len(T))
rowvars = []
for r in range(n):
    vs = set()
    for rp, col, _ in parts:
        vs.update(col[rp[r]:rp[r + 1]].tolist())
    vs.discard(1)
    rowvars.append(tuple(vs))
nv = 1 + max(max(v) if v else 0 for v in rowvars) + 8
lastw = np.zeros(nv + 1, np.int64); lastr = np.zeros(nv + 1, np.int64)
ready = np.zeros(n + 1, np.int64)
base = 0; maxlev = 0
levels = collections.Counter(); per_iter = []
cur_row = None; cur_writes = []; cur_pushed = []
tags = T[:, 0].tolist(); vals = T[:, 1].tolist()

def finish():
    global maxlev
    if cur_row is None:          # P-phase pushes / writes: available from the iteration's base level
        for t in cur_pushed: ready[t] = base
        for v in cur_writes:
            if v <= nv: lastw[v] = max(lastw[v], base)
        return
    rv = rowvars[cur_row - 1]
    lv = max(base, ready[cur_row])
    for v in rv:
        if lastw[v] > lv: lv = lastw[v]
    for v in cur_writes:
        if v <= nv and lastr[v] > lv: lv = lastr[v]
    lv += 1
    for v in rv:
        if lastr[v] < lv: lastr[v] = lv
    for v in cur_writes:
        if v <= nv: lastw[v] = lv
    for t in cur_pushed: ready[t] = lv
    levels[lv] += 1
    if lv > maxlev: maxlev = lv

for tag, val in zip(tags, vals):
    if tag == 1:
        finish(); cur_row = val; cur_writes = []; cur_pushed = []
    elif tag == 2: cur_writes.append(val)
    elif tag == 3: cur_pushed.append(val)
    else:
        finish(); cur_row = None; cur_writes = []; cur_pushed = []
        per_iter.append(maxlev); base = maxlev
finish()
print("ideal levels:", maxlev, "(outer iterations %d)" % len(per_iter))
print("levels at the start of each outer iteration:", per_iter)
h = sorted(levels.items())
big = [(l, c) for l, c in h if c >= 1000]
print("levels with >= 1000 pops: %d holding %d pops; levels with < 64 pops: %d" % (len(big), sum(c for _, c in big), sum(1 for _, c in h if c < 64)))
print("first 60 levels:", h[:60])
