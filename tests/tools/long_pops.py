"""Developer aid (CPU only): pops of long rows (more than 15 terms: no record line) in the sequential oracle's trace, per fixture --
how many there are, how many do anything (re-queue a variable), by shape class (linear / with A*B).
  python tests/tools/long_pops.py <fixture relpath> [trusted.r1cs:Name ...] """
import os, sys, tempfile
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE))); sys.path.insert(0, os.path.dirname(HERE))
import numpy as np
import ecneproject_amd as E, fixtures, orc

E.set_frontend(E.FRONTEND_HOST)
path = fixtures.path(sys.argv[1])
TR = [(fixtures.path(a.split(":")[0]), a.split(":")[1]) for a in sys.argv[2:]]
s = E.System(E.R1CS(path))
for tp, tn in sorted(TR, key=lambda x: -len(E.R1CS(x[0]))):
    s.abstract(E.R1CS(tp), tn)
parts = [s.rows(p) for p in range(3)]
n = len(s)
lens = np.zeros((3, n + 1), np.int64)
for p, (rp, col, _) in enumerate(parts):
    lens[p, 1:] = np.diff(rp[:n + 1])
tot = lens.sum(0)
tr = os.path.join(tempfile.gettempdir(), "ecne_trace_%d.bin" % os.getpid())
os.environ["ECNE_ORACLE_TRACE"] = tr
o = orc.run(path, [t[0] for t in TR], [t[1] for t in TR], secp_solve=True, want_states=False)
del os.environ["ECNE_ORACLE_TRACE"]
T = np.fromfile(tr, dtype=np.int64).reshape(-1, 2)
os.unlink(tr)
pop_idx = np.flatnonzero(T[:, 0] == 1)
rows = T[pop_idx, 1]
nxt = np.append(pop_idx[1:], len(T))
# a pop "does something" if a tag 2 (re-queued variable) record follows before the next pop / iteration mark
did = np.zeros(len(pop_idx), bool)
w = np.flatnonzero(T[:, 0] == 2)
did[np.unique(np.searchsorted(pop_idx, w, side="right") - 1)] = True
# (writes in P-phases follow a tag 0 record: exclude those)
it = np.flatnonzero(T[:, 0] == 0)
for i in it:
    k = np.searchsorted(pop_idx, i, side="right") - 1
    # writes after the iteration mark belong to the sweep, not to pop k: recompute pop k from records in front of the mark
    if k >= 0:
        seg = T[pop_idx[k] + 1:i]
        did[k] = bool((seg[:, 0] == 2).any())
L = tot[rows]
lin = (lens[0, rows] == 0) & (lens[1, rows] == 0)
print("rows", n, "pops", len(rows), "of rows > 15 terms:", int((L > 15).sum()), "(linear %d, of which active %d; with A*B %d, active %d)" % (
    int(((L > 15) & lin).sum()), int(((L > 15) & lin & did).sum()), int(((L > 15) & ~lin).sum()), int(((L > 15) & ~lin & did).sum())))
for lo, hi in ((16, 64), (65, 300), (301, 1 << 30)):
    m = (L >= lo) & (L <= hi)
    print("  %4d..%-6s terms: %5d pops of %4d rows, active %d" % (lo, hi if hi < 1 << 30 else "", int(m.sum()), len(np.unique(rows[m])), int((m & did).sum())))
