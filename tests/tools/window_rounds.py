"""Developer aid (CPU only): how many ROUNDS a window schedule needs -- a round takes the first min(W, queued) entries of the FIFO and
commits the longest prefix in which no row reads a variable an earlier row of the round writes (decide-then-commit: only true
dependencies cut) -- from the sequential oracle's pop trace.   python tests/tools/window_rounds.py <fixture relpath> [W ...]"""
import collections, os, sys, tempfile
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE))); sys.path.insert(0, os.path.dirname(HERE))
import numpy as np
import ecneproject_amd as E, fixtures, orc

E.set_frontend(E.FRONTEND_HOST)
path = fixtures.path(sys.argv[1])
Ws = [int(a) for a in sys.argv[2:]] or [64]
s = E.System(E.R1CS(path))
parts = [s.rows(p) for p in range(3)]
n = len(s)
tr = os.path.join(tempfile.gettempdir(), "ecne_trace_%d.bin" % os.getpid())
os.environ["ECNE_ORACLE_TRACE"] = tr
o = orc.run(path, want_states=False)
del os.environ["ECNE_ORACLE_TRACE"]
T = np.fromfile(tr, dtype=np.int64).reshape(-1, 2)
os.unlink(tr)
rowvars = []
for r in range(n):
    vs = set()
    for rp, col, _ in parts:
        vs.update(col[rp[r]:rp[r + 1]].tolist())
    rowvars.append(vs)
# pops: list of (row, writes, npushed); phase breaks
pops = []; cur = None; phases = []
qlen0 = 0
events = []      # ('pop', row, writes, pushed) or ('phase', pushed)
cur_row = None; w = []; pushed = 0
def flush():
    global cur_row, w, pushed
    if cur_row is not None: events.append(("pop", cur_row, w, pushed))
    elif pushed or w: events.append(("phase", None, w, pushed))
    cur_row = None; w = []; pushed = 0
for tag, val in T.tolist():
    if tag == 0: flush(); events.append(("iter", None, [], 0))
    elif tag == 1: flush(); cur_row = val
    elif tag == 2: w.append(val)
    elif tag == 3: pushed += 1
flush()
# initial queue length = pops before any push accounted: derive by simulation (queue never underflows): start with q0 such that it works
npop = sum(1 for e in events if e[0] == "pop")
print("rows", n, "pops", o.summary.pops, npop)
for W in Ws:
    # queue length tracking: initial queue = rows with <= 1 unknown; count = total pops - total pushes
    tot_push = sum(e[3] for e in events)
    qlen = npop - tot_push
    rounds = 0; i = 0; hist = collections.Counter()
    evs = events
    k = 0
    while k < len(evs):
        e = evs[k]
        if e[0] != "pop":
            qlen += e[3]; k += 1; continue
        # a round starts here
        take = min(W, qlen)
        written = set(); c = 0
        while c < take and k < len(evs) and evs[k][0] == "pop":
            _t, row, wr, pu = evs[k]
            if c > 0 and (rowvars[row - 1] & written): break
            written.update(wr)
            qlen += pu - 1
            c += 1; k += 1
        rounds += 1; hist[min(c, 16)] += 1
    print("W", W, "rounds", rounds, "pops/round %.2f" % (npop / rounds), "hist(rows committed)", sorted(hist.items()))
