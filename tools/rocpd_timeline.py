"""Developer aid: the dispatches of a rocprofv3 (rocpd sqlite) kernel trace in time order -- start and end relative to the first one shown, stream / queue,
grid -- for the last N dispatches whose kernel name contains a pattern.   usage: python tools/rocpd_timeline.py <results.db> [pattern] [N]"""
import sqlite3
import sys


def table(cur, prefix):
    for (n,) in cur.execute("select name from sqlite_master where type='table'"):
        if n.startswith(prefix):
            return n
    return None


db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
pat = sys.argv[2] if len(sys.argv) > 2 else "k_solve"
N = int(sys.argv[3]) if len(sys.argv) > 3 else 12
kd, ks = table(cur, "rocpd_kernel_dispatch"), table(cur, "rocpd_info_kernel_symbol")
cols = [r[1] for r in cur.execute("pragma table_info(%s)" % kd)]
scols = [r[1] for r in cur.execute("pragma table_info(%s)" % ks)]
name_col = "kernel_name" if "kernel_name" in scols else ("display_name" if "display_name" in scols else scols[-1])
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
q = "select s.%s, d.start, d.end, d.grid_size_x%s from %s d join %s s on d.kernel_id = s.id where s.%s like ? order by d.start" % (
    name_col, (", d." + qcol) if qcol else "", kd, ks, name_col)
rows = list(cur.execute(q, ("%" + pat + "%",)))[-N:]
if rows:
    t0 = rows[0][1]
    for r in rows:
        print("%-40s start %10.1f us  end %10.1f us  dur %9.1f us  grid %8d  queue %s" % (r[0][:40], (r[1] - t0) / 1e3, (r[2] - t0) / 1e3, (r[2] - r[1]) / 1e3, r[3], r[4] if qcol else "-"))
