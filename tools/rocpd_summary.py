"""Summarise a rocprofv3 (ROCm 7.2 rocpd sqlite) result: per-kernel call count / total / average
duration, and per-kernel PMC counter sums when the run collected counters.
usage: python tools/rocpd_summary.py <results.db> [...]"""
import sqlite3
import sys


def table(cur, prefix):
    for (n,) in cur.execute("select name from sqlite_master where type='table'"):
        if n.startswith(prefix):
            return n
    return None


def summarise(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    kd, ks = table(cur, "rocpd_kernel_dispatch"), table(cur, "rocpd_info_kernel_symbol")
    cols = [r[1] for r in cur.execute("pragma table_info(%s)" % kd)]
    scols = [r[1] for r in cur.execute("pragma table_info(%s)" % ks)]
    name_col = "kernel_name" if "kernel_name" in scols else ("display_name" if "display_name" in scols else scols[-1])
    # (one line per kernel AND grid size: ecne_warmup launches both solve kernels once on a three-row system -- 1 024 threads -- and those
    #  launches must not be averaged into the solves of the workload)
    q = ("select s.%s, count(*), sum(d.end - d.start), avg(d.end - d.start), min(d.end - d.start), max(d.end - d.start), "
         "d.grid_size_x, max(d.workgroup_size_x) from %s d join %s s on d.kernel_id = s.id group by s.%s, d.grid_size_x order by 3 desc"
         % (name_col, kd, ks, name_col))
    rows = list(cur.execute(q))
    tot = sum(r[2] for r in rows) or 1
    print("# %s" % path)
    print("%-60s %6s %12s %12s %12s %12s %7s %9s %6s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct", "grid_x", "wg_x"))
    for n, c, t, a, mn, mx, gx, wx in rows:
        print("%-60s %6d %12.1f %12.1f %12.1f %12.1f %6.1f%% %9d %6d" % (n[:60], c, t / 1e3, a / 1e3, mn / 1e3, mx / 1e3, 100.0 * t / tot, gx, wx))
    pe, pi = table(cur, "rocpd_pmc_event"), table(cur, "rocpd_info_pmc")
    if pe and pi:
        pcols = [r[1] for r in cur.execute("pragma table_info(%s)" % pe)]
        icols = [r[1] for r in cur.execute("pragma table_info(%s)" % pi)]
        if "event_id" in pcols and "pmc_id" in pcols:
            # event_id -> kernel dispatch via rocpd_event? fall back to a plain per-counter, per-kernel sum
            try:
                q = ("select s.%s, i.name, count(*), sum(p.value), avg(p.value), d.grid_size_x from %s p join %s i on p.pmc_id = i.id "
                     "join %s d on p.event_id = d.event_id join %s s on d.kernel_id = s.id group by s.%s, i.name, d.grid_size_x order by 4 desc"
                     % (name_col, pe, pi, kd, ks, name_col))
                rows = list(cur.execute(q))
                if rows:
                    print("%-60s %-14s %6s %18s %18s %9s" % ("kernel", "counter", "n", "sum", "avg_per_dispatch", "grid_x"))
                    for n, cn, c, sm, av, gx in rows:
                        print("%-60s %-14s %6d %18.1f %18.1f %9d" % (n[:60], cn, c, sm, av, gx))
            except sqlite3.Error as e:
                print("pmc query failed:", e, pcols, icols)
    print()


if __name__ == "__main__":
    for p in sys.argv[1:]:
        summarise(p)
