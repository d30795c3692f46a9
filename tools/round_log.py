"""Developer aid: summary of a per-round log of one solve.
  ECNE_BUILD_FLAGS=-DECNE_ROUNDLOG python -m ecneproject_amd.build --force      (one printf per round of the master workgroup)
  python tools/solve_case.py ecdsa 0 > log.txt     (GPU box; the printfs inflate the solve time, the per-round ticks stay usable)
  python tools/round_log.py log.txt [--seq]
Lines: `RL <kind> avail A n N c C dt T` -- kind = wave (fast wavefront round), multi (round on all workgroups), wg (general
workgroup round), burst (sequential pops), alone (long row popped by the workgroup), solo (drain round on the master alone); multi / solo
lines also carry `levels L team K` (drain levels, workgroups on the team); A rows queued, N examined, C committed, T 100 MHz ticks. solve_case.py solves three times: the last solve is summarised."""
import collections, sys

path = [a for a in sys.argv[1:] if not a.startswith("--")][0]
import re
PAT = re.compile(r"^RL (\w+) avail (\d+) n (\d+) c (\d+) dt (\d+)(?: levels (\d+) team (\d+))?")
L = []
for l in open(path, errors="replace"):
    m = PAT.match(l)
    if m:
        f = ["RL", m.group(1), "avail", m.group(2), "n", m.group(3), "c", m.group(4), "dt", m.group(5)]
        if m.group(6) is not None:
            f += ["levels", m.group(6), "team", m.group(7)]
        L.append(f)
L = L[-(len(L) // 3):] if len(L) >= 3 else L
tot, cnt, rows = collections.Counter(), collections.Counter(), collections.Counter()
for l in L:
    k, c, dt = l[1], int(l[7]), int(l[9])
    tot[k] += dt; cnt[k] += 1; rows[k] += c
print("%d rounds, %.2f ms" % (len(L), sum(tot.values()) * 1e-5))
for k in tot:
    print("  %-6s %5d rounds %9d rows %8.2f ms %7.1f us/round" % (k, cnt[k], rows[k], tot[k] * 1e-5, tot[k] / cnt[k] * 1e-2))
for kind, edges in (("wave", (1, 2, 4, 8, 16, 32, 63, 64)), ("multi", (63, 1023, 4095, 16383, 1 << 30))):
    h, ht = collections.Counter(), collections.Counter()
    for l in L:
        if l[1] != kind:
            continue
        c, dt = int(l[7]), int(l[9])
        b = next(e for e in edges if c <= e)
        h[b] += 1; ht[b] += dt
    for b in sorted(h):
        print("  %-5s committed <= %-10d %4d rounds %7.2f ms %6.1f us/round" % (kind, b, h[b], ht[b] * 1e-5, ht[b] / h[b] * 1e-2))
if "--seq" in sys.argv:
    print(" | ".join("%s a%d n%d c%d %.0fus%s" % (l[1][:2], int(l[3]), int(l[5]), int(l[7]), int(l[9]) * 1e-2, (" L%s K%s" % (l[11], l[13])) if len(l) >= 14 else "") for l in L))
