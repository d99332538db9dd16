"""Idle time between consecutive kernels of ONE time step in a rocprofv3 --kernel-trace database:
    python tools/gaps.py <results.db> [marker-kernel-substring=k_rk4]
The step is the span between the last two launches of the marker kernel.  Prints the span, the busy and idle sums and the
(previous kernel -> next kernel) pairs that account for most of the idle time (host synchronisation shows up as 20+ us gaps)."""
import sqlite3, sys, collections
cur = sqlite3.connect(sys.argv[1]).cursor()
marker = sys.argv[2] if len(sys.argv) > 2 else "k_rk4"
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
gcol = "grid_x" if "grid_x" in cols else ("grid_size_x" if "grid_size_x" in cols else None)
rows = list(cur.execute("select name%s, start, end from kernels order by start" % ((" || ' grid=' || " + gcol) if gcol else "")))
if not gcol: print("columns:", cols)
marks = [k for k, r in enumerate(rows) if marker in r[0]]
a, b = marks[-2], marks[-1]
step = rows[a + 1:b + 1]
span = (step[-1][2] - step[0][1]) / 1e3
busy = sum(r[2] - r[1] for r in step) / 1e3
print("kernels %d  span %.1f us  busy %.1f us  idle %.1f us" % (len(step), span, busy, span - busy))
short = lambda n: n.split("(")[0].replace("void ", "")[:48]
pairs = collections.defaultdict(lambda: [0, 0.0])
hist = collections.Counter()
for p, q in zip(step[:-1], step[1:]):
    g = (q[1] - p[2]) / 1e3
    e = pairs[(short(p[0]), short(q[0]))]
    e[0] += 1; e[1] += g
    hist[min(int(g // 5) * 5, 100)] += 1
print("gap histogram (us bucket: count):", sorted(hist.items()))
for (p, q), (n, t) in sorted(pairs.items(), key=lambda kv: -kv[1][1])[:40]:
    print("%8.1f us %5d x %6.1f  %s -> %s" % (t, n, t / n, p, q))
print("\nkernels of the step by total time:")
agg = collections.defaultdict(lambda: [0, 0.0])
for r in step:
    nm = short(r[0]) if "<" not in r[0] else r[0].split("(")[0].replace("void ", "")[:60]
    if "k_mg_" in nm and "grid=" in r[0]: nm += " grid=" + r[0].split("grid=")[-1]
    e = agg[nm]
    e[0] += 1; e[1] += (r[2] - r[1]) / 1e3
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:70]:
    print("%9.1f us %5d x %8.2f  %s" % (t, c, t / c, n))
