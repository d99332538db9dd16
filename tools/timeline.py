"""Timeline of one preconditioner application from a rocprofv3 --kernel-trace database:
    python tools/timeline.py <results.db> [index of the k_prec_stage1_v2 launch to start from]
prints start / end (us relative to the stage-1 kernel), duration, queue and name of every kernel up to the next stage 1."""
import sqlite3, sys
db = sys.argv[1]; which = int(sys.argv[2]) if len(sys.argv) > 2 else 40
cur = sqlite3.connect(db).cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
view = "kernels" if "kernels" in tabs else [t for t in tabs if "kernel_dispatch" in t][0]
cols = [r[1] for r in cur.execute("pragma table_info(%s)" % view)]
print("#", view, cols, file=sys.stderr)
qcol = "queue_id" if "queue_id" in cols else ("queue" if "queue" in cols else None)
scol = "stream_id" if "stream_id" in cols else None
sel = "start, end, name" + (", " + qcol if qcol else "") + (", " + scol if scol else "")
rows = list(cur.execute("select %s from %s order by start" % (sel, view)))
st = [k for k, r in enumerate(rows) if "k_prec_stage1" in r[2]]
a = st[which]; b = st[which + 1]
t0 = rows[a][0]
for r in rows[a:b + 1]:
    nm = r[2].split("(")[0].replace("void ", "")[:60]
    print("%9.1f %9.1f %7.1f  q=%s  %s" % ((r[0] - t0) / 1e3, (r[1] - t0) / 1e3, (r[1] - r[0]) / 1e3, r[3:] if len(r) > 3 else "", nm))
