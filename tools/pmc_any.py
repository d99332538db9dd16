"""Median of every counter per kernel from a rocprofv3 --pmc database:  python tools/pmc_any.py <results.db> [kernel substring]"""
import sqlite3, sys, statistics
db = sys.argv[1]; sub = sys.argv[2] if len(sys.argv) > 2 else ""
cur = sqlite3.connect(db).cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
view = "counters_collection" if "counters_collection" in tabs else [t for t in tabs if "counters_collection" in t][0]
acc = {}
cols = [r[1] for r in cur.execute("pragma table_info(%s)" % view)]
gcol = "grid_size" if "grid_size" in cols else ("grid_size_x" if "grid_size_x" in cols else ("grid_x" if "grid_x" in cols else None))
bygrid = len(sys.argv) > 3 and gcol          # third argument: one line per (kernel, grid size)
if len(sys.argv) > 3 and not gcol: print("no grid column among", cols)
for name, cname, val, gs in cur.execute("select kernel_name, counter_name, value, %s from %s" % (gcol if bygrid else "0", view)):
    if sub in name:
        acc.setdefault((name.split("(")[0][:58] + (" g=%s" % gs if bygrid else ""), cname), []).append(val)
for (name, cname), v in sorted(acc.items()):
    print("%-72s %-28s n %4d  median %.6g" % (name, cname, len(v), statistics.median(v)))
