"""Median counter value per kernel from a rocprofv3 --pmc database of the 3-D bench:
    python tools/pmc3d.py <results.db> <COUNTER> [factor]      (FETCH_SIZE: factor 2 x 1024... see profiles/traffic.json provenance)"""
import sqlite3, sys, statistics
db, counter = sys.argv[1], sys.argv[2]
cur = sqlite3.connect(db).cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
view = "counters_collection" if "counters_collection" in tabs else [t for t in tabs if "counters_collection" in t][0]
acc = {}
for name, grid, val in cur.execute("select kernel_name, grid_size, value from %s where counter_name = ?" % view, (counter,)):
    acc.setdefault((name.split("(")[0][:60], grid), []).append(val)
for (name, grid), v in sorted(acc.items(), key=lambda kv: -statistics.median(kv[1]) * len(kv[1]))[:25]:
    print("%-62s grid %10d  n %5d  median %.4g" % (name, grid, len(v), statistics.median(v)))
