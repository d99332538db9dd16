"""Per-kernel summary (calls, total, average, share) of a rocprofv3 --kernel-trace --stats database as CSV on stdout:
    python tools/kernel_stats.py <results.db>      (the view reports microseconds)"""
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
print("kernel,calls,total_us,average_us,percent")
for name, calls, tot, avg, pct in cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
    print('"%s",%d,%.1f,%.3f,%.2f' % (name, calls, tot, avg, pct))
