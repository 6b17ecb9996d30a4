#!/usr/bin/env python
"""Print the kernel summary (name, calls, total ms, avg us, %) of a rocprofv3 rocpd database (…_results.db)."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
print("%-90s %8s %12s %10s %7s" % ("kernel", "calls", "total_ms", "avg_us", "pct"))
for name, calls, total, avg, pct in rows:
    short = name.split("(")[0][-88:] if not name.startswith("void at::") else "torch::" + name[10:70]
    print("%-90s %8d %12.3f %10.2f %6.2f%%" % (short, calls, total / 1e6 if total > 1e7 else total / 1e3, avg / 1e3 if total > 1e7 else avg, pct))
