"""roctx ranges recorded by `rocprofv3 --marker-trace` (PVF_ROCTX=1): count and host-side span.  usage: roctx_regions.py <results.db>"""
import sqlite3
import sys
db = sqlite3.connect(sys.argv[1])
print("name, ranges, host ms inside")
for r in db.execute("select name, count(*), round(sum(end - start) / 1e6, 3) from regions group by name order by 3 desc limit 25"):
    print("  %-24s %5d %10.3f" % r)
