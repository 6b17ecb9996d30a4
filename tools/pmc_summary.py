#!/usr/bin/env python
"""Aggregate a rocprofv3 --pmc run (rocpd sqlite db) per kernel: mean counter value per dispatch."""
import sqlite3
import sys
import collections

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
if "counters_collection" in tabs:
    cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
    rows = cur.execute("select * from counters_collection")
    ik, ic, iv = cols.index("kernel_name") if "kernel_name" in cols else cols.index("name"), cols.index("counter_name"), cols.index("value")
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        agg[r[ik].split("(")[0][-60:]][r[ic]].append(r[iv])
    for k, d in sorted(agg.items(), key=lambda kv: -sum(sum(v) for v in kv[1].values())):
        n = max(len(v) for v in d.values())
        print("%-62s n=%-6d" % (k, n) + "  ".join("%s=%.4g" % (c, sum(v) / len(v)) for c, v in sorted(d.items())))
else:
    print("tables:", tabs)
