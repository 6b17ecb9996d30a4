#!/usr/bin/env python
"""GPU idle time inside the last bench step, from a rocprofv3 kernel-trace database: where the gaps between kernels are and which
kernels stand on either side.  usage: python tools/gpu_gaps.py <results.db> [min_gap_us]"""
import sqlite3
import sys
import collections

db = sqlite3.connect(sys.argv[1])
min_gap = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
view = "kernels" if "kernels" in tabs else None
if view is None:
    print("no 'kernels' view; tables:", tabs[:40])
    sys.exit(1)
cols = [r[1] for r in db.execute("pragma table_info(%s)" % view)]
name_c = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = list(db.execute("select %s, start, end from %s order by start" % (name_c, view)))
# the last step = from the last pair_tiles_k backwards to the one before it
marks = [i for i, r in enumerate(rows) if r[0].startswith("pair_tiles_k") or "pair_tiles_k" in r[0]]
if len(marks) >= 2:
    lo, hi = marks[-2] + 1, marks[-1]
else:
    lo, hi = 0, len(rows) - 1
seg = rows[lo:hi + 1]
t0, t1 = seg[0][1], seg[-1][2]
busy = 0
cur_end = seg[0][1]
gaps = []
for i, (n, s, e) in enumerate(seg):
    if s > cur_end:
        gaps.append((s - cur_end, seg[i - 1][0], n, cur_end - t0))
    busy += max(0, e - max(s, cur_end))
    cur_end = max(cur_end, e)
span = t1 - t0
print("step span %.2f ms, GPU busy %.2f ms, idle %.2f ms (%d kernels)" % (span / 1e6, busy / 1e6, (span - busy) / 1e6, len(seg)))
short = lambda n: n.split("(")[0].replace("void ", "")[:34]
agg = collections.defaultdict(lambda: [0, 0.0])
for g, a, b, at in gaps:
    k = (short(a), short(b))
    agg[k][0] += 1; agg[k][1] += g
print("gaps by neighbouring kernels (total ms, count):")
for k, (c, tot) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    print("  %8.3f ms %5d  %s -> %s" % (tot / 1e6, c, k[0], k[1]))
print("largest gaps (us, at ms into the step):")
for g, a, b, at in sorted(gaps, key=lambda x: -x[0])[:15]:
    print("  %9.1f us at %7.2f ms  %s -> %s" % (g / 1e3, at / 1e6, short(a), short(b)))
