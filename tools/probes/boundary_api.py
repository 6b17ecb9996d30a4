#!/usr/bin/env python
"""dev-time probe: what the host does while the GPU idles at a step boundary.  From a rocprofv3 --kernel-trace --hip-trace database: the
HIP API calls (thread, name, start, duration) between the last embedding kernel of a step and its first clustering kernel, and
between the step's last kernel and the next step's first one.   usage: python tools/probes/boundary_api.py <results.db>"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
cols = lambda t: [r[1] for r in db.execute("pragma table_info(%s)" % t)]
if len(sys.argv) > 2 and sys.argv[2] == "schema":
    for t in tabs:
        print(t, cols(t))
    sys.exit(0)
kern = list(db.execute("select name, start, end from kernels order by start"))
api_view = [t for t in ("regions", "hip_api", "api") if t in tabs]
if not api_view:
    print("no api view; tables:", tabs)
    sys.exit(1)
v = api_view[0]
c = cols(v)
tid = "tid" if "tid" in c else ([x for x in c if "thread" in x or x == "tid"] or ["0"])[0]
api = list(db.execute("select name, start, end, %s from %s order by start" % (tid, v)))
marks = [i for i, r in enumerate(kern) if "pair_tiles_k" in r[0]]
heads = [i for i, r in enumerate(kern) if r[0].startswith("head_k")]


def show(title, a, b):
    print("==== %s: %.3f ms" % (title, (b - a) / 1e6))
    for n, s, e, t in api:
        if e >= a and s <= b and (e - s) > 2000:
            print("  +%8.3f ms  %8.3f ms  tid %s  %s" % ((s - a) / 1e6, (e - s) / 1e6, t, n[:60]))


for m in marks[-2:]:
    h = max(i for i in heads if i < m)
    nxt = [r for r in kern[h + 1:m + 1] if not r[0].startswith("__amd_rocclr")]
    show("last head_k -> first clustering kernel (%s)" % nxt[0][0][:30], kern[h][2], nxt[0][1])
if len(marks) >= 2:
    m = marks[-2]
    last = max(i for i in range(m, len(kern)) if kern[i][1] < kern[m][1] + 5e6 and ("hac" in kern[i][0] or "pair" in kern[i][0] or "mirror" in kern[i][0]))
    nxt = kern[last + 1:last + 6]
    first = [r for r in nxt if "fill" in r[0].lower() or "resize" in r[0]][0]
    show("last clustering kernel -> next step's first detector op (%s)" % first[0][:30], kern[last][2], first[1])
