#!/usr/bin/env python
"""dev-time probe: the embedder's launches of ONE forward, in order, from a rocprofv3 --kernel-trace database of tools/bench_embed.py
usage: python tools/probes/embed_layers.py <results.db>"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, start, end, grid_x, grid_y from kernels order by start"))
heads = [i for i, r in enumerate(rows) if r[0].startswith("head_k")]
lo = heads[-2] + 1 if len(heads) >= 2 else 0
tot = 0.0
for n, s, e, gx, gy in rows[lo:heads[-1] + 1]:
    if n.startswith("__amd"): continue
    tot += (e - s) / 1e3
    print("%-28s grid %7d x %3d  %8.1f us" % (n.split("(")[0].replace("void ", "")[:28], gx, gy, (e - s) / 1e3))
print("sum %.1f us" % tot)
