#!/bin/bash
# the roctx ranges of the kernel families under rocprofv3 --marker-trace: bash tools/roctx_check.sh <outdir>
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/$1; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/prof_rx
PVF_ROCTX=1 timeout 300 rocprofv3 --marker-trace --kernel-trace --stats -d /tmp/prof_rx -- python $R/bench.py --cpu-frames 0 --no-host-ingest --no-dropin --steps 1 --warmup 0 > $O/roctx_bench.log 2>&1
find /tmp/prof_rx -name "*marker*" | head -5
F=$(find /tmp/prof_rx -name "*marker_api_stats.csv" | head -1)
[ -n "$F" ] && cp $F $O/roctx_marker_stats.csv && cat $F | head -20
DB=$(find /tmp/prof_rx -name "*_results.db" | head -1)
python $R/tools/probes/roctx_regions.py "$DB" | tee $O/roctx_regions.txt; python - "$DB" <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
print([t for t in tabs if 'marker' in t.lower() or 'region' in t.lower()][:10])
PY
