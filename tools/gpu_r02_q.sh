#!/bin/bash
# dev-time GPU session: host-side timeline of one unprofiled bench run (pipeline thread notes)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=gpurun_out/r02q; mkdir -p $O
PVF_TRACE=$R/$O/trace.json timeout 300 python bench.py --steps 2 --warmup 1 --cpu-frames 0 --no-host-ingest > $O/bench.log 2>&1
python - <<'PY'
import json
tr = json.load(open("gpurun_out/r02q/trace.json"))
t0 = tr[0][0]
for e in tr: print("%8.2f ms  %s" % ((e[0]-t0)*1e3, " ".join(str(x) for x in e[1:])))
for line in open("gpurun_out/r02q/bench.log"):
    if line.startswith("{"):
        d = json.loads(line); print(d["value"], d["ms_per_step"], d["stage_seconds_last_step"])
PY
