#!/bin/bash
# dev-time GPU session: all GPU tests + bench with the host timeline
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=gpurun_out/r02r; mkdir -p $O
timeout 900 python -m pytest tests -q -m gpu -x > $O/tests.log 2>&1; echo "tests rc=$?" > $O/summary.log
PVF_TRACE=$R/$O/trace.json timeout 300 python bench.py --steps 3 --warmup 1 --cpu-frames 0 --no-host-ingest > $O/bench.log 2>&1
grep -n "passed\|failed\|error" $O/tests.log | tail -3; cat $O/summary.log
python - <<'PY'
import json
tr = json.load(open("gpurun_out/r02r/trace.json"))
t0 = tr[0][0]
for e in tr: print("%8.2f ms  %s" % ((e[0]-t0)*1e3, " ".join(str(x) for x in e[1:])))
for line in open("gpurun_out/r02r/bench.log"):
    if line.startswith("{"):
        d = json.loads(line); print(d["value"], d["ms_per_step"], d["stage_seconds_last_step"], d["kernel_families_ms"])
PY
