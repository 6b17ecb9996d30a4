#!/bin/bash
# dev-time GPU session: tracker tests + micro-bench + end-to-end bench after a tracker kernel change
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=gpurun_out/r02k; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "tracker" > $O/tests_trk.log 2>&1; echo "tests_trk rc=$?" >> $O/summary.log
bash tools/gpu_r02_j.sh > $O/j.log 2>&1
timeout 300 python bench.py --steps 3 --warmup 1 --cpu-frames 0 --no-host-ingest > $O/bench.log 2>&1
tail -2 $O/tests_trk.log; cat $O/summary.log; tail -4 gpurun_out/r02j/dsst.txt; cat gpurun_out/r02j/pmc.txt | cut -c1-300
python - $O/bench.log <<'PY'
import json,sys
for line in open(sys.argv[1]):
    if line.startswith("{"):
        d=json.loads(line); print(d["value"], d["ms_per_step"], d["kernel_families_ms"], d["roofline"]["achieved"], d["stage_seconds_last_step"])
PY
