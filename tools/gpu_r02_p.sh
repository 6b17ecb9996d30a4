#!/bin/bash
# dev-time GPU session: parity tests + embedding micro-bench + end-to-end bench
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=gpurun_out/r02p; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_bench_config.py tests/test_gpu_e2e.py tests/test_golden.py -q -m gpu -x > $O/tests.log 2>&1; echo "tests rc=$?" > $O/summary.log
timeout 200 python tools/bench_embed.py 2000 5 > $O/embed.txt 2>&1
timeout 300 python bench.py --steps 3 --warmup 1 --cpu-frames 0 --no-host-ingest > $O/bench.log 2>&1
tail -2 $O/tests.log; cat $O/summary.log; tail -1 $O/embed.txt
python - $O/bench.log <<'PY'
import json,sys
for line in open(sys.argv[1]):
    if line.startswith("{"):
        d=json.loads(line); print(d["value"], d["ms_per_step"], d["kernel_families_ms"], d["roofline"]["achieved"], d["stage_seconds_last_step"])
PY
