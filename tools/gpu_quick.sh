#!/bin/bash
# quick GPU check after a tracker change: tracker parity tests, tracker micro-bench, one end-to-end line
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_bench_config.py tests/test_gpu_e2e.py -q -m gpu -x -k "tracker or pipeline" 2>&1 | grep "passed\|failed"
timeout 200 python tools/bench_dsst.py 2000 3 2>&1 | tail -2
timeout 200 python bench.py --steps 3 --warmup 1 --cpu-frames 0 --no-host-ingest 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['kernel_families_ms']['dsst'])"
