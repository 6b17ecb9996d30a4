#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_bench_config.py -q -m gpu -x -k "pyramid or detector or resize" 2>&1 | grep "passed\|failed"
timeout 100 python tools/bench_detect.py 64 3 2>&1 | tail -1
