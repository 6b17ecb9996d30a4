#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=gpurun_out/r02s; mkdir -p $O
timeout 900 python -m pytest tests -q -m gpu -x --durations=12 > $O/tests.log 2>&1
tail -22 $O/tests.log
