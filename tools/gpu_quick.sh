#!/bin/bash
# quick GPU check: one test module (default: the shot detector), e.g.  bash tools/gpu_quick.sh tests/test_gpu_e2e.py
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 800 python -m pytest ${1:-tests/test_shot.py} -q -m gpu -p no:cacheprovider 2>&1 | tail -12 | cut -c1-400
