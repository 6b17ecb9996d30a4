#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 200 python tools/bench_shot.py 1000 2>&1 | tail -3
