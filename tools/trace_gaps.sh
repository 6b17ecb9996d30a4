#!/bin/bash
# rocprofv3 kernel trace of the default bench workload -> per-kernel stats + the idle gaps of the last step: bash tools/trace_gaps.sh <outdir>
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/$1; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/prof_tg
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_tg -- python $R/bench.py --cpu-frames 0 --no-host-ingest --no-dropin --no-dense-leg --steps 3 --warmup 2 > $O/prof_bench.log 2>&1
DB=$(find /tmp/prof_tg -name "*_results.db" | head -1)
python $R/tools/rocprof_top.py $DB > $O/rocprof_kernel_stats.txt 2>&1
python $R/tools/gpu_gaps.py $DB 15 > $O/gpu_gaps.txt 2>&1
head -30 $O/gpu_gaps.txt
