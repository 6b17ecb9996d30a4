#!/bin/bash
# dev-time GPU session (round 2, fourth): new conv kernel, libpvface_dist.so (world 1), cleaned detect.hip
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=gpurun_out/r02d; mkdir -p $O
export OMP_WAIT_POLICY=passive
t() { name=$1; lim=$2; shift; shift; echo "=== $name" >> $O/summary.log; s=$(date +%s); ( timeout $lim "$@" ) > $O/$name.log 2>&1; echo "rc=$? $(( $(date +%s) - s ))s" >> $O/summary.log; tail -3 $O/$name.log | cut -c1-1800 >> $O/summary.log; }
t tests_all 1200  python -m pytest tests -q -m gpu
t smoke 300       python -c "import __graft_entry__ as g; g.smoke()"
t bench 200       python bench.py --steps 3 --warmup 1 --cpu-frames 0 --no-host-ingest
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof -- python $R/bench.py --steps 2 --warmup 1 --cpu-frames 0 --no-host-ingest > $R/$O/prof_bench.log 2>&1
DB=$(find /tmp/prof -name "*_results.db" | head -1); python $R/tools/rocprof_top.py $DB > $R/$O/kernel_stats.txt 2>&1
cd $R; cat $O/summary.log; head -12 $O/kernel_stats.txt
