#!/bin/bash
# dev-time GPU session (round 2, fifth): scoring beside pyramid/FHOG on two streams, conv epilogue
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=gpurun_out/r02e; mkdir -p $O
export OMP_WAIT_POLICY=passive
t() { name=$1; lim=$2; shift; shift; echo "=== $name" >> $O/summary.log; s=$(date +%s); ( timeout $lim "$@" ) > $O/$name.log 2>&1; echo "rc=$? $(( $(date +%s) - s ))s" >> $O/summary.log; tail -2 $O/$name.log | cut -c1-1500 >> $O/summary.log; }
t tests_det 600   python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_bench_config.py tests/test_gpu_e2e.py tests/test_golden.py -q -m gpu -x
t bench_100 200   python bench.py --steps 3 --warmup 1 --cpu-frames 0 --no-host-ingest
t bench_58  200   env PVF_SCORE_LDS_KB=58 python bench.py --steps 3 --warmup 1 --cpu-frames 0 --no-host-ingest
t bench_82  200   env PVF_SCORE_LDS_KB=82 python bench.py --steps 3 --warmup 1 --cpu-frames 0 --no-host-ingest
t bench_120 200   env PVF_SCORE_LDS_KB=120 python bench.py --steps 3 --warmup 1 --cpu-frames 0 --no-host-ingest
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof -- python $R/bench.py --steps 2 --warmup 1 --cpu-frames 0 --no-host-ingest > $R/$O/prof_bench.log 2>&1
DB=$(find /tmp/prof -name "*_results.db" | head -1); python $R/tools/rocprof_top.py $DB > $R/$O/kernel_stats.txt 2>&1
cd $R; cat $O/summary.log; head -10 $O/kernel_stats.txt
