#!/bin/bash
# dev-time GPU session (round 2, second): fused FHOG with read-add-write votes, MFMA pair distances, persistent HAC, cpu baseline leg
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r02b; mkdir -p $O
export PVF_VERBOSE=1 OMP_WAIT_POLICY=passive
t() { name=$1; lim=$2; shift; shift; echo "=== $name" >> $O/summary.log; s=$(date +%s); ( timeout $lim "$@" ) > $O/$name.log 2>&1; echo "rc=$? $(( $(date +%s) - s ))s" >> $O/summary.log; tail -3 $O/$name.log | cut -c1-1500 >> $O/summary.log; }
t tests_all 900   python -m pytest tests -x -q -m gpu
t bench_w2  200   python bench.py --steps 2 --warmup 1 --cpu-frames 0
t bench_w3  200   env PVF_FHOG_WAVES=3 python bench.py --steps 2 --warmup 1 --cpu-frames 0
t bench_oldscore 200 env PVF_SCORE=old python bench.py --steps 2 --warmup 1 --cpu-frames 0
t c5 600          python tools/c5_cluster.py $O/c5_cluster.json
t bench_full 420  python bench.py --steps 2 --warmup 1
( cd /tmp; export TMPDIR=/tmp; cd - >/dev/null; timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o r02b -- python bench.py --steps 2 --warmup 1 --cpu-frames 0 > $O/prof_bench.log 2>&1 )
python tools/rocprof_top.py $O/prof/r02b_results.db > $O/kernel_stats.txt 2>&1
rm -f $O/prof/*.db
cat $O/summary.log
