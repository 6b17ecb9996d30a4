#!/bin/bash
# dev-time GPU session (round 2, ninth): MFMA issue probe; correlation tracker with on-chip plane spectra + copy-on-write clones
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=gpurun_out/r02i; mkdir -p $O
export OMP_WAIT_POLICY=passive
t() { name=$1; lim=$2; shift; shift; echo "=== $name" >> $O/summary.log; s=$(date +%s); ( timeout $lim "$@" ) > $O/$name.log 2>&1; echo "rc=$? $(( $(date +%s) - s ))s" >> $O/summary.log; tail -3 $O/$name.log | cut -c1-1800 >> $O/summary.log; }
true
true
t tests_trk 600   python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "tracker"
t tests_all 900   python -m pytest tests -q -m gpu -x
t bench_base 200  python bench.py --steps 3 --warmup 1 --cpu-frames 0 --no-host-ingest
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof -- python $R/bench.py --steps 2 --warmup 1 --cpu-frames 0 --no-host-ingest > $R/$O/prof_bench.log 2>&1
DB=$(find /tmp/prof -name "*_results.db" | head -1); python $R/tools/rocprof_top.py $DB > $R/$O/kernel_stats.txt 2>&1
cd $R; cat $O/summary.log | grep -v "^$" | cut -c1-400; head -30 $O/kernel_stats.txt
python - $O/bench_base.log <<'PY'
import json,sys
for line in open(sys.argv[1]):
    if line.startswith("{"):
        d=json.loads(line); print(d["value"], d["ms_per_step"], d["kernel_families_ms"], d["roofline"]["achieved"], d["stage_seconds_last_step"])
PY
