#!/bin/bash
# dev-time GPU session (round 2, sixth): scoring pipeline flattened across feature-row steps, wave-slot skew A/B, host profile
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=gpurun_out/r02f; mkdir -p $O
export OMP_WAIT_POLICY=passive
t() { name=$1; lim=$2; shift; shift; echo "=== $name" >> $O/summary.log; s=$(date +%s); ( timeout $lim "$@" ) > $O/$name.log 2>&1; echo "rc=$? $(( $(date +%s) - s ))s" >> $O/summary.log; tail -2 $O/$name.log | cut -c1-1800 >> $O/summary.log; }
t tests_det 600   python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_bench_config.py tests/test_gpu_e2e.py tests/test_golden.py -q -m gpu -x
t bench_base 200  python bench.py --steps 3 --warmup 1 --cpu-frames 0 --no-host-ingest
t bench_s20 200   env PVF_SCORE_SKEW=512,20 python bench.py --steps 3 --warmup 1 --cpu-frames 0 --no-host-ingest
t bench_s40 200   env PVF_SCORE_SKEW=512,40 python bench.py --steps 3 --warmup 1 --cpu-frames 0 --no-host-ingest
t bench_s70 200   env PVF_SCORE_SKEW=512,70 python bench.py --steps 3 --warmup 1 --cpu-frames 0 --no-host-ingest
t bench_pyprof 200 env PVF_PYPROF=$R/$O/pyprof.txt python bench.py --steps 2 --warmup 1 --cpu-frames 0 --no-host-ingest
cat $O/summary.log | grep -v "^$" | cut -c1-400
for f in bench_base bench_s20 bench_s40 bench_s70; do python - $O/$f.log <<'PY'
import json,sys
for line in open(sys.argv[1]):
    if line.startswith("{"):
        d=json.loads(line); print(sys.argv[1], d["value"], d["ms_per_step"], d["kernel_families_ms"], d["roofline"]["achieved"], d["stage_seconds_last_step"])
PY
done
head -60 $O/pyprof.txt
