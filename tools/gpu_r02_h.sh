#!/bin/bash
# dev-time GPU session (round 2, eighth): scoring with one-wave workgroups over the valid tiles only
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=gpurun_out/r02h; mkdir -p $O
export OMP_WAIT_POLICY=passive
t() { name=$1; lim=$2; shift; shift; echo "=== $name" >> $O/summary.log; s=$(date +%s); ( timeout $lim "$@" ) > $O/$name.log 2>&1; echo "rc=$? $(( $(date +%s) - s ))s" >> $O/summary.log; tail -2 $O/$name.log | cut -c1-1800 >> $O/summary.log; }
t tests_det 600   python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_bench_config.py tests/test_gpu_e2e.py tests/test_golden.py -q -m gpu -x
t detect 200 python tools/bench_detect.py 32 3
t bench_base 200  python bench.py --steps 3 --warmup 1 --cpu-frames 0 --no-host-ingest
bash tools/pmc_detect.sh r02h/pmc.txt \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" \
  "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES"
cat $O/summary.log | grep -v "^$" | cut -c1-300; tail -1 $O/detect.log; grep score_mfma $O/pmc.txt | cut -c1-300
python - $O/bench_base.log <<'PY'
import json,sys
for line in open(sys.argv[1]):
    if line.startswith("{"):
        d=json.loads(line); print(d["value"], d["ms_per_step"], d["kernel_families_ms"], d["roofline"]["achieved"], d["stage_seconds_last_step"])
PY
