#!/bin/bash
# dev-time GPU session: embedding net parity + micro-bench + counters + end-to-end
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=gpurun_out/r02n; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_bench_config.py tests/test_gpu_e2e.py tests/test_golden.py -q -m gpu -x > $O/tests.log 2>&1; echo "tests rc=$?" > $O/summary.log
timeout 200 python tools/bench_embed.py 2000 3 > $O/embed.txt 2>&1
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_SCA"; do
  i=$((i+1)); rm -rf /tmp/pmcn$i
  rocprofv3 --kernel-trace --pmc $set -d /tmp/pmcn$i -- python $R/tools/bench_embed.py 2000 1 > /tmp/pmcn$i.log 2>&1
  DB=$(find /tmp/pmcn$i -name "*_results.db" | head -1)
  echo "== $set" >> $R/$O/pmc.txt
  if [ -n "$DB" ]; then python $R/tools/pmc_summary.py $DB | grep "conv_mfma" >> $R/$O/pmc.txt; else tail -3 /tmp/pmcn$i.log >> $R/$O/pmc.txt; fi
done
cd $R
timeout 300 python bench.py --steps 3 --warmup 1 --cpu-frames 0 --no-host-ingest > $O/bench.log 2>&1
tail -2 $O/tests.log; cat $O/summary.log; tail -1 $O/embed.txt; cat $O/pmc.txt | cut -c1-400
python - $O/bench.log <<'PY'
import json,sys
for line in open(sys.argv[1]):
    if line.startswith("{"):
        d=json.loads(line); print(d["value"], d["ms_per_step"], d["kernel_families_ms"], d["roofline"]["achieved"], d["stage_seconds_last_step"])
PY
