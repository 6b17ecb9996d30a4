#!/bin/bash
# dev-time GPU session: tracker micro-bench + SQ counters of the fused tracker kernels
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=gpurun_out/r02j; mkdir -p $O
timeout 200 python tools/bench_dsst.py 2000 3 > $O/dsst.txt 2>&1
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"; do
  i=$((i+1)); rm -rf /tmp/pmcj$i
  rocprofv3 --kernel-trace --pmc $set -d /tmp/pmcj$i -- python $R/tools/bench_dsst.py 2000 1 > /tmp/pmcj$i.log 2>&1
  DB=$(find /tmp/pmcj$i -name "*_results.db" | head -1)
  echo "== $set" >> $R/$O/pmc.txt
  if [ -n "$DB" ]; then python $R/tools/pmc_summary.py $DB | grep "fused_k\|scale_fft\|fhog1_feat" >> $R/$O/pmc.txt; else tail -3 /tmp/pmcj$i.log >> $R/$O/pmc.txt; fi
done
cd $R; cat $O/dsst.txt | tail -3; cat $O/pmc.txt | cut -c1-420; grep -i "lds" gpurun_out/r02g/sq_counters.txt | tr '\n' ' '
