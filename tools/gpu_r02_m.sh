#!/bin/bash
# dev-time GPU session: SQ counters of the convolution kernels
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=gpurun_out/r02m; mkdir -p $O
timeout 200 python tools/bench_embed.py 1000 3 > $O/embed.txt 2>&1
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_SCA" \
           "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL" \
           "SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_MISC SQ_IFETCH SQ_INSTS_VMEM_WR" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1)); rm -rf /tmp/pmcm$i
  rocprofv3 --kernel-trace --pmc $set -d /tmp/pmcm$i -- python $R/tools/bench_embed.py 1000 1 > /tmp/pmcm$i.log 2>&1
  DB=$(find /tmp/pmcm$i -name "*_results.db" | head -1)
  echo "== $set" >> $R/$O/pmc.txt
  if [ -n "$DB" ]; then python $R/tools/pmc_summary.py $DB | grep "conv_mfma\|maxpool\|prep_input\|head_k" >> $R/$O/pmc.txt; else tail -3 /tmp/pmcm$i.log >> $R/$O/pmc.txt; fi
done
rm -rf /tmp/profm; rocprofv3 --kernel-trace --stats -d /tmp/profm -- python $R/tools/bench_embed.py 1000 1 > /tmp/profm.log 2>&1
DB=$(find /tmp/profm -name "*_results.db" | head -1); python $R/tools/rocprof_top.py $DB > $R/$O/kernel_stats.txt 2>&1
cd $R; tail -2 $O/embed.txt; cat $O/pmc.txt | cut -c1-420; head -8 $O/kernel_stats.txt
