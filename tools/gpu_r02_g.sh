#!/bin/bash
# dev-time GPU session (round 2, seventh): where do the scoring kernel's non-MFMA cycles go (timing probes + SQ counters)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=gpurun_out/r02g; mkdir -p $O
for v in "0,0" "1,0" "2,0" "3,0"; do echo "== PVF_SCORE_SKEW=$v" >> $O/probe.txt; PVF_SCORE_SKEW=$v timeout 120 python tools/bench_detect.py 32 3 2>&1 | tail -1 >> $O/probe.txt; done
echo "== LDS 90 KB (1 block/CU)" >> $O/probe.txt; PVF_SCORE_LDS_KB=90 timeout 120 python tools/bench_detect.py 32 3 2>&1 | tail -1 >> $O/probe.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u > $R/$O/sq_counters.txt
cd $R
bash tools/pmc_detect.sh r02g/pmc.txt \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
  "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
  "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD" \
  "SQ_IFETCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS"
cat $O/probe.txt; cat $O/pmc.txt | cut -c1-400; wc -l $O/sq_counters.txt
