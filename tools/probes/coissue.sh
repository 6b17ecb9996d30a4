#!/bin/bash
# the co-issue experiment of round 6: tools/probes/coissue_probe.py with the shipped and the thin resize kernel, then
# one counter pass per side (a --pmc pass serialises the dispatches of a process, so the two sides cannot be counted while they overlap)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/coissue; mkdir -p $O; rm -f $O/*
python tools/probes/coissue_probe.py 4 $O/fat.json > /dev/null 2> $O/fat.err
PVF_RESIZE_THIN=1 python tools/probes/coissue_probe.py 4 $O/thin.json > /dev/null 2> $O/thin.err
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/pmc1 /tmp/pmc2
COISSUE_ONLY=equal:pyramid timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU -d /tmp/pmc1 -- python $R/tools/probes/coissue_probe.py 2 > /tmp/p1.log 2>&1
DB=$(find /tmp/pmc1 -name "*_results.db" | head -1); python $R/tools/pmc_summary.py $DB | grep -v "at::native\|rocclr" | head -6 > $O/pmc_pyramid.txt 2>&1
COISSUE_ONLY=equal:embed timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU -d /tmp/pmc2 -- python $R/tools/probes/coissue_probe.py 2 > /tmp/p2.log 2>&1
DB=$(find /tmp/pmc2 -name "*_results.db" | head -1); python $R/tools/pmc_summary.py $DB | grep -v "at::native\|rocclr" | head -12 > $O/pmc_embed.txt 2>&1
cd $R; cat $O/fat.json $O/thin.json $O/pmc_pyramid.txt $O/pmc_embed.txt; tail -n 3 $O/fat.err
