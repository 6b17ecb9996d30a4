R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/t3; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/p1 /tmp/p2
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d /tmp/p1 -- python $R/bench.py --cpu-frames 0 --no-host-ingest --no-dropin --steps 1 --warmup 0 > /tmp/p1.log 2>&1
DB=$(find /tmp/p1 -name "*_results.db" | head -1); python $R/tools/pmc_summary.py $DB | grep "score_\|fhog_fused" > $O/pmc1.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_LDS -d /tmp/p2 -- python $R/bench.py --cpu-frames 0 --no-host-ingest --no-dropin --steps 1 --warmup 0 > /tmp/p2.log 2>&1
DB=$(find /tmp/p2 -name "*_results.db" | head -1); python $R/tools/pmc_summary.py $DB | grep "score_\|fhog_fused" > $O/pmc2.txt 2>&1
cat $O/pmc1.txt $O/pmc2.txt
