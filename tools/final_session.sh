#!/bin/bash
# the round's last session: everything of tools/measure_round.sh except the FETCH_SIZE / WRITE_SIZE passes (the detector's sources are unchanged: profiles/r05_pmc_kernels.json still applies)
TAG=r05
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=gpurun_out/$TAG; mkdir -p $O
t() { name=$1; lim=$2; shift; shift; echo "=== $name" >> $O/summary.log; s=$(date +%s); ( timeout $lim "$@" ) > $O/$name.log 2>&1; echo "rc=$? $(( $(date +%s) - s ))s" >> $O/summary.log; tail -3 $O/$name.log | cut -c1-600 >> $O/summary.log; }
t tests 900 python -m pytest tests -q -m gpu --durations=5 -p no:cacheprovider
t smoke 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')"
timeout 500 python bench.py > $O/bench.json 2> $O/bench.err; echo "=== bench rc=$?" >> $O/summary.log
timeout 600 python bench.py --config c5 > $O/bench_c5.json 2> $O/bench_c5.err; echo "=== bench c5 rc=$?" >> $O/summary.log
timeout 600 python bench.py --config c3 --steps 1 > $O/bench_c3.json 2> $O/bench_c3.err; echo "=== bench c3 rc=$?" >> $O/summary.log
timeout 700 python bench.py --config c4 > $O/bench_c4.json 2> $O/bench_c4.err; echo "=== bench c4 rc=$?" >> $O/summary.log
t every 300 python bench.py --steps 2 --warmup 1 --cpu-frames 0 --no-host-ingest --no-dropin --detect-every 0.5
t detector 100 python tools/bench_detector.py 125 8
t dsst 100 python tools/bench_dsst.py 2000 4
t embed 100 python tools/bench_embed.py 4096 3
t ert 100 python tools/bench_ert.py 8000 3
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/prof /tmp/pmcc /tmp/pmcd /tmp/e
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof -- python $R/bench.py --cpu-frames 0 --no-host-ingest --no-dropin --no-dense-leg > $R/$O/prof_bench.log 2>&1
DB=$(find /tmp/prof -name "*_results.db" | head -1); python $R/tools/rocprof_top.py $DB > $R/$O/rocprof_kernel_stats.txt 2>&1; python $R/tools/gpu_gaps.py $DB 15 > $R/$O/gpu_gaps.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d /tmp/pmcc -- python $R/bench.py --cpu-frames 0 --no-host-ingest --no-dropin --steps 1 --warmup 0 > /tmp/pc.log 2>&1
DB=$(find /tmp/pmcc -name "*_results.db" | head -1); python $R/tools/pmc_summary.py $DB > $R/$O/pmc_mfma_busy.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_LDS -d /tmp/pmcd -- python $R/bench.py --cpu-frames 0 --no-host-ingest --no-dropin --steps 1 --warmup 0 > /tmp/pd.log 2>&1
DB=$(find /tmp/pmcd -name "*_results.db" | head -1); python $R/tools/pmc_summary.py $DB > $R/$O/pmc_sq_valu.txt 2>&1
timeout 200 rocprofv3 --kernel-trace -d /tmp/e -- python $R/tools/bench_embed.py 4096 2 > /tmp/e.log 2>&1
DB=$(find /tmp/e -name "*_results.db" | head -1); python $R/tools/probes/embed_layers.py $DB > $R/$O/embed_layers.txt 2>&1
cd $R; grep -h "passed\|failed" $O/tests.log; cat $O/summary.log | cut -c1-300; head -c 400 $O/bench.json; echo; head -8 $O/rocprof_kernel_stats.txt; head -3 $O/gpu_gaps.txt
