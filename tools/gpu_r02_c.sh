#!/bin/bash
# dev-time GPU session (round 2, third): pipelined fused FHOG, ingest ring, cv resize, min-size, CLI, host-ingest bench pass, PMC
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=gpurun_out/r02c; mkdir -p $O
export PVF_VERBOSE=1 OMP_WAIT_POLICY=passive
t() { name=$1; lim=$2; shift; shift; echo "=== $name" >> $O/summary.log; s=$(date +%s); ( timeout $lim "$@" ) > $O/$name.log 2>&1; echo "rc=$? $(( $(date +%s) - s ))s" >> $O/summary.log; tail -3 $O/$name.log | cut -c1-1800 >> $O/summary.log; }
t tests_all 1200  python -m pytest tests -q -m gpu
t bench_w2  200   python bench.py --steps 2 --warmup 1 --cpu-frames 0 --no-host-ingest
t bench_w3  200   env PVF_FHOG_WAVES=3 python bench.py --steps 2 --warmup 1 --cpu-frames 0 --no-host-ingest
t bench_oldfhog 200 env PVF_FHOG=old python bench.py --steps 2 --warmup 1 --cpu-frames 0 --no-host-ingest
t c5 600          python tools/c5_cluster.py $O/c5_cluster.json
t bench_full 500  python bench.py --steps 3 --warmup 1
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof -- python $R/bench.py --steps 2 --warmup 1 --cpu-frames 0 --no-host-ingest > $R/$O/prof_bench.log 2>&1
DB=$(find /tmp/prof -name "*_results.db" | head -1); python $R/tools/rocprof_top.py $DB > $R/$O/kernel_stats.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmca -- python $R/bench.py --cpu-frames 0 --no-host-ingest --steps 1 --warmup 0 > /tmp/pa.log 2>&1
DB=$(find /tmp/pmca -name "*_results.db" | head -1); python $R/tools/pmc_summary.py $DB > $R/$O/pmc_fetch.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pmcb -- python $R/bench.py --cpu-frames 0 --no-host-ingest --steps 1 --warmup 0 > /tmp/pb.log 2>&1
DB=$(find /tmp/pmcb -name "*_results.db" | head -1); python $R/tools/pmc_summary.py $DB > $R/$O/pmc_write.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d /tmp/pmcc -- python $R/bench.py --cpu-frames 0 --no-host-ingest --steps 1 --warmup 0 > /tmp/pc.log 2>&1
DB=$(find /tmp/pmcc -name "*_results.db" | head -1); python $R/tools/pmc_summary.py $DB > $R/$O/pmc_mfma.txt 2>&1
cd $R; cat $O/summary.log
