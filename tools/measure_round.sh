#!/bin/bash
# One GPU session that produces everything profiles/ holds for a round: usage  bash tools/measure_round.sh r03 [quick]
#   tests (-m gpu) + smoke; a FETCH_SIZE and a WRITE_SIZE pass over the bench workload, turned into profiles/<tag>_pmc_kernels.json (bench.py
#   attaches it as roofline.traffic when the hash of the detector's sources matches); the default bench line (cpu_baseline / parity / host_ingest / dropin_cli);
#   the other BASELINE.json configurations (c3 streamed long video, c4 clip farm, c5 4K crowd); a rocprofv3 kernel-trace summary of the default
#   workload; WRITE_SIZE, matrix-pipe and SQ counter passes (every --pmc pass on its own, with --kernel-trace only); the C5 clustering stressor.
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=gpurun_out/$TAG; mkdir -p $O
t() { name=$1; lim=$2; shift; shift; echo "=== $name" >> $O/summary.log; s=$(date +%s); ( timeout $lim "$@" ) > $O/$name.log 2>&1; echo "rc=$? $(( $(date +%s) - s ))s" >> $O/summary.log; tail -3 $O/$name.log | cut -c1-600 >> $O/summary.log; }
t tests 900 python -m pytest tests -q -m gpu --durations=5 -p no:cacheprovider
t smoke 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')"
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/prof /tmp/pmca /tmp/pmcb /tmp/pmcc /tmp/pmcd /tmp/pmce
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmca -- python $R/bench.py --cpu-frames 0 --no-host-ingest --no-dropin --steps 1 --warmup 0 > /tmp/pa.log 2>&1
DB=$(find /tmp/pmca -name "*_results.db" | head -1); python $R/tools/pmc_summary.py $DB > $R/$O/pmc_fetch_size.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pmcb -- python $R/bench.py --cpu-frames 0 --no-host-ingest --no-dropin --steps 1 --warmup 0 > /tmp/pb.log 2>&1
DB=$(find /tmp/pmcb -name "*_results.db" | head -1); python $R/tools/pmc_summary.py $DB > $R/$O/pmc_write_size.txt 2>&1
cd $R
python tools/pmc_kernels_json.py $O/pmc_fetch_size.txt $O/pmc_write_size.txt $TAG
timeout 500 python bench.py > $O/bench.json 2> $O/bench.err; echo "=== bench rc=$?" >> $O/summary.log
if [ "$2" != "quick" ]; then
  timeout 600 python bench.py --config c5 > $O/bench_c5.json 2> $O/bench_c5.err; echo "=== bench c5 rc=$?" >> $O/summary.log
  timeout 600 python bench.py --config c3 --steps 1 > $O/bench_c3.json 2> $O/bench_c3.err; echo "=== bench c3 rc=$?" >> $O/summary.log
  timeout 700 python bench.py --config c4 > $O/bench_c4.json 2> $O/bench_c4.err; echo "=== bench c4 rc=$?" >> $O/summary.log
  t every 300 python bench.py --steps 2 --warmup 1 --cpu-frames 0 --no-host-ingest --no-dropin --detect-every 0.5
fi
t c5 400 python tools/c5_cluster.py $O/c5_cluster.json
t detector 100 python tools/bench_detector.py 125 8
t dsst 100 python tools/bench_dsst.py 2000 4
t embed 100 python tools/bench_embed.py 4096 3
t ert 100 python tools/bench_ert.py 8000 3
t overlap 200 python tools/probes/overlap_probe.py 3 $O/overlap_probe.json
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof -- python $R/bench.py --cpu-frames 0 --no-host-ingest --no-dropin --no-dense-leg > $R/$O/prof_bench.log 2>&1
DB=$(find /tmp/prof -name "*_results.db" | head -1); python $R/tools/rocprof_top.py $DB > $R/$O/rocprof_kernel_stats.txt 2>&1; python $R/tools/gpu_gaps.py $DB 15 > $R/$O/gpu_gaps.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d /tmp/pmcc -- python $R/bench.py --cpu-frames 0 --no-host-ingest --no-dropin --steps 1 --warmup 0 > /tmp/pc.log 2>&1
DB=$(find /tmp/pmcc -name "*_results.db" | head -1); python $R/tools/pmc_summary.py $DB > $R/$O/pmc_mfma_busy.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_LDS -d /tmp/pmcd -- python $R/bench.py --cpu-frames 0 --no-host-ingest --no-dropin --steps 1 --warmup 0 > /tmp/pd.log 2>&1
DB=$(find /tmp/pmcd -name "*_results.db" | head -1); python $R/tools/pmc_summary.py $DB > $R/$O/pmc_sq_valu.txt 2>&1
# the tracker's fused kernels and the clustering kernels, each in the micro-bench that isolates them
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CU_CYCLES -d /tmp/pmce -- python $R/tools/bench_dsst.py 2000 2 > /tmp/pe.log 2>&1
DB=$(find /tmp/pmce -name "*_results.db" | head -1); python $R/tools/pmc_summary.py $DB | grep -v "at::native\|rocclr" | head -14 > $R/$O/pmc_tracker_sq.txt 2>&1
rm -rf /tmp/pmcf; timeout 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d /tmp/pmcf -- python $R/tools/c5_cluster.py > /tmp/pf.log 2>&1
DB=$(find /tmp/pmcf -name "*_results.db" | head -1); python $R/tools/pmc_summary.py $DB | grep -v "at::native\|rocclr" | head -8 > $R/$O/pmc_cluster_mfma.txt 2>&1
cd $R; grep -h "passed\|failed" $O/tests.log; cat $O/summary.log | cut -c1-400; head -c 700 $O/bench.json; echo; head -14 $O/rocprof_kernel_stats.txt; head -4 $O/pmc_fetch_size.txt; head -3 $O/pmc_mfma_busy.txt
