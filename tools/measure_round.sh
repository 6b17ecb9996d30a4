mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
python -m pytest tests -q -m gpu -x --no-header -p no:cacheprovider 2>&1 | tail -4 > gpurun_out/t44.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" >> gpurun_out/t44.log 2>&1
python bench.py > gpurun_out/b44.json 2> gpurun_out/b44.err
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof44 /tmp/pmc44a /tmp/pmc44b
rocprofv3 --kernel-trace --stats -d /tmp/prof44 -- python $R/bench.py --cpu-frames 0 > /tmp/p44.log 2>&1
DB=$(find /tmp/prof44 -name "*_results.db" | head -1); python $R/tools/rocprof_top.py $DB > $R/gpurun_out/prof44_stats.txt
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmc44a -- python $R/bench.py --cpu-frames 0 --steps 1 --warmup 0 > /tmp/p44a.log 2>&1
DB=$(find /tmp/pmc44a -name "*_results.db" | head -1); python $R/tools/pmc_summary.py $DB > $R/gpurun_out/pmc44_fetch.txt
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pmc44b -- python $R/bench.py --cpu-frames 0 --steps 1 --warmup 0 > /tmp/p44b.log 2>&1
DB=$(find /tmp/pmc44b -name "*_results.db" | head -1); python $R/tools/pmc_summary.py $DB > $R/gpurun_out/pmc44_write.txt
cd $R; cat gpurun_out/t44.log; head -c 300 gpurun_out/b44.json; echo; head -12 gpurun_out/prof44_stats.txt; head -5 gpurun_out/pmc44_fetch.txt
