# PMC passes over the detector micro-bench (one pass per counter set); usage: bash tools/pmc_detect.sh out.txt "SET1" "SET2" ...
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1; shift
rm -f $OUT
i=0
for set in "$@"; do
  i=$((i+1))
  rm -rf /tmp/pmc$i
  rocprofv3 --kernel-trace --pmc $set -d /tmp/pmc$i -- python $R/tools/bench_detect.py 32 1 > /tmp/pmc$i.log 2>&1
  DB=$(find /tmp/pmc$i -name "*_results.db" | head -1)
  echo "== $set" >> $OUT
  if [ -n "$DB" ]; then python $R/tools/pmc_summary.py $DB | grep -v "at::native\|rocclr" | head -8 >> $OUT; else tail -5 /tmp/pmc$i.log >> $OUT; fi
done
