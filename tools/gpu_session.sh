#!/bin/bash
# One GPU session: usage  bash tools/gpu_session.sh <tag> <what...>   what = tests | c2 | c5 | c4 | c3s | c3 | <any shell command in quotes>
# Every item runs under its own timeout; outputs land in gpurun_out/<tag>/ and a summary is printed at the end.
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=gpurun_out/$TAG; mkdir -p $O
t() { name=$1; lim=$2; shift; shift; echo "=== $name" >> $O/summary.log; s=$(date +%s); ( timeout $lim "$@" ) > $O/$name.log 2> $O/$name.err; echo "rc=$? $(( $(date +%s) - s ))s" >> $O/summary.log; tail -c 1500 $O/$name.log | tail -4 | cut -c1-1200 >> $O/summary.log; tail -3 $O/$name.err | cut -c1-400 >> $O/summary.log; }
for w in "$@"; do
  case "$w" in
    tests) t tests 1200 python -m pytest tests -q -m gpu --durations=8 -p no:cacheprovider ;;
    smoke) t smoke 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ;;
    c2) t bench_c2 500 python bench.py ;;
    c2q) t bench_c2q 300 python bench.py --cpu-frames 0 --no-host-ingest --no-dropin --steps 3 ;;
    c5) t bench_c5 700 python bench.py --config c5 ;;
    c4) t bench_c4 700 python bench.py --config c4 ;;
    c3s) t bench_c3s 400 python bench.py --config c3 --frames 3000 --steps 1 ;;
    c3) t bench_c3 600 python bench.py --config c3 --steps 1 ;;
    *) t custom_$(echo "$w" | tr -c 'a-zA-Z0-9' '_' | cut -c1-40) 900 bash -c "$w" ;;
  esac
done
cat $O/summary.log | cut -c1-1500
