#!/bin/bash
# dev-time GPU session (round 2): new fused FHOG + systolic scoring, A/B against the old kernels
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r02a; mkdir -p $O
export PVF_VERBOSE=1
t() { name=$1; shift; echo "=== $name" >> $O/summary.log; ( timeout 600 "$@" ) > $O/$name.log 2>&1; echo "rc=$?" >> $O/summary.log; tail -4 $O/$name.log >> $O/summary.log; }
t small_new      python -m pytest tests/test_gpu_parity.py tests/test_golden.py -x -q -m gpu
t small_fusedonly env PVF_SCORE=old python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "detector or features or detect"
t small_sysonly   env PVF_FHOG=old python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "detector or features or detect"
t small_bperm     env PVF_FHOG=bperm PVF_SCORE=old python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "features"
t benchcfg_new   python -m pytest tests/test_gpu_parity_bench_config.py -x -q -m gpu
t bench_new      python bench.py --steps 2 --warmup 1
t bench_oldscore env PVF_SCORE=old python bench.py --steps 2 --warmup 1 --cpu-frames 0
t bench_oldfhog  env PVF_FHOG=old python bench.py --steps 2 --warmup 1 --cpu-frames 0
cd /tmp && export TMPDIR=/tmp
( cd "$OLDPWD" && timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o r02a -- python bench.py --steps 2 --warmup 1 --cpu-frames 0 > $O/prof_bench.log 2>&1 )
cd "$OLDPWD"
f=$(ls $O/prof/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && head -30 "$f" > $O/kernel_stats_head.csv
rm -rf $O/prof/*/*.db 2>/dev/null
cat $O/summary.log
