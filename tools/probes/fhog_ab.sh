#!/bin/bash
# parity of the detector's stages, then the detector alone (tools/bench_detector.py) with the two FHOG kernels and with dense scoring
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=gpurun_out/fhog_ab; mkdir -p $O; rm -f $O/*.log
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_screen.py tests/test_gpu_parity_bench_config.py -q -m gpu -p no:cacheprovider -x 2>&1 | tail -15 | cut -c1-300 ) > $O/parity.log
for rep in 1 2; do
  for v in 0 1; do
    echo "PVF_FHOG_SPLIT=$v" >> $O/ab.log
    PVF_FHOG_SPLIT=$v timeout 200 python tools/bench_detector.py 125 8 2>&1 | cut -c1-200 >> $O/ab.log
  done
done
echo "dense" >> $O/ab.log
PVF_DETECTOR_SCREENING=0 timeout 200 python tools/bench_detector.py 125 4 2>&1 | cut -c1-200 >> $O/ab.log
cat $O/parity.log; cat $O/ab.log
