#!/bin/bash
# bench_detector with every library build under csrc/_exp (experiment variants) and the shipped one; FHOG task heights with the shipped one
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=gpurun_out/fhog_exp; mkdir -p $O; rm -f $O/*.log
for rep in 1 2; do
  for lib in "" $(ls pyannote-video_amd/csrc/_exp/*.so 2>/dev/null); do
    echo "lib=$lib" >> $O/exp.log
    if [ -z "$lib" ]; then timeout 200 python tools/bench_detector.py 125 8 2>&1 | cut -c1-170 >> $O/exp.log
    else PVF_LIBRARY=$R/$lib timeout 200 python tools/bench_detector.py 125 8 2>&1 | cut -c1-170 >> $O/exp.log; fi
  done
  for ch in $FHOG_CHUNKS; do
    echo "lib=chunk$ch" >> $O/exp.log
    PVF_FHOG_CHUNK=$ch timeout 200 python tools/bench_detector.py 125 8 2>&1 | cut -c1-170 >> $O/exp.log
  done
done
cat $O/exp.log
