#!/bin/bash
# dev-time GPU session: A/B of conv variants (channel permutation x staging depth), each library swapped in for the micro-bench
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=gpurun_out/r02o; mkdir -p $O
L=pyannote-video_amd/pyannote_video_amd/libpvface.so
cp $L /tmp/lib_keep.so
for v in 11 10 01 00 11 00; do
  cp tools/variants/libpvface_$v.so $L
  for n in 1000 2000; do echo "variant perm/stage3=$v n=$n: $(timeout 120 python tools/bench_embed.py $n 5 2>&1 | tail -1)" >> $O/ab.txt; done
done
cp /tmp/lib_keep.so $L
cat $O/ab.txt
