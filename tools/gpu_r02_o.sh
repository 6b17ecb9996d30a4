#!/bin/bash
# dev-time GPU session: timing probes of the conv kernel (wrong results on purpose): bit 0 no re-park, 1 no re-load, 2 no epilogue, 3 no barriers
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=gpurun_out/r02o; mkdir -p $O; rm -f $O/ab.txt
L=pyannote-video_amd/pyannote_video_amd/libpvface.so
cp $L /tmp/lib_keep.so
for v in 0 1 2 3 4 7 8 15 0; do
  cp tools/variants/libpvface_d$v.so $L
  echo "probe bits=$v: $(timeout 120 python tools/bench_embed.py 2000 5 2>&1 | tail -1)" >> $O/ab.txt
done
cp /tmp/lib_keep.so $L
cat $O/ab.txt
