# A/B of builds of the library on the detector alone: bash tools/probes/ab_screen.sh <variant>[:SEG] ...   (variants = _ab/libpvface_<variant>.so)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for a in "$@"; do v=${a%%:*}; seg=${a#*:}; [ "$seg" = "$a" ] && seg=""; echo "== $a"; PVF_SCREEN_SEG=$seg PVF_LIBRARY=$R/_ab/libpvface_$v.so timeout 120 python tools/bench_detector.py 125 6 2>&1 | tail -1 | cut -c1-330; done
