#!/bin/bash
# The host side of libpvface.so (model file readers, association / Munkres, text formatter and parser, plan building, the C ABI's argument
# checks) under AddressSanitizer: builds csrc/_build_asan/libpvface_asan.so (`make asan`: host code instrumented, device code as shipped) and
# runs the CPU test suite against it.  No GPU needed.   usage: bash tools/asan_host_tests.sh [pytest arguments]
set -e
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
make -C pyannote-video_amd/csrc -j8 asan > /dev/null
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0 PVF_LIBRARY=$R/pyannote-video_amd/csrc/_build_asan/libpvface_asan.so \
  python -m pytest tests -q -m "not gpu" -p no:cacheprovider "$@"
