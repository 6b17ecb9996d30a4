cd ${GRAFT_REPO_ROOT:-/root/repo}
for k in "hundreds" "fhog_bit_exact or hundreds" "level_features or hundreds" "detector_raw or hundreds" "detect_batch or hundreds" "detect_many or hundreds" "pyramid_levels or hundreds" "chips or landmarks or embed or hundreds" "tracker or dsst or hundreds" "pair_mean or hac or hundreds"; do
  echo "=== $k"; timeout 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -p no:cacheprovider -k "$k" 2>&1 | tail -3
done
