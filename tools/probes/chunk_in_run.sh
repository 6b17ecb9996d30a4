#!/bin/bash
# the FHOG task height (PVF_FHOG_CHUNK) inside the whole step: taller tasks are faster alone but hold their CUs longer before the other stream's kernels get a turn
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for rep in 1 2; do for ch in ${CHUNKS:-160 64 32}; do
  PVF_FHOG_CHUNK=$ch python bench.py --steps 4 --warmup 2 --cpu-frames 0 --no-host-ingest --no-dropin --no-dense-leg --no-other-configs 2>/dev/null > /tmp/l.json
  python - $ch <<'PY'
import json, sys
d = json.load(open("/tmp/l.json")); f = d["kernel_families_ms"]
print("chunk", sys.argv[1], d["value"], d["ms_per_step"], {k: round(v["ms"] / 4, 1) for k, v in f.items() if v["ms"] > 4})
PY
done; done
