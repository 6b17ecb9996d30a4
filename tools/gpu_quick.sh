#!/bin/bash
# quick GPU check after a pipeline / detector change: end-to-end + golden + bench-config parity, then two bench lines with the host timeline
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
mkdir -p gpurun_out/quick
timeout 400 python -m pytest tests/test_gpu_e2e.py tests/test_golden.py tests/test_gpu_parity_bench_config.py tests/test_cli.py tests/test_gpu_sharded.py -q -m gpu -x 2>&1 | tail -2
for i in 1 2; do
PVF_TRACE=$R/gpurun_out/quick/trace.json timeout 200 python bench.py --steps 3 --warmup 1 --cpu-frames 0 --no-host-ingest 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['stage_seconds_last_step'])"
done
python - <<'PY'
import json
tr = json.load(open("gpurun_out/quick/trace.json"))
t0 = tr[0][0]
for e in tr: print("%8.2f ms  %s" % ((e[0]-t0)*1e3, " ".join(str(x) for x in e[1:])))
PY
