#!/bin/bash
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 2400 python -m pytest tests -m gpu -q --timeout 600 --durations=12 2>&1 | tail -40 ) > gpurun_out/r6_s5_all.log
tail -22 gpurun_out/r6_s5_all.log
timeout 600 python bench.py 2>&1 | grep '^{' | tail -1 > gpurun_out/r6_s5_bench.json
python - <<'PY'
import json
d = json.load(open('gpurun_out/r6_s5_bench.json'))
print(d['ms_per_step'], d['roofline']['families'])
for k, v in d['extra']['other_workloads'].items():
    print(k, v['ms_per_step'], v['families'], v['step_traffic'])
PY
