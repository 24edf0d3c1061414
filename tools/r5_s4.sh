#!/bin/bash
# round 5, GPU session 4: the paired dW launch (DEP_DW_PAIR) -- bit identity, A/B of the train step
set -u
export TMPDIR=/tmp
out=$PWD/gpurun_out/r5s4; mkdir -p $out
{
echo "== bit identity"
timeout 900 python -m pytest tests/test_presplit_gpu.py -q -x -k "paired" -p no:cacheprovider 2>&1 | tail -5
echo "== bench step A/B (DEP_DW_PAIR 0 / 1)"
for pr in 0 1 0 1 0 1; do DEP_DW_PAIR=$pr timeout 200 python bench.py --no-cpu-baseline --no-other-workloads --profile-run 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pair=$pr', d['ms_per_step'], d['roofline'].get('kernels_ms_per_step'))"; done
} > $out/log.txt 2>&1
tail -40 $out/log.txt
