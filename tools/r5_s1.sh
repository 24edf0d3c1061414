#!/bin/bash
# round 5, GPU session 1: the all-gather backward sweep (DEP_BWD_AG=1) -- parity, A/B against the reduce-scatter kernel, phase traces
set -u
export TMPDIR=/tmp
out=$PWD/gpurun_out/r5s1; mkdir -p $out
{
echo "== A/B (rnn operator only, STEPS=10)"
for i in 1 2; do
  DEP_BWD_AG=0 STEPS=10 timeout 120 python tools/bench_rnn.py gru
  DEP_BWD_AG=1 STEPS=10 timeout 120 python tools/bench_rnn.py gru
done
echo "== parity AG"
DEP_BWD_AG=1 timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "rnn and gru" -p no:cacheprovider 2>&1 | tail -5
DEP_BWD_AG=1 timeout 600 python -m pytest tests/test_fullsize_gpu.py -q -x -k "gru" -p no:cacheprovider 2>&1 | tail -5
DEP_BWD_AG=1 timeout 600 python -m pytest tests/test_presplit_gpu.py -q -x -k "leave_every or 16bit" -p no:cacheprovider 2>&1 | tail -5
echo "== stress AG"
DEP_BWD_AG=1 timeout 300 python tests/stress_handoff.py --cell gru --iters 10 2>&1 | grep '^{' | tail -1 | cut -c1-400
DEP_BWD_AG=1 DEP_CLUSTER_NOFAST=1 timeout 300 python tests/stress_handoff.py --cell gru --iters 6 2>&1 | grep '^{' | tail -1 | cut -c1-400
echo "== traces"
echo "-- reduce-scatter"; DEP_TRACE=1 DEP_BWD_AG=0 timeout 120 python tools/trace_bwd.py
echo "-- all-gather"; DEP_TRACE=1 DEP_BWD_AG=1 timeout 120 python tools/trace_bwd.py
echo "-- fused forward"; DEP_TRACE=1 timeout 120 python tools/trace_fused.py
echo "== bench step A/B"
for ag in 0 1 0 1; do DEP_BWD_AG=$ag timeout 200 python bench.py --no-cpu-baseline --no-other-workloads --profile-run 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ag=$ag', d['ms_per_step'], d['roofline'].get('kernels_ms_per_step'))"; done
} > $out/log.txt 2>&1
tail -80 $out/log.txt
