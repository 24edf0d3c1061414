#!/bin/bash
# round 5, GPU session 2: how the all-gather sweep waits -- DEP_BWD_AGPOLL 0 (vector poll, both members) / 1 (member by member) / 2 (scalar-path poll)
set -u
export TMPDIR=/tmp
out=$PWD/gpurun_out/r5s2; mkdir -p $out
{
echo "== A/B (rnn operator only, STEPS=10), DEP_BWD_AG=1"
for i in 1 2; do for ap in 0 1 2; do echo "agpoll=$ap"; DEP_BWD_AG=1 DEP_BWD_AGPOLL=$ap STEPS=10 timeout 120 python tools/bench_rnn.py gru 2>&1 | grep -v amdgpu.ids; done; done
echo "== parity AG, agpoll=2"
DEP_BWD_AG=1 timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "rnn and gru" -p no:cacheprovider 2>&1 | tail -3
DEP_BWD_AG=1 timeout 600 python -m pytest tests/test_fullsize_gpu.py -q -x -k "gru" -p no:cacheprovider 2>&1 | tail -3
DEP_BWD_AG=1 timeout 300 python tests/stress_handoff.py --cell gru --iters 10 2>&1 | grep '^{' | tail -1 | cut -c1-300
DEP_BWD_AG=1 timeout 300 python tests/stress_handoff.py --cell gru --iters 8 --load --load-phase bwd 2>&1 | grep '^{' | tail -1 | cut -c1-300
echo "== trace agpoll=2"; DEP_TRACE=1 DEP_BWD_AG=1 timeout 120 python tools/trace_bwd.py 2>&1 | grep -v amdgpu.ids
echo "== trace agpoll=0"; DEP_TRACE=1 DEP_BWD_AG=1 DEP_BWD_AGPOLL=0 timeout 120 python tools/trace_bwd.py 2>&1 | grep -v amdgpu.ids
echo "== bench step"
for ap in 0 2 0 2; do DEP_BWD_AG=1 DEP_BWD_AGPOLL=$ap timeout 200 python bench.py --no-cpu-baseline --no-other-workloads --profile-run 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('agpoll=$ap', d['ms_per_step'], d['roofline'].get('kernels_ms_per_step'))"; done
} > $out/log.txt 2>&1
tail -60 $out/log.txt
