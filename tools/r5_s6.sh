#!/bin/bash
# round 5, GPU session 6: fused all-gather backward v2 (16-bit gates in, PK image out, paired dW GEMMs): parity, bit identity, A/B
set -u
export TMPDIR=/tmp
out=$PWD/gpurun_out/r5s6; mkdir -p $out
{
echo "== parity fused AG backward"
DEP_FUSED2_BWD=1 timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "rnn and gru" -p no:cacheprovider 2>&1 | tail -8
DEP_FUSED2_BWD=1 timeout 600 python -m pytest tests/test_fullsize_gpu.py -q -x -k "gru or audio" -p no:cacheprovider 2>&1 | tail -8
DEP_FUSED2_BWD=1 timeout 900 python -m pytest tests/test_presplit_gpu.py -q -x -k "leave_every or 16bit or paired" -p no:cacheprovider 2>&1 | tail -8
DEP_FUSED2_BWD=1 timeout 900 python -m pytest tests/test_models_gpu.py -q -x -p no:cacheprovider 2>&1 | tail -5
echo "== stress"
DEP_FUSED2_BWD=1 timeout 300 python tests/stress_handoff.py --cell gru --iters 10 2>&1 | grep '^{' | tail -1 | cut -c1-300
DEP_FUSED2_BWD=1 DEP_CLUSTER_NOFAST=1 timeout 300 python tests/stress_handoff.py --cell gru --iters 6 2>&1 | grep '^{' | tail -1 | cut -c1-300
DEP_FUSED2_BWD=1 timeout 300 python tests/stress_handoff.py --cell gru --iters 8 --load --load-phase bwd 2>&1 | grep '^{' | tail -1 | cut -c1-300
echo "== A/B rnn operator"
for i in 1 2; do for fb in 0 1; do echo "fused_bwd=$fb"; DEP_FUSED2_BWD=$fb STEPS=10 timeout 120 python tools/bench_rnn.py gru 2>&1 | grep -v amdgpu.ids; done; done
echo "== bench step"
for fb in 0 1 0 1; do DEP_FUSED2_BWD=$fb timeout 200 python bench.py --no-cpu-baseline --no-other-workloads --profile-run 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fused_bwd=$fb', d['ms_per_step'], d['roofline'].get('kernels_ms_per_step'))"; done
} > $out/log.txt 2>&1
tail -70 $out/log.txt
