#!/bin/bash
# round 5, GPU session 13: the fused GRU forward with the sentinel hand-off (DEP_FWD_SX, default 1): parity, fallback / stress, trace, A/B
set -u
export TMPDIR=/tmp
out=$PWD/gpurun_out/${OUT:-r5s13}; mkdir -p $out
{
echo "== parity (default = SX)"
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "rnn and gru" -p no:cacheprovider 2>&1 | tail -6
timeout 600 python -m pytest tests/test_fullsize_gpu.py -q -x -k "gru or audio or fusion" -p no:cacheprovider 2>&1 | tail -6
timeout 900 python -m pytest tests/test_models_gpu.py -q -x -p no:cacheprovider 2>&1 | tail -4
timeout 900 python -m pytest tests/test_presplit_gpu.py -q -x -k "16bit or bf16_storage" -p no:cacheprovider 2>&1 | tail -4
echo "== stress / fallback"
timeout 300 python tests/stress_handoff.py --cell gru --iters 10 2>&1 | grep '^{' | tail -1 | cut -c1-300
DEP_CLUSTER_NOFAST=1 timeout 300 python tests/stress_handoff.py --cell gru --iters 6 2>&1 | grep '^{' | tail -1 | cut -c1-300
DEP_NUM_CUS=200 timeout 300 python tests/stress_handoff.py --cell gru --iters 4 2>&1 | grep '^{' | tail -1 | cut -c1-300
for how in 1 2 3; do DEP_FORCE_SOFT_FALLBACK=$how timeout 300 python tests/stress_handoff.py --cell gru --iters 4 --two-refs 2>&1 | grep '^{' | tail -1 | cut -c1-300; done
timeout 300 python tests/stress_handoff.py --cell gru --iters 8 --load --load-m 1024 --two-refs 2>&1 | grep '^{' | tail -1 | cut -c1-300
timeout 600 python -m pytest tests/test_stress_gpu.py -q -x -k "fall or shared" -p no:cacheprovider 2>&1 | tail -4
echo "== traces"
for sx in 1 0; do echo "-- DEP_FWD_SX=$sx"; DEP_TRACE=1 DEP_FWD_SX=$sx timeout 120 python tools/trace_fused.py 2>&1 | grep -v amdgpu.ids | head -16; done
echo "== A/B rnn operator"
for i in 1 2; do for sx in 0 1; do echo "sx=$sx"; DEP_FWD_SX=$sx STEPS=10 timeout 120 python tools/bench_rnn.py gru 2>&1 | grep -v amdgpu.ids; done; done
echo "== bench step"
for sx in 0 1 0 1; do DEP_FWD_SX=$sx timeout 200 python bench.py --no-cpu-baseline --no-other-workloads --profile-run 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('sx=$sx', d['ms_per_step'], d['roofline'].get('kernels_ms_per_step'))"; done
} > $out/log.txt 2>&1
tail -70 $out/log.txt
