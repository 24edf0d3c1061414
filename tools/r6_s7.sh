#!/bin/bash
# round 6, session 7: sentinel hand-off in the fused GRU backward: bits, parity, stress, A/B against the previous library (tools/_prev)
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_presplit_gpu.py -m gpu -q -x --timeout 600 -k "device_bits and gru" 2>&1 | tail -6
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_fullsize_gpu.py -m gpu -q -x --timeout 600 -k "gru" 2>&1 | tail -4
timeout 1200 python -m pytest tests/test_stress_gpu.py -m gpu -q -x --timeout 600 -k "handoff_stress and gru" 2>&1 | tail -6
{
for i in 1 2; do
  echo "== previous library"; DEP_LIB_PATH=$PWD/tools/_prev/libdep_rnn.so timeout 300 python tools/bench_rnn.py gru 2>&1 | grep -v "Warn\|amdgpu" | tail -3
  echo "== sentinel backward"; timeout 300 python tools/bench_rnn.py gru 2>&1 | grep -v "Warn\|amdgpu" | tail -3
done
} | tee gpurun_out/r6_s7_bwd_sentinel_ab.txt
