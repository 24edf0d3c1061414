#!/bin/bash
# round 6, session 6: the PING-PONG 256 x 256-tile GEMM (DEP_GEMM_BIG) against the 256 x 128 kernel: isolated forms with checks, clock, device bits, bench lines
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
{
for b in 0 1; do
  echo "== DEP_GEMM_BIG=$b"
  DEP_GEMM_BIG=$b timeout 300 python tools/bench_gemm.py split --check 2>&1 | grep -v "Warn\|amdgpu.ids"
  DEP_GEMM_BIG=$b timeout 300 python tools/bench_gemm_cfg3.py 2>&1 | grep -v "Warn\|amdgpu.ids"
  DEP_GEMM_BIG=$b timeout 300 python tools/gemm_clock.py cfg3 2>&1 | grep "^cfg3"
  DEP_GEMM_BIG=$b timeout 300 python tools/gemm_clock.py cfg2 2>&1 | grep "^cfg2"
done
} | tee gpurun_out/r6_s6_gemm_pingpong.txt
timeout 900 python -m pytest tests/test_presplit_gpu.py -m gpu -q -x --timeout 600 -k "device_bits or paired or pk_gate" 2>&1 | tail -5 | tee gpurun_out/r6_s6_bits.log
for b in 0 1; do
  for wl in audio_gru text_bilstm; do
    DEP_GEMM_BIG=$b timeout 300 python bench.py --workload $wl --no-cpu-baseline --no-other-workloads 2>&1 | grep "^{" | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('BIG=$b', '$wl', d['ms_per_step'], 'ms', d['roofline']['kernels_ms_per_step'])"
  done
done | tee gpurun_out/r6_s6_bench.txt
