#!/bin/bash
# round 6, session 21: 256 x 128-tile form of the LDS-DMA weight-gradient contraction (BiLSTM dW_hh pairs) -- anchors, parity, cfg3 timing A/B
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r6_s21.log; : > $O
( timeout 900 python -m pytest tests/test_presplit_gpu.py -m gpu -q --timeout 600 2>&1 | tail -4 ) >> $O
( timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_fullsize_gpu.py -m gpu -q --timeout 900 -k "lstm or dma or gemm" 2>&1 | tail -4 ) >> $O
for w in 1 0; do
  echo "== DEP_GEMM_TN_DMA=$w (cfg3)" >> $O
  ( DEP_GEMM_TN_DMA=$w timeout 300 python bench.py --workload text_bilstm --no-cpu-baseline --no-other-workloads --profile-run 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['roofline']['kernels_ms_per_step'])" ) >> $O 2>&1
done
cat $O
