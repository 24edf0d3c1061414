#!/bin/bash
# round 6, session 15: the LDS-DMA weight-gradient contraction inside the library -- bit anchors, parity, timing
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r6_s15.log; : > $O
( timeout 900 python -m pytest tests/test_presplit_gpu.py -m gpu -q -x --timeout 600 2>&1 | tail -5 ) >> $O
( timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_fullsize_gpu.py -m gpu -q -x --timeout 900 2>&1 | tail -5 ) >> $O
for w in 1 0; do
  echo "== DEP_GEMM_TN_DMA=$w" >> $O
  ( DEP_GEMM_TN_DMA=$w timeout 600 python bench.py --gpus 1 2>&1 | grep "^{" | tail -1 > gpurun_out/r6_s15_bench_$w.json; python -c "
import json; d = json.load(open('gpurun_out/r6_s15_bench_$w.json')); print(d['value'], d['ms_per_step'], d['roofline']['kernels_ms_per_step'], {k: (v['ms_per_step'], v.get('kernels_ms_per_step')) for k, v in d['extra']['other_workloads'].items()})" ) >> $O 2>&1
done
cat $O
