#!/bin/bash
# round 6, session 18: the LDS-DMA input projection inside the library -- bit anchors, parity, timing
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r6_s18.log; : > $O
( timeout 900 python -m pytest tests/test_presplit_gpu.py -m gpu -q --timeout 600 2>&1 | tail -8 ) >> $O
( timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_fullsize_gpu.py tests/test_models_gpu.py -m gpu -q --timeout 900 2>&1 | tail -8 ) >> $O
for w in 1 0; do
  echo "== DEP_GEMM_NT_DMA=$w" >> $O
  ( DEP_GEMM_NT_DMA=$w timeout 600 python bench.py --gpus 1 --no-cpu-baseline 2>&1 | grep "^{" | tail -1 > gpurun_out/r6_s18_bench_$w.json; python -c "
import json; d = json.load(open('gpurun_out/r6_s18_bench_$w.json')); print(d['value'], d['ms_per_step'], d['roofline']['kernels_ms_per_step'], {k: (v['ms_per_step'], v.get('kernels_ms_per_step')) for k, v in d['extra']['other_workloads'].items()})" ) >> $O 2>&1
done
cat $O
