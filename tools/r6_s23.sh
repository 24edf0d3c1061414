#!/bin/bash
# round 6, session 23: hand-placed fragment reads (ds_read2st64_b32) in the LDS-DMA weight-gradient kernel -- anchors, parity, timing
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r6_s23.log; : > $O
( timeout 900 python -m pytest tests/test_presplit_gpu.py -m gpu -q --timeout 600 2>&1 | tail -4 ) >> $O
( timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_fullsize_gpu.py -m gpu -q --timeout 900 2>&1 | tail -4 ) >> $O
for a in "1024 1024 153600 16 0" "768 256 153600 42 1"; do timeout 120 ./tools/micro/gemm_tn_dma $a 2>&1 | grep "ping-pong, 4\|check"; done >> $O
for i in 1 2; do
  ( timeout 300 python bench.py --no-cpu-baseline --no-other-workloads --profile-run 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['roofline']['kernels_ms_per_step'])" ) >> $O 2>&1
  ( timeout 300 python bench.py --workload text_bilstm --no-cpu-baseline --no-other-workloads --profile-run 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['roofline']['kernels_ms_per_step'])" ) >> $O 2>&1
done
cat $O
