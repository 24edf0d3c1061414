#!/bin/bash
# round 6, session 11: the fused forward with its layers one barrier slot apart -- bit anchors, kernel parity, trace, operator + step timing
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r6_s11.log; : > $O
( timeout 900 python -m pytest tests/test_presplit_gpu.py -m gpu -q -x --timeout 600 -k "recorded_device_bits" 2>&1 | tail -5 ) >> $O
( timeout 1200 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x --timeout 600 -k "gru" 2>&1 | tail -5 ) >> $O
( STEPS=20 timeout 300 python tools/bench_rnn.py gru 2>&1 | tail -2 ) >> $O
( DEP_TRACE=1 timeout 300 python tools/trace_fused.py 2>&1 | tail -14 ) >> $O
( timeout 600 python bench.py --gpus 1 2>&1 | grep "^{" | tail -1 > gpurun_out/r6_s11_bench.json; python -c "
import json; d = json.load(open('gpurun_out/r6_s11_bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['kernels_ms_per_step'])" ) >> $O
cat $O
