#!/bin/bash
# round 6, session 20: the whole GPU suite at the DMA-GEMM digest, smoke, the driver's bench command
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 2400 python -m pytest tests -m gpu -q --timeout 900 2>&1 | tail -15 ) > gpurun_out/r6_s20_tests.log; cat gpurun_out/r6_s20_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 600 python bench.py --gpus 1 2>&1 | grep "^{" | tail -1 > gpurun_out/r6_s20_bench.json
python -c "
import json; d = json.load(open('gpurun_out/r6_s20_bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline']['kernels_ms_per_step'], {k: v['ms_per_step'] for k, v in d['extra']['other_workloads'].items()})"
