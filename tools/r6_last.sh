#!/bin/bash
# round 6, last session: the whole GPU suite, smoke, the driver's bench command
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 2400 python -m pytest tests -m gpu -q --timeout 600 2>&1 | grep -E "passed|failed|error" | tail -3 ) > gpurun_out/r6_last_tests.log; cat gpurun_out/r6_last_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 600 python bench.py --gpus 1 2>&1 | grep "^{" | tail -1 > gpurun_out/r6_last_bench.json
python -c "
import json; d = json.load(open('gpurun_out/r6_last_bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline']['families']['dominant'], {k: v['ms_per_step'] for k, v in d['extra']['other_workloads'].items()})"
