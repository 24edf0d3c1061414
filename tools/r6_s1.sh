#!/bin/bash
# round 6, session 1: the new model-form parity cases + bench-step oracle + instance coverage, then the whole GPU suite and a bench line
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_fullsize_gpu.py tests/test_kernels_gpu.py tests/test_step_coverage_gpu.py -m gpu -q --timeout 600 -k "full_size_stack or rnn_stack or interlayer or step_coverage" --durations=15 2>&1 | tail -80 ) > gpurun_out/r6_s1_new.log
tail -40 gpurun_out/r6_s1_new.log
( timeout 1500 python -m pytest tests -m gpu -q --timeout 600 --durations=25 2>&1 | tail -120 ) > gpurun_out/r6_s1_all.log
tail -30 gpurun_out/r6_s1_all.log
timeout 600 python bench.py 2>&1 | tail -3 > gpurun_out/r6_s1_bench.log
cut -c1-600 gpurun_out/r6_s1_bench.log
