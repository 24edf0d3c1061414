#!/bin/bash
# One GPU-box session: parity tests (+ optional bench / profile).  Usage: tools/gpu_session.sh [tests|bench|prof|all]
# Everything to keep is written under gpurun_out/ (merged back by gpurun).
set -u
mode=${1:-tests}
mkdir -p gpurun_out
export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -m2 -E "gfx|Marketing" > gpurun_out/device.txt
if [[ $mode == tests || $mode == all ]]; then
  timeout 1500 python -m pytest tests -m gpu -q -x --timeout 600 2>&1 | tail -60 > gpurun_out/pytest_gpu.log
  tail -25 gpurun_out/pytest_gpu.log
fi
if [[ $mode == testsall ]]; then
  timeout 1500 python -m pytest tests -m gpu -q --timeout 600 2>&1 | tail -150 > gpurun_out/pytest_gpu.log
  tail -60 gpurun_out/pytest_gpu.log
fi
if [[ $mode == bench || $mode == all ]]; then
  timeout 900 python bench.py 2>&1 | tail -5 | tee gpurun_out/bench.log
fi
