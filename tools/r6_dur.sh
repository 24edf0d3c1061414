export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 --durations=60 2>&1 | tail -75 > gpurun_out/r6_durations.txt; tail -4 gpurun_out/r6_durations.txt
