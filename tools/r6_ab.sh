export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_presplit_gpu.py -m gpu -q -x --timeout 600 -k "device_bits and gru" 2>&1 | tail -2
run() { timeout 300 python bench.py --no-cpu-baseline --no-other-workloads --steps 30 2>&1 | grep "^{" | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['roofline']['kernels_ms_per_step']; print('$1', d['ms_per_step'], k)"; }
for i in 1 2 3; do
  DEP_LIB_PATH=$PWD/tools/_prev/libdep_rnn.so run previous
  run new
done
