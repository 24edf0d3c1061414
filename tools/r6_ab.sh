export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_presplit_gpu.py -m gpu -q -x --timeout 600 -k "device_bits and gru" 2>&1 | tail -2
for i in 1 2; do
  echo "== previous library"; DEP_LIB_PATH=$PWD/tools/_prev/libdep_rnn.so timeout 300 python tools/bench_rnn.py gru 2>&1 | grep "^gru"
  echo "== new"; timeout 300 python tools/bench_rnn.py gru 2>&1 | grep "^gru"
done
echo "== bench step (model form): previous / new"
DEP_LIB_PATH=$PWD/tools/_prev/libdep_rnn.so timeout 300 python bench.py --no-cpu-baseline --no-other-workloads 2>&1 | grep "^{" | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernels_ms_per_step'])"
timeout 300 python bench.py --no-cpu-baseline --no-other-workloads 2>&1 | grep "^{" | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernels_ms_per_step'])"
