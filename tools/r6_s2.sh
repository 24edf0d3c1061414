#!/bin/bash
# round 6, session 2: device-bit record with the round-5 library, the same cases on the pruned library, shader clock during the GEMM forms, GPU suite
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
DEP_LIB_PATH=$PWD/tools/_r05/libdep_rnn.so timeout 900 python tests/golden/make_device_bits.py gpurun_out/device_bits_r05.json 2>&1 | tail -5
timeout 900 python tests/golden/make_device_bits.py gpurun_out/device_bits_pruned.json 2>&1 | tail -5
python - <<'PY'
import json
a=json.load(open('gpurun_out/device_bits_r05.json'))['cases']; b=json.load(open('gpurun_out/device_bits_pruned.json'))['cases']
bad=[(c,k) for c in a for k in a[c] if a[c][k]!=b[c].get(k)]
print('device bits r05 vs pruned: cases', len(a), 'mismatches', bad)
PY
( timeout 300 python tools/gemm_clock.py cfg2; timeout 300 python tools/gemm_clock.py cfg3 ) 2>&1 | grep -v Warn | tee gpurun_out/r6_s2_gemm_clock.txt
cp gpurun_out/device_bits_r05.json tests/golden/device_bits.json
( timeout 1800 python -m pytest tests -m gpu -q -x --timeout 600 --durations=15 2>&1 | tail -40 ) > gpurun_out/r6_s2_all.log
tail -25 gpurun_out/r6_s2_all.log
timeout 600 python bench.py 2>&1 | tail -1 > gpurun_out/r6_s2_bench.log
cut -c1-400 gpurun_out/r6_s2_bench.log
