#!/bin/bash
# round 5, GPU sessions 9/10: the direct-fragment BiLSTM forward (DEP_LSTM_DF) and the per-step-stream backward (DEP_LSTM_SE): parity, bit-identity, stress, traces, A/B
set -u
export TMPDIR=/tmp
out=$PWD/gpurun_out/${OUT:-r5s10}; mkdir -p $out
{
echo "== parity (default = DF)"
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "lstm" -p no:cacheprovider 2>&1 | tail -8
timeout 600 python -m pytest tests/test_presplit_gpu.py -q -x -k "bilstm" -p no:cacheprovider 2>&1 | tail -8
timeout 600 python -m pytest tests/test_fullsize_gpu.py -q -x -k "lstm or text or fusion" -p no:cacheprovider 2>&1 | tail -6
echo "== stress"
timeout 300 python tests/stress_handoff.py --cell lstm --iters 12 2>&1 | grep '^{' | tail -1 | cut -c1-300
DEP_CLUSTER_NOFAST=1 timeout 300 python tests/stress_handoff.py --cell lstm --iters 6 --load 2>&1 | grep '^{' | tail -1 | cut -c1-300
DEP_NUM_CUS=200 timeout 300 python tests/stress_handoff.py --cell lstm --iters 4 2>&1 | grep '^{' | tail -1 | cut -c1-300
DEP_LSTM_DF=1 timeout 300 python tests/stress_handoff.py --cell lstm --iters 4 2>&1 | grep '^{' | tail -1 | cut -c1-300
echo "== traces"
for df in 2 1 0; do echo "-- DEP_LSTM_DF=$df"; DEP_TRACE=1 DEP_LSTM_DF=$df timeout 120 python tools/trace_lstm.py 2>&1 | grep -v amdgpu.ids; done
for se in 1 0; do echo "-- backward, DEP_LSTM_SE=$se"; DEP_TRACE=1 DEP_LSTM_SE=$se timeout 120 python tools/trace_lstm.py bwd 2>&1 | grep -v amdgpu.ids; done
echo "== A/B rnn operator"
for i in 1 2; do for se in 0 1; do echo "se=$se df=$((se+1))"; DEP_LSTM_DF=$((se+1)) DEP_LSTM_SE=$se STEPS=10 timeout 120 python tools/bench_rnn.py lstm 2>&1 | grep -v amdgpu.ids; done; done
echo "== bench step cfg3"
for df in 0 1 0 1; do DEP_LSTM_DF=$((df+1)) DEP_LSTM_SE=$df timeout 200 python bench.py --workload text_bilstm --no-cpu-baseline --no-other-workloads --profile-run 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('se=$df df=$((df+1))', d['ms_per_step'], d['roofline'].get('kernels_ms_per_step'))"; done
echo "== bench step fusion"
for df in 0 1; do DEP_LSTM_SE=$df timeout 200 python bench.py --workload fusion --no-cpu-baseline --no-other-workloads 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('se=$df', d['ms_per_step'])"; done
} > $out/log.txt 2>&1
tail -80 $out/log.txt
