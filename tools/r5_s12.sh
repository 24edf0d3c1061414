#!/bin/bash
# round 5, GPU session 12: the BiLSTM forward with the sentinel hand-off (DEP_LSTM_DF=3): parity, bit-identity, stress, trace, A/B against DF=2
set -u
export TMPDIR=/tmp
out=$PWD/gpurun_out/${OUT:-r5s12}; mkdir -p $out
{
echo "== parity DEP_LSTM_DF=3"
DEP_LSTM_DF=3 timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "lstm" -p no:cacheprovider 2>&1 | tail -8
DEP_LSTM_DF=3 timeout 600 python -m pytest tests/test_fullsize_gpu.py -q -x -k "lstm or text or fusion" -p no:cacheprovider 2>&1 | tail -6
timeout 600 python -m pytest tests/test_presplit_gpu.py -q -x -k "direct_fragment_bilstm" -p no:cacheprovider 2>&1 | tail -6
echo "== stress DEP_LSTM_DF=3"
DEP_LSTM_DF=3 timeout 300 python tests/stress_handoff.py --cell lstm --iters 12 2>&1 | grep '^{' | tail -1 | cut -c1-300
DEP_LSTM_DF=3 DEP_CLUSTER_NOFAST=1 timeout 300 python tests/stress_handoff.py --cell lstm --iters 6 --load 2>&1 | grep '^{' | tail -1 | cut -c1-300
DEP_LSTM_DF=3 DEP_NUM_CUS=200 timeout 300 python tests/stress_handoff.py --cell lstm --iters 4 2>&1 | grep '^{' | tail -1 | cut -c1-300
DEP_LSTM_DF=3 timeout 300 python tests/stress_handoff.py --cell lstm --iters 6 --load 2>&1 | grep '^{' | tail -1 | cut -c1-300
echo "== traces"
for df in 3 2; do echo "-- DEP_LSTM_DF=$df"; DEP_TRACE=1 DEP_LSTM_DF=$df timeout 120 python tools/trace_lstm.py 2>&1 | grep -v amdgpu.ids; done
echo "== A/B rnn operator"
for i in 1 2; do for df in 2 3; do echo "df=$df"; DEP_LSTM_DF=$df STEPS=10 timeout 120 python tools/bench_rnn.py lstm 2>&1 | grep -v amdgpu.ids; done; done
echo "== bench step cfg3"
for df in 2 3 2 3; do DEP_LSTM_DF=$df timeout 200 python bench.py --workload text_bilstm --no-cpu-baseline --no-other-workloads --profile-run 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('df=$df', d['ms_per_step'], d['roofline'].get('kernels_ms_per_step'))"; done
} > $out/log.txt 2>&1
tail -60 $out/log.txt
