timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "lstm" 2>&1 | tail -2
for e in 0 4 0 4; do echo "LSTM_BURST=$e"; DEP_LSTM_BURST=$e STEPS=10 python tools/bench_rnn.py lstm 2>&1 | grep -v amdgpu | tail -1; done
