cd $GRAFT_REPO_ROOT
for rep in 1 2; do for e in 2 6; do
echo -n "exp=$e  "; DEP_FUSED2_BWD=0 DEP_FF_EXP=$e timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernels_ms_per_step']['gru_fwd_sweep'], d['eval_forward']['ms_per_batch'])"
done; done
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "rnn" 2>&1 | tail -2
DEP_FF_EXP=6 timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "rnn" 2>&1 | tail -2
