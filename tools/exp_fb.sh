cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "rnn" 2>&1 | tail -2
for e in 0 2 4 6; do
echo -n "exp=$e  "; DEP_FB_EXP=$e timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernels_ms_per_step']['gru_bwd_sweep'])"
done
