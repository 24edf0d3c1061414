cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "rnn" 2>&1 | tail -3
for i in 1 2; do DEP_FUSED2_BWD=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline']['kernels_ms_per_step'], d['eval_forward'])"; done
