cd $GRAFT_REPO_ROOT
for rep in 1 2; do for e in 0 32; do
echo -n "exp=$e  "; DEP_FB_EXP=$e timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernels_ms_per_step'])"
done; done
echo -n "per-layer bwd  "; DEP_FUSED2_BWD=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernels_ms_per_step'])"
DEP_FB_EXP=32 timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "rnn" 2>&1 | tail -2
