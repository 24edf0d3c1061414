cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > gpurun_out/t2.log
python tools/exp_overlap.py > gpurun_out/overlap.log 2>&1
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_a.json 2> gpurun_out/bench_a.err
tail -5 gpurun_out/t2.log; cat gpurun_out/overlap.log; cat gpurun_out/bench_a.json; tail -3 gpurun_out/bench_a.err
