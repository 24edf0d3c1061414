cd $GRAFT_REPO_ROOT
for ls in main side-sync side-once; do
python tests/stress_handoff.py --cell gru --iters 3 --load --fwd-only --load-m 1024 --load-stream $ls 2>&1 | tail -1 | cut -c1-400
done
python tests/stress_handoff.py --cell gru --iters 3 --load --fwd-only --load-m 1024 --load-kind torch --load-stream side-once 2>&1 | tail -1 | cut -c1-400
