for v in 0 1; do
  export DEP_BWD_NT=$v
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('nt=$v',d['ms_per_step'],d['roofline']['kernels_ms_per_step'])"
  bash tools/prof_pmc.sh r03nt$v python $PWD/bench.py --steps 3 --warmup 1 --profile-run 2>&1 | grep -E "gru_bwd|gru2_fwd" 
done
