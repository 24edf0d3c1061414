set -u
export TMPDIR=/tmp
tag=r4final; out=$PWD/gpurun_out/$tag; mkdir -p $out
for wl in audio_gru text_bilstm fusion; do
  bash tools/prof_pmc.sh $tag/pmc_$wl python $PWD/bench.py --steps 3 --warmup 1 --profile-run --no-other-workloads --workload $wl > $out/pmc_$wl.txt 2>&1
done
grep -h "launches=" $out/pmc_audio_gru.txt | head -4
python tools/update_pmc_traffic.py $out 4 > $out/pmc_traffic_update.txt 2>&1
cp profiles/pmc_traffic.json $out/pmc_traffic.json
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o bench -- python $OLDPWD/bench.py --steps 10 --warmup 3 --profile-run --no-other-workloads ) > $out/stats.log 2>&1
find $out/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
bash tools/prof_sq.sh $tag/sq python $PWD/bench.py --steps 3 --warmup 1 --profile-run --no-other-workloads > $out/sq.txt 2>&1
python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_cfg2.json 2> $out/bench.err
python bench.py --steps 20 --warmup 5 --no-other-workloads --workload text_bilstm > $out/bench_cfg3.json 2>> $out/bench.err
python bench.py --steps 20 --warmup 5 --no-other-workloads --workload fusion > $out/bench_cfg4_fusion.json 2>> $out/bench.err
for i in 1 2 3; do python bench.py --no-cpu-baseline --no-other-workloads --profile-run 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('repeat', d['ms_per_step'], d['value'])"; done > $out/bench_repeat.txt
DEP_GEMM_MODE=bf16s python tests/stress_handoff.py --cell gru --F 256 --iters 60 2>/dev/null | grep '^{' | tail -1 > $out/soak_bf16s.txt
python - "$out" <<'PY'
import json, sys
for f in ('bench_cfg2', 'bench_cfg3', 'bench_cfg4_fusion'):
    d = json.loads(open(f'{sys.argv[1]}/{f}.json').read().strip().splitlines()[-1])
    print(f, d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline']['step_traffic']['pmc_bytes_per_step'])
PY
cat $out/bench_repeat.txt; cat $out/soak_bf16s.txt | cut -c1-200
