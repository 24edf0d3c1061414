#!/bin/bash
# round 5 final session: PMC traffic (3 workloads), rocprofv3 kernel stats of the driver's command, SQ counters, the three bench lines, repeats, traces, soak
set -u
export TMPDIR=/tmp
tag=r5final; out=$PWD/gpurun_out/$tag; mkdir -p $out
for wl in audio_gru text_bilstm fusion; do
  bash tools/prof_pmc.sh $tag/pmc_$wl python $PWD/bench.py --steps 3 --warmup 1 --profile-run --no-other-workloads --workload $wl > $out/pmc_$wl.txt 2>&1
done
python tools/update_pmc_traffic.py $out 4 > $out/pmc_traffic_update.txt 2>&1
cp profiles/pmc_traffic.json $out/pmc_traffic.json
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o bench -- python $OLDPWD/bench.py --steps 10 --warmup 3 --profile-run --no-other-workloads ) > $out/stats.log 2>&1
find $out/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
bash tools/prof_sq.sh $tag/sq python $PWD/bench.py --steps 3 --warmup 1 --profile-run --no-other-workloads > $out/sq.txt 2>&1
python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_cfg2.json 2> $out/bench.err
python bench.py --steps 20 --warmup 5 --no-other-workloads --workload text_bilstm > $out/bench_cfg3.json 2>> $out/bench.err
python bench.py --steps 20 --warmup 5 --no-other-workloads --workload fusion > $out/bench_cfg4_fusion.json 2>> $out/bench.err
for i in 1 2 3; do python bench.py --no-cpu-baseline --no-other-workloads --profile-run 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('repeat', d['ms_per_step'], d['value'], d['roofline']['kernels_ms_per_step'])"; done > $out/bench_repeat.txt
{ echo "== fused two-layer backward (rnn_fused2_bwd.hip), tools/trace_fbwd.py"; DEP_TRACE=1 python tools/trace_fbwd.py 2>&1 | grep -v amdgpu.ids;
  echo "== per-layer all-gather backward sweep (DEP_FUSED2_BWD=0), tools/trace_bwd.py"; DEP_TRACE=1 DEP_FUSED2_BWD=0 python tools/trace_bwd.py 2>&1 | grep -v amdgpu.ids;
  echo "== per-layer reduce-scatter backward sweep (DEP_FUSED2_BWD=0 DEP_BWD_AG=0)"; DEP_TRACE=1 DEP_FUSED2_BWD=0 DEP_BWD_AG=0 python tools/trace_bwd.py 2>&1 | grep -v amdgpu.ids; } > $out/trace_bwd.txt 2>&1
{ echo "== fused two-layer forward (rnn_fused2.hip gru2_fwd_fused, sentinel hand-off = default), tools/trace_fused.py"; DEP_TRACE=1 python tools/trace_fused.py 2>&1 | grep -v amdgpu.ids;
  echo "== the same with DEP_FWD_SX=0 (acknowledgement wait + flag + poll)"; DEP_TRACE=1 DEP_FWD_SX=0 python tools/trace_fused.py 2>&1 | grep -v amdgpu.ids; } > $out/trace_fwd.txt 2>&1
{ for df in 3 2 1 0; do echo "== BiLSTM-128 forward sweep, layer 0 of cfg3's T and B (rnn_cluster_lstm.hip), DEP_LSTM_DF=$df, tools/trace_lstm.py"; DEP_TRACE=1 DEP_LSTM_DF=$df python tools/trace_lstm.py 2>&1 | grep -v amdgpu.ids; done
  for se in 1 0; do echo "== BiLSTM-128 backward sweep, DEP_LSTM_SE=$se, tools/trace_lstm.py bwd"; DEP_TRACE=1 DEP_LSTM_SE=$se python tools/trace_lstm.py bwd 2>&1 | grep -v amdgpu.ids; done; } > $out/trace_lstm.txt 2>&1
{ echo "== A/B of the round's switches, full train step (bench.py --profile-run), same session";
  for env in "" "DEP_FUSED2_BWD=0" "DEP_FUSED2_BWD=0 DEP_BWD_AG=0" "DEP_DW_PAIR=0" "DEP_FUSED2_BWD=0 DEP_BWD_AG=0 DEP_DW_PAIR=0" "DEP_FWD_DF=1" "DEP_FWD_SX=0" ""; do
    echo "-- ${env:-default}"; env $env python bench.py --no-cpu-baseline --no-other-workloads --profile-run 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['roofline']['kernels_ms_per_step'])"; done;
  echo "== cfg3 (--workload text_bilstm)";
  for env in "" "DEP_LSTM_DF=0 DEP_LSTM_SE=0" "DEP_LSTM_DF=1 DEP_LSTM_SE=0" "DEP_LSTM_DF=2" "DEP_LSTM_DF=0" "DEP_LSTM_SE=0" ""; do
    echo "-- ${env:-default}"; env $env python bench.py --workload text_bilstm --no-cpu-baseline --no-other-workloads --profile-run 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['roofline']['kernels_ms_per_step'])"; done;
  echo "== fusion (--workload fusion)";
  for env in "" "DEP_LSTM_DF=0 DEP_LSTM_SE=0" ""; do
    echo "-- ${env:-default}"; env $env python bench.py --workload fusion --no-cpu-baseline --no-other-workloads 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"; done; } > $out/ab_switches.txt 2>&1
bash tools/soak.sh $out/soak.txt > /dev/null 2>&1
python - "$out" <<'PY'
import json, sys
for f in ('bench_cfg2', 'bench_cfg3', 'bench_cfg4_fusion'):
    d = json.loads(open(f'{sys.argv[1]}/{f}.json').read().strip().splitlines()[-1])
    print(f, d['ms_per_step'], d['value'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline']['step_traffic']['pmc_bytes_per_step'])
PY
cat $out/bench_repeat.txt; cat $out/ab_switches.txt
