#!/bin/bash
# round 6, final session at the LDS-DMA GEMM digest: PMC traffic (3 workloads), rocprofv3 kernel stats of the driver's command, SQ counters, the three bench lines, repeats, traces, switch A/Bs, soak
set -u
export TMPDIR=/tmp
tag=r6final2; out=$PWD/gpurun_out/$tag; mkdir -p $out
for wl in audio_gru text_bilstm fusion; do
  timeout 400 bash tools/prof_pmc.sh $tag/pmc_$wl python $PWD/bench.py --steps 3 --warmup 1 --profile-run --no-other-workloads --workload $wl > $out/pmc_$wl.txt 2>&1
done
python tools/update_pmc_traffic.py $out 4 > $out/pmc_traffic_update.txt 2>&1
cp profiles/pmc_traffic.json $out/pmc_traffic.json
for wl in audio_gru text_bilstm; do
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_$wl -o bench -- python $OLDPWD/bench.py --steps 10 --warmup 3 --profile-run --no-other-workloads --workload $wl ) > $out/stats_$wl.log 2>&1
  find $out/stats_$wl -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats_$wl.csv
done
timeout 400 bash tools/prof_sq.sh $tag/sq python $PWD/bench.py --steps 3 --warmup 1 --profile-run --no-other-workloads > $out/sq.txt 2>&1
timeout 400 bash tools/prof_sq.sh $tag/sq3 python $PWD/bench.py --steps 3 --warmup 1 --profile-run --no-other-workloads --workload text_bilstm > $out/sq_cfg3.txt 2>&1
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_cfg2.json 2> $out/bench.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-other-workloads --workload text_bilstm > $out/bench_cfg3.json 2>> $out/bench.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-other-workloads --workload fusion > $out/bench_cfg4_fusion.json 2>> $out/bench.err
for i in 1 2 3; do timeout 200 python bench.py --no-cpu-baseline --no-other-workloads --profile-run 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('repeat', d['ms_per_step'], d['value'], d['roofline']['kernels_ms_per_step'])"; done > $out/bench_repeat.txt
{ echo "== fused two-layer backward (rnn_fused2_bwd.hip), tools/trace_fbwd.py"; DEP_TRACE=1 timeout 200 python tools/trace_fbwd.py 2>&1 | grep -v amdgpu.ids;
  echo "== per-layer all-gather backward sweep (DEP_FUSED2_BWD=0), tools/trace_bwd.py"; DEP_TRACE=1 DEP_FUSED2_BWD=0 timeout 200 python tools/trace_bwd.py 2>&1 | grep -v amdgpu.ids; } > $out/trace_bwd.txt 2>&1
{ echo "== fused two-layer forward (rnn_fused2.hip gru2_fwd_fused, sentinel hand-off), tools/trace_fused.py"; DEP_TRACE=1 timeout 200 python tools/trace_fused.py 2>&1 | grep -v amdgpu.ids; } > $out/trace_fwd.txt 2>&1
{ echo "== BiLSTM-128 forward sweep, layer 0 of cfg3's T and B (rnn_cluster_lstm.hip), tools/trace_lstm.py"; DEP_TRACE=1 timeout 200 python tools/trace_lstm.py 2>&1 | grep -v amdgpu.ids;
  echo "== BiLSTM-128 backward sweep, tools/trace_lstm.py bwd"; DEP_TRACE=1 timeout 200 python tools/trace_lstm.py bwd 2>&1 | grep -v amdgpu.ids; } > $out/trace_lstm.txt 2>&1
{ echo "== A/B of the switches that remain, full train step (bench.py --profile-run), same session";
  for env in "" "DEP_GEMM_TN_DMA=0" "DEP_GEMM_NT_DMA=0" "DEP_FUSED2_BWD=0" "DEP_DW_PAIR=0" "DEP_DGI_PK=0" "DEP_SV16=0" "DEP_FUSED2=0" "DEP_EXCLUSIVE=0" ""; do
    echo "-- ${env:-default}"; env $env timeout 200 python bench.py --no-cpu-baseline --no-other-workloads --profile-run 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['roofline']['kernels_ms_per_step'])"; done;
  echo "== cfg3 (--workload text_bilstm)";
  for env in "" "DEP_GEMM_TN_DMA=0" "DEP_GEMM_NT_DMA=0" "DEP_LSTM_SV16=1" "DEP_DGI_PK=0" ""; do
    echo "-- ${env:-default}"; env $env timeout 200 python bench.py --workload text_bilstm --no-cpu-baseline --no-other-workloads --profile-run 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['roofline']['kernels_ms_per_step'])"; done; } > $out/ab_switches.txt 2>&1
timeout 900 bash tools/soak.sh $out/soak.txt > /dev/null 2>&1
python - "$out" <<'PY'
import json, sys
for f in ('bench_cfg2', 'bench_cfg3', 'bench_cfg4_fusion'):
    d = json.loads(open(f'{sys.argv[1]}/{f}.json').read().strip().splitlines()[-1])
    print(f, d['ms_per_step'], d['value'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline']['step_traffic'])
PY
cat $out/bench_repeat.txt; cat $out/ab_switches.txt
