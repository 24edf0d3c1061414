#!/bin/bash
# One profiling session of a round (run on the GPU box):  tools/prof_round.sh <tag>
#   1. rocprofv3 --kernel-trace --stats of the default bench command  -> gpurun_out/<tag>/stats/  (+ the bench line)
#   2. separate --pmc passes (FETCH_SIZE, WRITE_SIZE; kernel-trace only) per workload -> HBM bytes per launch
#   3. SQ counters of the headline workload
# then tools/update_pmc_traffic.py turns (2) into profiles/pmc_traffic.json, stamped with the kernel-source digest.
tag=${1:-r02}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
out=$PWD/gpurun_out/$tag
mkdir -p $out
python bench.py --steps 20 --warmup 5 > $out/bench_cfg2.json 2> $out/bench_cfg2.err
python bench.py --steps 20 --warmup 5 --workload text_bilstm > $out/bench_cfg3.json 2> $out/bench_cfg3.err
python bench.py --steps 20 --warmup 5 --workload fusion > $out/bench_cfg4_fusion.json 2> $out/bench_cfg4.err
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o bench -- python $OLDPWD/bench.py --steps 10 --warmup 3 --profile-run ) > $out/stats.log 2>&1
for wl in audio_gru text_bilstm fusion; do
  bash tools/prof_pmc.sh $tag/pmc_$wl python $PWD/bench.py --steps 3 --warmup 1 --profile-run --workload $wl > $out/pmc_$wl.txt 2>&1
done
bash tools/prof_sq.sh $tag/sq python $PWD/bench.py --steps 3 --warmup 1 --profile-run > $out/sq.txt 2>&1
find $out -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
ls $out
