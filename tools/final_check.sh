set -u
export TMPDIR=/tmp
tag=r03c; out=$PWD/gpurun_out/$tag; mkdir -p $out
for wl in audio_gru text_bilstm fusion; do
  bash tools/prof_pmc.sh $tag/pmc_$wl python $PWD/bench.py --steps 3 --warmup 1 --profile-run --workload $wl > $out/pmc_$wl.txt 2>&1
done
grep -h "launches=" $out/pmc_audio_gru.txt | head -4
python -m pytest tests/ -x -q -m gpu 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_cfg2.json 2> $out/bench.err
tail -c 700 $out/bench_cfg2.json
