cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
tag=r02_final; out=$PWD/gpurun_out/$tag; mkdir -p $out
for wl in audio_gru text_bilstm fusion; do
  bash tools/prof_pmc.sh $tag/pmc_$wl python $PWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline --workload $wl > $out/pmc_$wl.txt 2>&1
done
bash tools/prof_sq.sh $tag/sq python $PWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $out/sq.txt 2>&1
grep "launches=" $out/pmc_audio_gru.txt | head -6; grep "wave_cyc" $out/sq.txt | head -8
