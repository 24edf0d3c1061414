#!/bin/bash
# Hand-off soak of a round's final kernels (run on the GPU box): tests/stress_handoff.py at full size, every iteration's outputs and
# gradients compared bit for bit with the first one, the status word read after every step.  usage: tools/soak.sh <out file>
out=${1:-gpurun_out/soak.txt}
{
echo "# tests/stress_handoff.py on the final kernels (one MI355X, full-size stacks: B=512, T=300 unless noted; every iteration bit for bit against the"
echo "# first one (--two-refs: or against the fallback path's reference), status word read after every step).  Final kernels (rounds 5-6): fused all-gather backward, paired dW launch, PK gate gradients,"
echo "# 16-bit saved gates (GRU), non-temporal streams, sentinel hand-off of the fused GRU forward and the BiLSTM forward, per-step streams in both BiLSTM sweeps."
run() { note=$1; shift; python tests/stress_handoff.py "$@" 2>/dev/null | grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); t=sorted(d.get('t_iter',[0])); 
print(json.dumps({'cmd': '$note', 'cell': d['cell'], 'iters': d['iters'], 'mismatches': d['mismatches'], 'status_bad': d['status_bad'], 'nan_iters': d['nan_iters'], 'load': d['load'], 'fallbacks': d['fallbacks'], 't_iter_median_s': t[len(t)//2]}))"; }
run "--cell gru --iters 300" --cell gru --iters 300
run "--cell gru --iters 100 --load --two-refs" --cell gru --iters 100 --load --two-refs
run "--cell lstm --iters 200" --cell lstm --iters 200
run "--cell lstm --iters 60 --load" --cell lstm --iters 60 --load
run "--cell gru --H 128 --iters 60" --cell gru --H 128 --iters 60
run "--cell gru --H 512 --T 100 --iters 30" --cell gru --H 512 --T 100 --iters 30
DEP_EXCLUSIVE=0 run "DEP_EXCLUSIVE=0 --cell gru --iters 60 --load" --cell gru --iters 60 --load
DEP_GEMM_MODE=bf16s run "DEP_GEMM_MODE=bf16s --cell gru --F 256 --iters 60 (bf16-storage mode: repeatability only)" --cell gru --F 256 --iters 60
} > $out 2>&1
cat $out
