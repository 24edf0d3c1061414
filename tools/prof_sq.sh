#!/bin/bash
# SQ-level PMC pass (kernel-trace only): where do the waves spend their cycles?  usage: tools/prof_sq.sh <name> <cmd...>
name=$1; shift
export TMPDIR=/tmp
out=$PWD/gpurun_out/$name
mkdir -p $out
( cd /tmp && rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $out -o sq -- "$@" ) > $out/run.log 2>&1
# second pass: how busy are the matrix pipes?  (SQ_VALU_MFMA_BUSY_CYCLES counts cycles with an MFMA in flight per SIMD-quad
# unit; SQ_BUSY_CU_CYCLES the cycles a CU had waves; their ratio / 4 ~ fraction of pipe-cycles used)
( cd /tmp && rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/mfma -o sq -- "$@" ) > $out/run_mfma.log 2>&1
python - "$out" <<'PY'
import csv, glob, sys, collections, re
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(out + '/**/*counter_collection.csv', recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '')
        k = re.split(r'\(', k)[0][:44]
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
        key = (k, r['Dispatch_Id'])
        if key not in seen: seen.add(key); cnt[k] += 1
rows = sorted(acc.items(), key=lambda kv: -kv[1].get('SQ_WAVE_CYCLES', 0))
for k, v in rows[:10]:
    wc = v.get('SQ_WAVE_CYCLES', 1)
    print('%-44s n=%-3d wave_cyc/launch=%10.3e wait_any=%4.1f%% wait_inst=%4.1f%% active=%4.1f%% lds_conf/lds_active=%5.2f' % (
        k, cnt[k], wc / cnt[k], 100 * v.get('SQ_WAIT_ANY', 0) / wc, 100 * v.get('SQ_WAIT_INST_ANY', 0) / wc,
        100 * v.get('SQ_ACTIVE_INST_ANY', 0) / wc, v.get('SQ_LDS_BANK_CONFLICT', 0) / max(1.0, v.get('SQ_LDS_IDX_ACTIVE', 1))))
print('--- matrix pipes (second pass)')
for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get('SQ_VALU_MFMA_BUSY_CYCLES', 0))[:8]:
    if 'SQ_VALU_MFMA_BUSY_CYCLES' not in v: continue
    n = max(1, cnt[k]); gui = v.get('GRBM_GUI_ACTIVE', 0) / n
    print('%-44s n=%-3d mfma_busy_cyc/launch=%10.3e  busy_cu_cyc/launch=%10.3e  gui_active/launch=%10.3e  mfma_busy/(4*busy_cu)=%5.1f%%  bf16_mops=%10.3e f32_mops=%10.3e' % (
        k, n, v['SQ_VALU_MFMA_BUSY_CYCLES'] / n, v.get('SQ_BUSY_CU_CYCLES', 0) / n, gui,
        100 * v['SQ_VALU_MFMA_BUSY_CYCLES'] / max(1.0, 4 * v.get('SQ_BUSY_CU_CYCLES', 1)), v.get('SQ_INSTS_VALU_MFMA_MOPS_BF16', 0) / n, v.get('SQ_INSTS_VALU_MFMA_MOPS_F32', 0) / n))
PY
tail -2 $out/run.log
