#!/bin/bash
# SQ-level PMC pass (kernel-trace only): where do the waves spend their cycles?  usage: tools/prof_sq.sh <name> <cmd...>
name=$1; shift
export TMPDIR=/tmp
out=$PWD/gpurun_out/$name
mkdir -p $out
( cd /tmp && rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $out -o sq -- "$@" ) > $out/run.log 2>&1
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
PY
tail -2 $out/run.log
