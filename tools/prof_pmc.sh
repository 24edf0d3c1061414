#!/bin/bash
# HBM traffic per kernel launch from PMC counters: two separate rocprofv3 passes (FETCH_SIZE, WRITE_SIZE),
# kernel-trace only, as MI355X_MICROARCH.md prescribes.  usage: tools/prof_pmc.sh <name> <cmd...>
name=$1; shift
export TMPDIR=/tmp
out=$PWD/gpurun_out/$name
mkdir -p $out
for ctr in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/$ctr -o pmc -- "$@" ) > $out/$ctr.log 2>&1
done
python - "$out" <<'PY'
import csv, glob, sys, json, collections
out = sys.argv[1]
res = collections.defaultdict(dict)
for ctr in ('FETCH_SIZE', 'WRITE_SIZE'):
    for f in glob.glob(f'{out}/{ctr}/**/*counter_collection.csv', recursive=True):
        acc = collections.defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(f)):
            if r.get('Counter_Name') != ctr: continue
            k = r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0][:60]
            acc[k][0] += float(r['Counter_Value']); acc[k][1] += 1
        for k, (v, n) in acc.items():
            res[k][ctr] = v / n; res[k]['launches'] = n
rows = sorted(res.items(), key=lambda kv: -(kv[1].get('FETCH_SIZE', 0) + kv[1].get('WRITE_SIZE', 0)))
json.dump(dict(rows), open(f'{out}/pmc_per_launch_kb.json', 'w'), indent=1)
for k, v in rows[:14]:
    print('%-62s launches=%-4d FETCH_KB/launch=%12.1f WRITE_KB/launch=%12.1f' % (k, v.get('launches', 0), v.get('FETCH_SIZE', 0), v.get('WRITE_SIZE', 0)))
PY
