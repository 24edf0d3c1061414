#!/bin/bash
# rocprofv3 kernel-trace summary of a command; prints the top kernels and saves the CSV under gpurun_out/<name>/
# usage: tools/prof_stats.sh <name> <cmd...>
name=$1; shift
export TMPDIR=/tmp
out=$PWD/gpurun_out/$name
mkdir -p $out
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out -o $name -- "$@" ) > $out/cmd.log 2>&1
python - "$out" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True):
    rows = list(csv.DictReader(open(f)))
    print(f)
    for r in rows[:34]:
        print('%-78s calls=%-5s avg_us=%10.1f total_ms=%9.3f  %5s%%' % (r['Name'][:78], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6, r['Percentage']))
PY
