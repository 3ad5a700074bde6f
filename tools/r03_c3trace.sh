#!/bin/bash
# kernel trace of the default bench (configs[1] + the configs[2] job) on the final code
set -o pipefail
root=$(pwd); out=$root/gpurun_out/r03c3; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $root
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python3 bench.py --no-cpu-baseline --no-profile --steps 8 --warmup 8 --min-timed-ms 0 > $out/bench.json 2> $out/kt.err || { tail -5 $out/kt.err; exit 1; }
f=$(find $out/kt -name '*kernel_stats.csv' | head -1)
grep -E "^\"Name\"|dj::" $f > $out/kernel_stats.csv
rm -rf $out/kt
python3 - <<'PY'
import csv
for r in csv.DictReader(open('gpurun_out/r03c3/kernel_stats.csv')):
    print(r['Name'][:64], r['Calls'], round(float(r['AverageNs'])/1000,1), round(float(r['MinNs'])/1000,1), round(float(r['MaxNs'])/1000,1))
PY
