#!/bin/bash
set -o pipefail
root=$(pwd); out=$root/gpurun_out/r03z; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $root
for shape in "--zipf-frac 0.2" "--hot-frac 0.02"; do
tag=$(echo $shape | tr -d ' -.')
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_$tag -- python3 bench.py --no-cpu-baseline --no-profile --config3 0 --steps 16 --warmup 8 --min-timed-ms 20 $shape > $out/bench_$tag.json 2> $out/kt.err || { tail -5 $out/kt.err; exit 1; }
f=$(find $out/kt_$tag -name '*kernel_stats.csv' | head -1)
grep -E "^\"Name\"|dj::" $f > $out/kernel_stats_$tag.csv
rm -rf $out/kt_$tag
echo "== $shape"
python3 - $out/kernel_stats_$tag.csv <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    print(r['Name'][:70], r['Calls'], round(float(r['AverageNs'])/1000,1), r['Percentage'])
PY
done
