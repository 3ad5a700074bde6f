#!/bin/bash
set -o pipefail
root=$(pwd); out=$root/gpurun_out/r03r; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $root
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python3 bench.py --no-cpu-baseline --no-profile --steps 8 --warmup 8 --min-timed-ms 0 > $out/bench.json 2> $out/kt.err || { tail -5 $out/kt.err; exit 1; }
f=$(find $out/kt -name '*kernel_stats.csv' | head -1)
grep -E "Name|dj::" $f > $out/kernel_stats.csv
rm -rf $out/kt
cut -c1-200 $out/kernel_stats.csv | head -40
