#!/bin/bash
set -o pipefail
root=$(pwd); out=$root/gpurun_out/r03w; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $root
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- tools/_build/batcher_bench 16 1000 1 0 > $out/bench.txt 2> $out/kt.err || { tail -5 $out/kt.err; exit 1; }
f=$(find $out/kt -name '*kernel_stats.csv' | head -1)
cut -c1-60,100- $f | head -20 | cut -c1-170
cat $out/bench.txt
rm -rf $out/kt
