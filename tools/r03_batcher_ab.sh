#!/bin/bash
# A/B of the per-request front: previous library (tools/_build/prev) against the current one, interleaved
set -o pipefail
out=gpurun_out/r03batcher; mkdir -p $out
: > $out/ab.txt
for rep in 1 2 3; do
for cfg in "16 2000 1 0" "64 1000 1 0" "256 400 1 0"; do
  echo -n "prev: " >> $out/ab.txt
  LD_LIBRARY_PATH=$(pwd)/tools/_build/prev:$LD_LIBRARY_PATH timeout -k 10 120 tools/_build/batcher_bench $cfg >> $out/ab.txt 2>&1 || { tail -5 $out/ab.txt; exit 1; }
  echo -n "new:  " >> $out/ab.txt
  timeout -k 10 120 tools/_build/batcher_bench $cfg >> $out/ab.txt 2>&1 || { tail -5 $out/ab.txt; exit 1; }
done
done
cut -c1-110 $out/ab.txt
