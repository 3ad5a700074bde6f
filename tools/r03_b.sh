#!/bin/bash
# the two forms of the join passes on a table that no cache holds (2^25 slots), batches of 2^20 events
for mode in 8 16; do
  echo "== 2^25 slots, hot-path $mode"
  timeout -k 10 300 python bench.py --no-cpu-baseline --config3 0 --capacity 16777216 --hot-path $mode > gpurun_out/big_$mode.json 2>>gpurun_out/big_err.log || exit 1
  python tests/show_bench.py gpurun_out/big_$mode.json | grep -v bound
done
