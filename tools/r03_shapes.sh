#!/bin/bash
# other workload shapes on the final build: record shapes of configs #4 / #5, ragged messages, max-wait joins, skew
set -o pipefail
out=gpurun_out/r03shapes; mkdir -p $out
: > $out/other_shapes.txt
run() {
  echo "== $*" >> $out/other_shapes.txt
  timeout -k 10 300 python bench.py --no-cpu-baseline --config3 0 "$@" > $out/one.json 2> $out/one.err || { tail -20 $out/one.err; exit 1; }
  python tests/show_bench.py $out/one.json | grep -v "^  {" >> $out/other_shapes.txt
}
run --count 16 --payload 256 --barriers 524288
run --count 16 --payload 1024 --barriers 262144
run --payload 48
run --payload 100
run --payload 128
run --join-max-wait 30
run --hot-frac 0.02
run --zipf-frac 0.2
run --capacity 16777216
run --barriers 8388608 --steps 8 --warmup 8
cat $out/other_shapes.txt | grep -E "^==|^value"
