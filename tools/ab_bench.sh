#!/bin/bash
# A/B of two builds of the engine on the same GPU: tools/ab_bench.sh <base.so> [bench args]
base=$1; shift
for round in 1 2; do
  DJ_ENGINE_LIB=$base timeout -k 10 300 python bench.py --no-cpu-baseline "$@" > gpurun_out/ab_base_$round.json 2>>gpurun_out/ab_err.log || exit 1
  echo "== base (round $round)"; python tests/show_bench.py gpurun_out/ab_base_$round.json | grep -v bound
  timeout -k 10 300 python bench.py --no-cpu-baseline "$@" > gpurun_out/ab_new_$round.json 2>>gpurun_out/ab_err.log || exit 1
  echo "== new (round $round)"; python tests/show_bench.py gpurun_out/ab_new_$round.json | grep -v bound
done
