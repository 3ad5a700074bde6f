#!/bin/bash
# The bench with each of several builds of the engine on the same GPU: tools/variants_bench.sh "<lib> <lib> ..." [bench args]
libs=$1; shift
for round in 1 2; do
  for lib in $libs; do
    DJ_ENGINE_LIB=$lib timeout -k 10 300 python bench.py --no-cpu-baseline "$@" > gpurun_out/var_$(basename $lib .so)_$round.json 2>>gpurun_out/var_err.log || exit 1
    echo "== $lib (round $round)"; python tests/show_bench.py gpurun_out/var_$(basename $lib .so)_$round.json | grep -v bound
  done
done
