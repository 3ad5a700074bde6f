#!/bin/bash
# the per-request front: tests, then blocking callers with and without linger
set -o pipefail
out=gpurun_out/r03batcher; mkdir -p $out
timeout -k 10 600 python -u -m pytest tests/test_batcher.py tests/test_native_router_ranks_gpu.py -m gpu -x -q --timeout 300 --timeout-method thread > $out/pytest.log 2>&1 || { tail -30 $out/pytest.log; exit 1; }
tail -2 $out/pytest.log
: > $out/batcher_bench.txt
for cfg in "16 2000 1 0" "16 2000 1 100" "64 1000 1 0" "256 400 1 0" "1024 100 1 0" "16 20000 0 100" "16 50000 2 100"; do
  timeout -k 10 120 tools/_build/batcher_bench $cfg >> $out/batcher_bench.txt 2>&1 || { tail -5 $out/batcher_bench.txt; exit 1; }
done
cat $out/batcher_bench.txt
