#!/bin/bash
# round 3, first GPU pass: parity suite, the sweep on a 32 M-slot table, A/B of the engine against round 2's build
set -o pipefail
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/gputest_b.log 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/gputest_b.log
tail -3 gpurun_out/gputest_b.log
grep -q "rc=0" gpurun_out/gputest_b.log || exit 1
timeout -k 10 300 python tools/sweep_bench.py 25 > gpurun_out/sweep_bench_r03a.txt 2>&1 || exit 1
cat gpurun_out/sweep_bench_r03a.txt
bash tools/ab_bench.sh tools/_build/libdj_engine_r02.so --config3 0 || exit 1
