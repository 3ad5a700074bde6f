#!/bin/bash
set -o pipefail
timeout -k 10 500 python -u -m pytest tests/test_router_two_ranks_gpu.py tests/test_bench_two_ranks_gpu.py "tests/test_engine_more.py::test_router_single_rank_roundtrip" -m gpu -x -q --timeout 240 --timeout-method thread > gpurun_out/gputest_i.log 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/gputest_i.log
tail -5 gpurun_out/gputest_i.log
grep -q "rc=0" gpurun_out/gputest_i.log || exit 1
bash tools/r03_g.sh
