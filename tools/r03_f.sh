#!/bin/bash
set -o pipefail
timeout -k 10 500 python -u -m pytest tests/test_native_router_ranks_gpu.py tests/test_router_two_ranks_gpu.py tests/test_bench_two_ranks_gpu.py "tests/test_engine_more.py::test_router_single_rank_roundtrip" "tests/test_engine_more.py::test_native_router_single_rank_roundtrip" "tests/test_engine_more.py::test_partition_kernel_is_a_stable_partition_by_owner" -m gpu -x -q --timeout 240 --timeout-method thread > gpurun_out/gputest_f.log 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/gputest_f.log
tail -30 gpurun_out/gputest_f.log
grep -q "rc=0" gpurun_out/gputest_f.log || exit 1
bash tools/r03_e.sh
