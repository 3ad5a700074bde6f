#!/bin/bash
set -o pipefail
timeout -k 10 500 python -u -m pytest tests/test_native_router_ranks_gpu.py tests/test_batcher.py tests/test_router_two_ranks_gpu.py -m gpu -x -q --timeout 240 --timeout-method thread > gpurun_out/gputest_d.log 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/gputest_d.log
tail -30 gpurun_out/gputest_d.log
