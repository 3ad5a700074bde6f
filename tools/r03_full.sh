#!/bin/bash
# whole GPU suite, one process
set -o pipefail
mkdir -p gpurun_out/r03full
timeout -k 10 1100 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r03full/pytest.log 2>&1 || { tail -40 gpurun_out/r03full/pytest.log; exit 1; }
tail -3 gpurun_out/r03full/pytest.log
