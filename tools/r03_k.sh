#!/bin/bash
set -o pipefail
timeout -k 10 700 python -u -m pytest tests -m gpu -x -q --timeout 240 --timeout-method thread > gpurun_out/gputest_k.log 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/gputest_k.log
tail -5 gpurun_out/gputest_k.log
grep -q "rc=0" gpurun_out/gputest_k.log || exit 1
bash tools/r03_e.sh > gpurun_out/route_profile.txt 2>&1; tail -22 gpurun_out/route_profile.txt
