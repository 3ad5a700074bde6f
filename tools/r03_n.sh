#!/bin/bash
# fat buckets: parity, then configs[1] + config3, then the 8M-barrier job
set -o pipefail
mkdir -p gpurun_out/r03n
timeout -k 10 900 python -u -m pytest tests/test_engine_parity.py tests/test_engine_more.py tests/test_engine_golden.py -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r03n/pytest.log 2>&1 || { tail -30 gpurun_out/r03n/pytest.log; exit 1; }
tail -3 gpurun_out/r03n/pytest.log
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r03n/bench_default.json 2> gpurun_out/r03n/bench_default.err || { tail -20 gpurun_out/r03n/bench_default.err; exit 1; }
cat gpurun_out/r03n/bench_default.json
timeout -k 10 300 python bench.py --no-cpu-baseline --config3 0 --barriers 8388608 --steps 8 --warmup 8 > gpurun_out/r03n/bench_8m.json 2> gpurun_out/r03n/bench_8m.err || { tail -20 gpurun_out/r03n/bench_8m.err; exit 1; }
cat gpurun_out/r03n/bench_8m.json
