#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r03x
timeout -k 10 600 python -u -m pytest tests/test_engine_parity.py -k "million or small_counts" -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r03x/pytest.log 2>&1 || { tail -30 gpurun_out/r03x/pytest.log; exit 1; }
tail -2 gpurun_out/r03x/pytest.log
for rep in 1 2; do
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r03x/bench_default.json 2> gpurun_out/r03x/bench_default.err || { tail -20 gpurun_out/r03x/bench_default.err; exit 1; }
python - <<'PY'
import json
j=json.load(open('gpurun_out/r03x/bench_default.json'))
print('c1', j['value'], j['ms_per_step'], {k:v['avg_ms'] for k,v in j['kernels'].items()})
c=j['config3']; print('c3', c['value'], c['job_ms'], c['join_path'], {k:v['avg_ms'] for k,v in c['kernels'].items()})
PY
done
