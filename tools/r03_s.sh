#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r03s
timeout -k 10 600 python -u -m pytest tests/test_engine_parity.py -k "million or small_counts or variable or kilobyte or longer" -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r03s/pytest.log 2>&1 || { tail -30 gpurun_out/r03s/pytest.log; exit 1; }
tail -2 gpurun_out/r03s/pytest.log
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r03s/bench_default.json 2> gpurun_out/r03s/bench_default.err || { tail -20 gpurun_out/r03s/bench_default.err; exit 1; }
python - <<'PY'
import json
j=json.load(open('gpurun_out/r03s/bench_default.json'))
print('c1', j['value'], j['ms_per_step'], {k:v['avg_ms'] for k,v in j['kernels'].items()})
c=j['config3']; print('c3', c['value'], c['job_ms'], c['join_path'], {k:v['avg_ms'] for k,v in c['kernels'].items()})
PY
timeout -k 10 300 python bench.py --no-cpu-baseline --config3 0 --barriers 8388608 --steps 8 --warmup 8 > gpurun_out/r03s/bench_8m.json 2> gpurun_out/r03s/bench_8m.err || { tail -20 gpurun_out/r03s/bench_8m.err; exit 1; }
python - <<'PY'
import json
j=json.load(open('gpurun_out/r03s/bench_8m.json'))
print('8m', j['value'], j['ms_per_step'], {k:v['avg_ms'] for k,v in j['kernels'].items()})
PY
