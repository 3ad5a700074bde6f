#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r03y
timeout -k 10 900 python -u -m pytest tests/test_engine_parity.py -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r03y/pytest.log 2>&1 || { tail -30 gpurun_out/r03y/pytest.log; exit 1; }
tail -2 gpurun_out/r03y/pytest.log
for a in "" "--join-max-wait 30" "--join-max-wait 100000"; do
timeout -k 10 300 python bench.py --no-cpu-baseline --config3 0 $a > gpurun_out/r03y/b.json 2> gpurun_out/r03y/b.err || { tail -20 gpurun_out/r03y/b.err; exit 1; }
python - "$a" <<'PY'
import json,sys
j=json.load(open('gpurun_out/r03y/b.json'))
print(sys.argv[1], round(j['value']/1e9,3), round(j['ms_per_step'],4), {k:round(v['avg_ms']*1000,1) for k,v in j['kernels'].items()})
PY
done
