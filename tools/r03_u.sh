#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r03u
for p in 64 48 100; do
timeout -k 10 300 python bench.py --no-cpu-baseline --config3 0 --payload $p > gpurun_out/r03u/p_$p.json 2> gpurun_out/r03u/p.err || { tail -20 gpurun_out/r03u/p.err; exit 1; }
python - $p <<'PY'
import json,sys
j=json.load(open(f'gpurun_out/r03u/p_{sys.argv[1]}.json'))
print(sys.argv[1], round(j['value']/1e9,3), round(j['ms_per_step'],4), {k:round(v['avg_ms']*1000,1) for k,v in j['kernels'].items()})
PY
done
