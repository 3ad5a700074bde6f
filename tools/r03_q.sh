#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r03q
for m in "96 1000" "96 32" "64 32" "48 24" "64 1000" "32 1000"; do
set -- $m
DJ_X_A=$1 DJ_X_B=$2 timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r03q/h_$1_$2.json 2> gpurun_out/r03q/h.err || { tail -20 gpurun_out/r03q/h.err; exit 1; }
python - $1_$2 <<'PY'
import json,sys
j=json.load(open(f'gpurun_out/r03q/h_{sys.argv[1]}.json'))
c=j['config3']
print(sys.argv[1], 'c1', round(j['value']/1e9,3), j['ms_per_step'], 'gather', j['kernels']['gather']['avg_ms'], 'copy', j['roofline']['measured_copy_GBps'], 'c3', round(c['value']/1e9,3), c['job_ms'], 'gather', c['kernels']['gather']['avg_ms'])
PY
done
