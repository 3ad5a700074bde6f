#!/bin/bash
timeout -k 10 300 python bench.py --route --no-cpu-baseline --config3 0 --no-profile > gpurun_out/route_n1b.json 2>gpurun_out/route_err.log || { tail -5 gpurun_out/route_err.log; exit 1; }
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/route_n1b.json') if l.startswith('{')][-1])
print('routed n=1: ms/step %.4f host issue ms/step %.4f value %.3e'%(d['ms_per_step'], d['host_issue_ms_per_step'], d['value']))
PY
timeout -k 10 300 python bench.py --no-cpu-baseline --config3 0 --no-profile > gpurun_out/plain_n1b.json 2>>gpurun_out/route_err.log || exit 1
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/plain_n1b.json') if l.startswith('{')][-1])
print('plain  n=1: ms/step %.4f host issue ms/step %.4f value %.3e'%(d['ms_per_step'], d['host_issue_ms_per_step'], d['value']))
PY
