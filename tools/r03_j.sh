#!/bin/bash
for args in "--route --barriers 65536" "--barriers 65536"; do
timeout -k 10 300 python bench.py $args --no-cpu-baseline --config3 0 --no-profile > gpurun_out/small.json 2>gpurun_out/route_err.log || { tail -5 gpurun_out/route_err.log; exit 1; }
python - "$args" <<'PY'
import json,sys
d=json.loads([l for l in open('gpurun_out/small.json') if l.startswith('{')][-1])
print('%-30s ms/step %.4f host issue ms/step %.4f'%(sys.argv[1], d['ms_per_step'], d['host_issue_ms_per_step']))
PY
done
