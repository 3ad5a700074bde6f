#!/bin/bash
set -o pipefail
export DJ_GATHER_FLAT=1
timeout -k 10 600 python -u -m pytest tests/test_engine_parity.py tests/test_engine_more.py tests/test_engine_golden.py tests/test_engine_sweep_paths.py -m gpu -x -q --timeout 240 --timeout-method thread > gpurun_out/gputest_m.log 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/gputest_m.log
tail -3 gpurun_out/gputest_m.log
grep -q "rc=0" gpurun_out/gputest_m.log || exit 1
for flat in 0 1; do
export DJ_GATHER_FLAT=$flat
echo "== DJ_GATHER_FLAT=$flat"
timeout -k 10 400 python bench.py --no-cpu-baseline > gpurun_out/bench_m$flat.json 2>>gpurun_out/big_err.log || exit 1
python - $flat <<'PY'
import json,sys
d=json.loads([l for l in open('gpurun_out/bench_m%s.json'%sys.argv[1]) if l.startswith('{')][-1])
print('configs[1] %.3e ms/step %.4f gather %.4f'%(d['value'],d['ms_per_step'],d['kernels']['gather']['avg_ms']))
c3=d['config3']
print('config3 value %.3e job_ms %.2f gather %s pop %s'%(c3['value'],c3['job_ms'],c3['kernels']['gather'],c3['kernels']['pop']))
PY
done
