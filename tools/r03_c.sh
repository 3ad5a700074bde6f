#!/bin/bash
# the bucketed join passes with the first pass's stash: parity, then timing on a small and a large table and configs[2]
set -o pipefail
timeout -k 10 600 python -u -m pytest tests/test_engine_parity.py tests/test_engine_more.py tests/test_engine_golden.py tests/test_zz_no_alloc.py -m gpu -x -q --timeout 240 --timeout-method thread > gpurun_out/gputest_c.log 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/gputest_c.log
tail -3 gpurun_out/gputest_c.log
grep -q "rc=0" gpurun_out/gputest_c.log || exit 1
bash tools/r03_b.sh || exit 1
echo "== default bench with configs[2]"
timeout -k 10 400 python bench.py --no-cpu-baseline > gpurun_out/bench_c.json 2>>gpurun_out/big_err.log || exit 1
python tests/show_bench.py gpurun_out/bench_c.json | grep -v bound
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/bench_c.json') if l.startswith('{')][-1])
c3=d['config3']
print('config3 value %.3e job_ms %.2f join_path %s'%(c3['value'],c3['job_ms'],c3['join_path']))
for k,v in c3['kernels'].items(): print('  ',k,v)
PY
echo "== 8 M barriers per job (1 GB table, batches of 8 M events)"
timeout -k 10 400 python bench.py --no-cpu-baseline --config3 0 --barriers 8388608 --steps 8 --warmup 8 > gpurun_out/bench_8m.json 2>>gpurun_out/big_err.log || exit 1
python tests/show_bench.py gpurun_out/bench_8m.json | grep -v bound
