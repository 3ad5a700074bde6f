#!/bin/bash
# bench.py's N>1 flow rehearsed with 4 and 8 ranks on the one GPU (gloo collectives)
set -o pipefail
mkdir -p gpurun_out/r03v
for n in 4 8; do
timeout -k 10 280 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500 + n)) bench.py --gpus $n --rehearse-on-one-gpu --barriers 65536 --steps 16 --warmup 8 --no-cpu-baseline --config3 0 --min-timed-ms 50 > gpurun_out/r03v/n$n.json 2> gpurun_out/r03v/n$n.err || { tail -30 gpurun_out/r03v/n$n.err; exit 1; }
python - $n <<'PY'
import json,sys
lines=[l for l in open(f'gpurun_out/r03v/n{sys.argv[1]}.json') if l.startswith('{')]
assert len(lines)==1, lines
j=json.loads(lines[0])
print(sys.argv[1], j['n_gpus'], round(j['value']/1e9,3), j['ms_per_step'], j['continuations_fired'], j['timed_steps'], j['config']['sharding'])
PY
done
