#!/bin/bash
# routed mode with one rank: the step, and where its time goes
set -o pipefail
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 python bench.py --route --no-cpu-baseline --config3 0 > gpurun_out/route_n1.json 2>gpurun_out/route_err.log || { tail -5 gpurun_out/route_err.log; exit 1; }
python tests/show_bench.py gpurun_out/route_n1.json | grep -v bound
out=gpurun_out/prof_route
rm -rf $out; mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python3 bench.py --route --steps 16 --warmup 8 --no-cpu-baseline --config3 0 --min-timed-ms 20 > $out/bench_under_rocprof.json 2> $out/kt.err || { tail -5 $out/kt.err; exit 1; }
f=$(find $out/kt -name '*kernel_stats.csv' | head -1)
python3 - "$f" $out/kernel_stats.csv <<'PY'
import sys
with open(sys.argv[1]) as f, open(sys.argv[2], "w") as g:
    for i, line in enumerate(f):
        if i == 0 or "dj::" in line or "rccl" in line.lower() or "copyBuffer" in line or "fillBuffer" in line:
            g.write(line)
PY
rm -rf $out/kt
cut -c1-60,200- $out/kernel_stats.csv | head -30
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/prof_route/kernel_stats.csv')))
for r in rows: print("%-50s calls %5s avg %8.1f us total %8.1f ms"%(r['Name'][:50], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6))
PY
