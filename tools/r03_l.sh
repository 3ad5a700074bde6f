#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_sweep
rm -rf $out; mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/sw -- python3 tools/sweep_bench.py 25 > $out/sweep_bench_under_rocprof.txt 2> $out/sw.err || { tail -5 $out/sw.err; exit 1; }
f=$(find $out/sw -name '*kernel_stats.csv' | head -1)
grep -E "Name|dj::" "$f" > $out/sweep_kernel_stats.csv
rm -rf $out/sw
cat $out/sweep_bench_under_rocprof.txt
python - <<'PY'
import csv
for r in csv.DictReader(open('gpurun_out/prof_sweep/sweep_kernel_stats.csv')):
    print("%-40s calls %5s avg %8.1f min %8.1f max %8.1f"%(r['Name'].split('(')[0][-40:], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
PY
