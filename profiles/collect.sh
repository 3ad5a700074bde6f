#!/bin/bash
# Collects the rocprofv3 evidence for bench.py on the GPU box (run from the repo root):
#   gpurun --timeout 900 -- 'bash profiles/collect.sh <tag>'
# Three separate passes (the counter slots do not allow FETCH_SIZE and WRITE_SIZE together, and
# counters never share a run with the kernel trace's --stats):
#   1. --kernel-trace --stats   -> gpurun_out/prof_<tag>/kernel_stats.csv  (+ the bench's own JSON line)
#   2. --pmc FETCH_SIZE         -> gpurun_out/prof_<tag>/pmc_fetch_size.csv
#   3. --pmc WRITE_SIZE         -> gpurun_out/prof_<tag>/pmc_write_size.csv
#   4. --kernel-trace --stats of tools/sweep_bench.py 25 -> gpurun_out/prof_<tag>/sweep_kernel_stats.csv
# Copy the results into profiles/rNN/ afterwards and run profiles/summarize_pmc.py on 2 and 3.
set -u
tag=${1:-x}
root=$(pwd)
out=$root/gpurun_out/prof_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
cd "$root"

keep_dj() {   # header + rows of the engine's kernels
    python3 - "$1" "$2" <<'EOF'
import sys
src, dst = sys.argv[1], sys.argv[2]
with open(src) as f, open(dst, "w") as g:
    for i, line in enumerate(f):
        if i == 0 or "dj::" in line:
            g.write(line)
EOF
}

timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt" -- \
    python3 bench.py --steps 16 --warmup 8 --no-cpu-baseline --config3 0 --min-timed-ms 20 > "$out/bench_under_rocprof.json" 2> "$out/kt.err" || { echo "kernel-trace pass failed"; tail -5 "$out/kt.err"; exit 1; }
f=$(find "$out/kt" -name '*kernel_stats.csv' | head -1)
keep_dj "$f" "$out/kernel_stats.csv"
echo "pass 1 done"

timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$out/pf" -- \
    python3 bench.py --steps 8 --warmup 8 --no-cpu-baseline --no-profile --config3 0 --min-timed-ms 0 > /dev/null 2> "$out/pf.err" || { echo "FETCH_SIZE pass failed"; tail -5 "$out/pf.err"; exit 1; }
f=$(find "$out/pf" -name '*counter_collection.csv' | head -1)
keep_dj "$f" "$out/pmc_fetch_size.csv"
echo "pass 2 done"

timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$out/pw" -- \
    python3 bench.py --steps 8 --warmup 8 --no-cpu-baseline --no-profile --config3 0 --min-timed-ms 0 > /dev/null 2> "$out/pw.err" || { echo "WRITE_SIZE pass failed"; tail -5 "$out/pw.err"; exit 1; }
f=$(find "$out/pw" -name '*counter_collection.csv' | head -1)
keep_dj "$f" "$out/pmc_write_size.csv"
echo "pass 3 done"
# 4. the sweep of a large, mostly quiet table (32 M slots, 64 K due): scan form against ready-list form
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/sw" -- \
    python3 tools/sweep_bench.py 25 > "$out/sweep_bench_under_rocprof.txt" 2> "$out/sw.err" || { echo "sweep pass failed"; tail -5 "$out/sw.err"; exit 1; }
f=$(find "$out/sw" -name '*kernel_stats.csv' | head -1)
keep_dj "$f" "$out/sweep_kernel_stats.csv"
echo "pass 4 done"
rm -rf "$out/kt" "$out/pf" "$out/pw" "$out/sw"
ls -la "$out"
