#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_route_tl
rm -rf $out; mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/kt -- python3 bench.py --route --steps 16 --warmup 8 --no-cpu-baseline --config3 0 --min-timed-ms 20 --no-profile > $out/bench.json 2> $out/kt.err || { tail -5 $out/kt.err; exit 1; }
f=$(find $out/kt -name '*kernel_trace.csv' | head -1)
python tools/timeline.py "$f" 12 70
rm -rf $out/kt
