#!/bin/bash
# final evidence of the round: rocprofv3 passes, routed profile, plain bench runs
set -o pipefail
root=$(pwd); out=$root/gpurun_out/final; mkdir -p $out
bash profiles/collect.sh z || exit 1
cd /tmp && export TMPDIR=/tmp && cd $root
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/rt -- python3 bench.py --route --no-cpu-baseline --config3 0 --no-profile > $out/bench_route_under_rocprof.json 2> $out/rt.err || { tail -5 $out/rt.err; exit 1; }
f=$(find $out/rt -name '*kernel_stats.csv' | head -1)
grep -E "^\"Name\"|dj::" $f > $out/bench_route_n1_kernel_stats.csv
rm -rf $out/rt
echo "route profile done"
timeout -k 10 300 python bench.py --route --no-cpu-baseline --config3 0 > $out/bench_route_n1.json 2> $out/route.err || { tail -5 $out/route.err; exit 1; }
echo "route bench done"
timeout -k 10 400 python bench.py > $out/bench_n1_final.json 2> $out/bench.err || { tail -5 $out/bench.err; exit 1; }
echo "default bench done"
timeout -k 10 300 python bench.py --no-cpu-baseline --config3 0 --barriers 8388608 --steps 8 --warmup 8 > $out/bench_8m.json 2> $out/b8.err || { tail -5 $out/b8.err; exit 1; }
timeout -k 10 300 python tools/sweep_bench.py 25 > $out/sweep_bench.txt 2> $out/sw.err || { tail -5 $out/sw.err; exit 1; }
ls -la $out
