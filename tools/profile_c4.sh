#!/bin/bash
# Collects, on the GPU box, what profiles/ holds for the bench workload: rocprofv3 kernel-trace stats of
# `bench.py --config C4 --steps 2 --warmup 1`, then three counter passes (FETCH_SIZE, WRITE_SIZE, TCC hit/miss) of a
# one-step run, each in its own rocprofv3 invocation (counters never together with the stats run).
#   usage (from the repo root, on the box):  bash tools/profile_c4.sh <tag>
set -u
tag=${1:-r02_f}
root=$(pwd)
out=$root/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o run -- python3 $root/bench.py --config C4 --steps 2 --warmup 1 --no-cpu-baseline > $out/bench_under_rocprof.json 2> $out/stats.log || { echo "stats run failed"; tail -5 $out/stats.log; exit 1; }
for grp in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  name=$(echo $grp | cut -d' ' -f1)
  timeout -k 10 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/pmc_$name -o run -- python3 $root/bench.py --config C4 --steps 1 --warmup 0 --no-cpu-baseline > $out/pmc_$name.json 2> $out/pmc_$name.log || { echo "pmc $name failed"; tail -5 $out/pmc_$name.log; exit 1; }
done
cd $root
python3 tools/pmc_summary.py "C4 exact, K=1024, T=10, tile width 32 (bench.py --config C4)" $out/pmc_summary.json $out/traffic_C4.json $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE $out/pmc_TCC_HIT_sum > $out/pmc_print.log
cp $(find $out/stats -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
find $out -name "*kernel_stats.csv" | head -3
cat $out/bench_under_rocprof.json | tail -1 | cut -c1-600
