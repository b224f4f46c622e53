#!/bin/bash
# Kernel-trace summary of the single-seed call on the GPU box:  bash tools/ss_trace.sh <tag> C2 C3  -> gpurun_out/ss_<tag>_<cfg>/
tag=$1; shift
root=$(pwd)
cd /tmp && export TMPDIR=/tmp
for c in "$@"; do
  timeout -k 10 250 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/ss_${tag}_$c -o run -- python3 $root/tools/single_seed_latency.py $c > $root/gpurun_out/ss_${tag}_$c.log 2>&1 || echo "trace $c failed"
  grep -E "seed [0-9]+:" $root/gpurun_out/ss_${tag}_$c.log
  python3 $root/tools/ss_trace_print.py $(find $root/gpurun_out/ss_${tag}_$c -name "*kernel_stats.csv" | head -1)
done
