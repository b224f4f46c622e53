#!/bin/bash
# Refresh of the single-seed evidence after a change to the chain / SpMV kernels: kernel-trace stats + counter passes of the
# single-seed call on C3 and C2, then the C2 / C3 bench lines.   bash tools/profile_single_final.sh
set -u
root=$(pwd)
cd /tmp && export TMPDIR=/tmp
for cfg in C3 C2; do
  out=$root/gpurun_out/prof_single_$cfg
  rm -rf $out; mkdir -p $out
  timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o run -- python3 $root/tools/single_seed_latency.py $cfg > $out/stats.log 2>&1 || { echo "stats $cfg failed"; tail -5 $out/stats.log; exit 1; }
  cp $(find $out/stats -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
  cd $root && bash tools/profile_single_seed.sh $cfg > $out/pmc_print2.log 2>&1 || { echo "pmc $cfg failed"; exit 1; }
  cd /tmp
  echo "done single $cfg"
done
cd $root
mkdir -p gpurun_out/final3
for cfg in C2 C3; do python3 bench.py --config $cfg --steps 3 --warmup 1 > gpurun_out/final3/$cfg.json 2> gpurun_out/final3/$cfg.log || echo "$cfg failed"; echo "done $cfg line"; done
