#!/bin/bash
# Everything profiles/r02_* holds, collected on the GPU box (repo root):  bash tools/profile_r02.sh
#   1. bench workload (C4): rocprofv3 kernel-trace stats + three counter passes  -> tools/profile_c4.sh
#   2. single-seed call (the kernel BASELINE.json's 70 % target names) on C2 / C3 / C4: kernel-trace stats + counter passes
set -u
root=$(pwd)
bash tools/profile_c4.sh r02_f || exit 1
cd /tmp && export TMPDIR=/tmp
for cfg in C2 C3 C4; do
  out=$root/gpurun_out/prof_single_$cfg
  mkdir -p $out
  timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o run -- python3 $root/tools/single_seed_latency.py $cfg exact > $out/stats.log 2>&1 || { echo "stats $cfg failed"; tail -5 $out/stats.log; exit 1; }
  cp $(find $out/stats -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
  cd $root && bash tools/profile_single_seed.sh $cfg > $out/pmc_print2.log 2>&1 || { echo "pmc $cfg failed"; exit 1; }
  cd /tmp
  echo "done single $cfg"
done
