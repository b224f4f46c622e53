#!/bin/bash
# Everything profiles/r03_* holds, collected on the GPU box (repo root):  bash tools/profile_r03.sh
#   1. bench workload (C4): rocprofv3 kernel-trace stats + three counter passes  -> tools/profile_c4.sh
#   2. single-seed call on C3 (the sweep kernel's configuration) and C4: kernel-trace stats + counter passes
#   3. bench lines: C4 (default command), C2, C3, C1 (ego networks from 10 host threads), C5 seed path and row-partitioned
set -u
root=$(pwd)
bash tools/profile_c4.sh r03_f || exit 1
cd /tmp && export TMPDIR=/tmp
for cfg in C3 C2; do
  out=$root/gpurun_out/prof_single_$cfg
  mkdir -p $out
  timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o run -- python3 $root/tools/single_seed_latency.py $cfg > $out/stats.log 2>&1 || { echo "stats $cfg failed"; tail -5 $out/stats.log; exit 1; }
  cp $(find $out/stats -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
  cd $root && bash tools/profile_single_seed.sh $cfg > $out/pmc_print2.log 2>&1 || { echo "pmc $cfg failed"; exit 1; }
  cd /tmp
  echo "done single $cfg"
done
cd $root
out=$root/gpurun_out/r03_lines
mkdir -p $out
python3 bench.py > $out/C4.json 2> $out/C4.log || echo "C4 line failed"
echo "done C4 line"
for cfg in C2 C3; do python3 bench.py --config $cfg --steps 3 --warmup 1 > $out/$cfg.json 2> $out/$cfg.log || echo "$cfg failed"; echo "done $cfg line"; done
python3 bench.py --config C1 --steps 3 --warmup 1 > $out/C1.json 2> $out/C1.log || echo "C1 failed"
echo "done C1 line"
python3 bench.py --config C5 --steps 2 --warmup 1 --cpu-threads 4 --cpu-seeds 2 > $out/C5.json 2> $out/C5.log || echo "C5 failed"
echo "done C5 line"
python3 bench.py --config C5 --partition rows --steps 3 --warmup 1 --no-cpu-baseline > $out/C5_rows.json 2> $out/C5_rows.log || echo "C5 rows failed"
echo "done C5 rows line"
