#!/bin/bash
# End-of-round refresh after a change to the dominant kernel's sources: C4 stats + counters (-> profiles/traffic_C4.json),
# then the default bench line and the C5 row-partitioned line.   bash tools/profile_final.sh <tag>
set -u
tag=${1:-r03_g}
root=$(pwd)
bash tools/profile_c4.sh $tag || exit 1
cp gpurun_out/prof_$tag/traffic_C4.json profiles/traffic_C4.json     # (so that the lines below report roofline.traffic / fabric)
mkdir -p gpurun_out/final_lines
python3 bench.py > gpurun_out/final_lines/C4.json 2> gpurun_out/final_lines/C4.log || echo "C4 line failed"
python3 bench.py --config C5 --partition rows --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/final_lines/C5_rows.json 2> gpurun_out/final_lines/C5_rows.log || echo "C5 rows failed"
python3 bench.py --config C1 --steps 5 --warmup 2 > gpurun_out/final_lines/C1.json 2> gpurun_out/final_lines/C1.log || echo "C1 failed"
echo done
