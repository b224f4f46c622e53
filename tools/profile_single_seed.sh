#!/bin/bash
# Counter passes (FETCH_SIZE, WRITE_SIZE, TCC hit/miss/req -- one rocprofv3 invocation per group) of the single-seed call
# (tools/single_seed_latency.py <config> exact), summarised per kernel by tools/pmc_summary.py.
#   usage (repo root, on the GPU box):  bash tools/profile_single_seed.sh C2
set -u
cfg=${1:-C2}
root=$(pwd)
out=$root/gpurun_out/prof_single_$cfg
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for grp in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  name=$(echo $grp | cut -d' ' -f1)
  timeout -k 10 500 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/pmc_$name -o run -- python3 $root/tools/single_seed_latency.py $cfg > $out/pmc_$name.log 2>&1 || { echo "pmc $name failed"; tail -5 $out/pmc_$name.log; exit 1; }
done
cd $root
python3 tools/pmc_summary.py "$cfg single seed exact (tools/single_seed_latency.py)" $out/pmc_summary.json $out/traffic_unused.json $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE $out/pmc_TCC_HIT_sum > $out/pmc_print.log
python3 - <<PY
import json
s=json.load(open("$out/pmc_summary.json"))
for k,v in s["kernels"].items():
    if "spmv" in k or "cs_" in k:
        print(k, {a: (round(b/1e6,2) if "bytes" in a else round(b,1)) for a,b in v.items() if "launches" not in a})
PY
