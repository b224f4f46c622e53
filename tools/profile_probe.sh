#!/bin/bash
# C4 locality experiment (DESIGN.md "One decisive locality experiment"): dense SpMM launch time and L2 hit rate of
#   base   : the shipped kernel (every XCD's L2 faces the whole source distribution of a tile)
#   probe1 : source-sliced gather pattern, 8 equal-width node slices, XCD x gathers slice x only   (timing-only, wrong sums)
#   probe2 : the same with slices of equal gather mass
# plus the same counters for the C3 SpMM.  Usage (on the GPU box, repo root): bash tools/profile_probe.sh <tag>
set -u
tag=${1:-r02}
root=$(pwd)
out=$root/gpurun_out/probe_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for v in base probe1 probe2; do
  case $v in base) p=0;; probe1) p=1;; probe2) p=2;; esac
  RWR_SPMM_SLICE_PROBE=$p python3 $root/bench.py --config C4 --steps 1 --warmup 1 --no-cpu-baseline > $out/time_$v.json 2> $out/time_$v.log || echo "timing $v failed"
  export RWR_SPMM_SLICE_PROBE=$p
  timeout -k 10 500 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $out/pmc_$v -o run -- python3 $root/bench.py --config C4 --steps 1 --warmup 0 --seeds-per-gpu 256 --no-cpu-baseline > $out/pmc_$v.json 2> $out/pmc_$v.log || echo "pmc $v failed"
  unset RWR_SPMM_SLICE_PROBE
  echo "done $v"
done
for grp in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TCP_TA_DATA_STALL_CYCLES_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD"; do
  name=$(echo $grp | cut -d' ' -f1)
  timeout -k 10 500 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/c3_$name -o run -- python3 $root/bench.py --config C3 --steps 1 --warmup 0 --seeds-per-gpu 1024 --no-cpu-baseline > $out/c3_$name.json 2> $out/c3_$name.log || echo "c3 pmc $name failed"
  echo "done c3 $name"
done
cd $root
python3 - <<'PY'
import csv, glob, json, os, sys
out = sorted(glob.glob("gpurun_out/probe_*"))[-1]
res = {}
for d in sorted(glob.glob(out + "/pmc_*")) + sorted(glob.glob(out + "/c3_*")):
    if not os.path.isdir(d):
        continue
    acc = {}
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            if "k_spmm" not in k:
                continue
            short = k.split("(")[0].replace("void ", "")
            e = acc.setdefault(short, {})
            c = e.setdefault(row["Counter_Name"], [0.0, set()])
            c[0] += float(row["Counter_Value"])
            c[1].add(row["Dispatch_Id"])
    res[os.path.basename(d)] = {k: {c: v[0] / max(len(v[1]), 1) for c, v in e.items()} | {"launches": max(len(v[1]) for v in e.values())} for k, e in acc.items()}
for v in ("base", "probe1", "probe2"):
    try:
        j = json.load(open(f"{out}/time_{v}.json"))
        res["time_" + v] = {"dense_avg_launch_ms": j["roofline"]["avg_launch_ms"], "seeds_per_s": j["value"]}
    except Exception as e:
        res["time_" + v] = str(e)
json.dump(res, open(out + "/summary.json", "w"), indent=1)
print(json.dumps(res, indent=1)[:6000])
PY
