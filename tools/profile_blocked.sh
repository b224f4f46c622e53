#!/bin/bash
# Counters of the LDS-staged single-seed SpMV experiment (spmv_blocked.hip, RWR_SPMV_BLOCKED=1) next to the shipped kernels,
# MovieLens-shaped configuration C3: L2 requests / hits, fabric bytes, wave-cycle split, LDS bank conflicts.
#   usage (repo root, on the GPU box): bash tools/profile_blocked.sh
set -u
root=$(pwd)
out=$root/gpurun_out/prof_blocked
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for v in blocked shipped; do
  if [ $v = blocked ]; then export RWR_SPMV_BLOCKED=1; else unset RWR_SPMV_BLOCKED; fi
  for grp in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
    name=$(echo $grp | cut -d' ' -f1)
    timeout -k 10 400 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/${v}_$name -o run -- python3 $root/tools/k1_time.py C3 exact > $out/${v}_$name.log 2>&1 || echo "pmc $v $name failed"
  done
  echo "done $v"
done
unset RWR_SPMV_BLOCKED
cd $root
python3 - <<'PY'
import csv, glob, json, os
out = "gpurun_out/prof_blocked"
res = {}
for d in sorted(glob.glob(out + "/*_*")):
    if not os.path.isdir(d):
        continue
    acc = {}
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            if "k_spmv" not in k:
                continue
            short = k.split("(")[0].replace("void ", "")
            e = acc.setdefault(short, {})
            c = e.setdefault(row["Counter_Name"], [0.0, set()])
            c[0] += float(row["Counter_Value"])
            c[1].add(row["Dispatch_Id"])
    res[os.path.basename(d)] = {k: {**{c: v[0] / max(len(v[1]), 1) for c, v in e.items()}, "launches": max(len(v[1]) for v in e.values())} for k, e in acc.items()}
json.dump(res, open(out + "/summary.json", "w"), indent=1)
print(json.dumps(res, indent=1)[:5000])
PY
