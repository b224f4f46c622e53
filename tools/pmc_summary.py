"""Aggregates rocprofv3 counter passes (one --pmc pass per counter group, --output-format csv) into the per-kernel
summary committed under profiles/, and the launch-weighted HBM-side bytes per SpMM launch that bench.py reports as
roofline.traffic.

    python tools/pmc_summary.py <tag> <out_summary.json> <out_traffic.json> <pass_dir> [<pass_dir> ...]

Corrections (profiles/r01_calibration_fetch_size.json, MI355X_MICROARCH.md "HBM"): FETCH_SIZE is in KB and reports
half of the bytes on gfx950 for this project's access widths (x2); WRITE_SIZE (KB) is exact; Infinity-Cache hits are
included in FETCH_SIZE."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name: str) -> str:
    name = re.sub(r"^void ", "", name).replace("(anonymous namespace)::", "")
    return name.split("(")[0]


def main():
    tag, out_summary, out_traffic = sys.argv[1:4]
    acc = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))   # kernel -> counter -> dispatch -> value
    for d in sys.argv[4:]:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(f, newline="") as fh:
                for row in csv.DictReader(fh):
                    k = short(row["Kernel_Name"])
                    if not k.startswith("rwr::"):
                        continue
                    acc[k][row["Counter_Name"]][(f, row["Dispatch_Id"])] += float(row["Counter_Value"])
    kernels = {}
    for k, counters in sorted(acc.items()):
        e = {}
        for cname, per in counters.items():
            e[f"{cname}_per_launch"] = sum(per.values()) / len(per)
            e[f"launches_in_pass_{cname}"] = len(per)
        if e.get("TCC_REQ_sum_per_launch"):
            e["l2_hit_rate"] = e.get("TCC_HIT_sum_per_launch", 0.0) / e["TCC_REQ_sum_per_launch"]
        if "FETCH_SIZE_per_launch" in e:
            e["hbm_side_bytes_per_launch_corrected"] = (2.0 * e["FETCH_SIZE_per_launch"] + e.get("WRITE_SIZE_per_launch", 0.0)) * 1024.0
        kernels[k] = e
    summary = {"config": tag,
               "correction": "FETCH_SIZE (KB) x 2 (profiles/r01_calibration_fetch_size.json); WRITE_SIZE (KB) exact; "
                             "Infinity-Cache hits are included in FETCH_SIZE",
               "kernels": kernels}
    json.dump(summary, open(out_summary, "w"), indent=1)
    # the DENSE variant only (template arguments CHECK = false, WRITE = false): the launches bench.py prices
    spmm = {k: v for k, v in kernels.items() if k.startswith("rwr::k_spmm_chunked") and ", false, false," in k
            and "hbm_side_bytes_per_launch_corrected" in v}
    launches = sum(v["launches_in_pass_FETCH_SIZE"] for v in spmm.values())
    if launches:
        total = sum(v["hbm_side_bytes_per_launch_corrected"] * v["launches_in_pass_FETCH_SIZE"] for v in spmm.values())
        m = re.search(r"tile width (\d+)", tag)
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench
        json.dump({"config": tag.split()[0], "mode": "exact" if "exact" in tag else "fast",
                   "tile_seeds": int(m.group(1)) if m else 0,
                   "csrc_sha": bench.kernel_source_sha(),
                   "kernel": "k_spmm_chunked, dense launches (no frontier skipping)", "launches": launches,
                   "bytes_per_launch": total / launches,
                   "source": f"{os.path.basename(out_summary)} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; FETCH_SIZE x2 per "
                             "profiles/r01_calibration_fetch_size.json; includes Infinity-Cache hits)"},
                  open(out_traffic, "w"), indent=1)
    print(json.dumps({k: {c: round(v, 1) for c, v in e.items()} for k, e in kernels.items() if "spmm" in k or "cs_" in k}, indent=1))


if __name__ == "__main__":
    main()
