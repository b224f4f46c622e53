"""Prints the single-seed call's kernels from a rocprofv3 kernel_stats.csv (tools/ss_trace.sh)."""
import csv
import sys

for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
    if n.startswith("rwr::"):
        print(f"  {n[:44]:46s} {int(r['Calls']):4d} avg {float(r['AverageNs']) / 1e3:8.1f} us  min {float(r['MinNs']) / 1e3:7.1f}  max {float(r['MaxNs']) / 1e3:7.1f}")
