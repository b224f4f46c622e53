"""Timeline of ONE single-seed call from a rocprofv3 kernel trace (csv): kernels of the last call in start order with the gap
in front of each.   rocprofv3 --kernel-trace --output-format csv -d DIR -o run -- python3 tools/single_call_trace.py C2 ;
python3 tools/call_timeline.py DIR/run_kernel_trace.csv"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last call = from the last k_init_seeds on
idx = [i for i, r in enumerate(rows) if "k_init_seeds" in r["Kernel_Name"]]
rows = rows[idx[-1] - 3:] if idx else rows[-80:]
t0 = int(rows[0]["Start_Timestamp"])
prev_end = t0
tot_k = 0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("void ", "").replace("rwr::", "").replace("(anonymous namespace)::", "").split("(")[0][:44]
    print(f"{(s - t0) / 1e3:9.1f} us  +gap {(s - prev_end) / 1e3:7.1f}  {name:46s} {(e - s) / 1e3:8.1f} us  stream/queue {r.get('Queue_Id', '?')}")
    prev_end = max(prev_end, e)
    tot_k += e - s
print(f"span {(prev_end - t0) / 1e3:.1f} us, kernel time summed {tot_k / 1e3:.1f} us")
