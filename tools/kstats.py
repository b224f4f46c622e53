"""Per-kernel summary of a rocprofv3 run kept as a rocpd database (`rocprofv3 --kernel-trace -d DIR -- cmd` writes
DIR/<host>/<pid>_results.db):  python tools/kstats.py DIR [name-filter] -> count / avg / min / max / total per kernel."""
import glob
import sqlite3
import sys

root = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for f in sorted(glob.glob(root + "/**/*_results.db", recursive=True)):
    db = sqlite3.connect(f)
    rows = db.execute("select name, count(*), avg(end-start), min(end-start), max(end-start), sum(end-start) from kernels "
                      "group by name order by 6 desc").fetchall()
    print("==", f)
    for r in rows[:40]:
        if flt and flt not in r[0]:
            continue
        print("  %-64s n=%5d avg=%9.1f us min=%9.1f max=%9.1f total=%9.2f ms" % (r[0][:64], r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3, r[5] / 1e6))
