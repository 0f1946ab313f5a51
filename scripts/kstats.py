"""Per-kernel GPU durations out of a rocprofv3 run (--kernel-trace; the rocpd .db or the *_kernel_trace.csv), grouped by (kernel, grid) in
launch order -- the host-side timers of the microbenchmarks cannot resolve launches of a few tens of microseconds.
   python scripts/kstats.py <dir or file> [name filter]"""
import csv, glob, os, sqlite3, sys
from collections import OrderedDict

path = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
rows = []
files = [path] if os.path.isfile(path) else sorted(glob.glob(os.path.join(path, "**", "*_results.db"), recursive=True)) + sorted(glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True))
for f in files:
    if f.endswith(".db"):
        db = sqlite3.connect(f)
        rows += db.execute("select name, grid_x / workgroup_x, end - start, start from kernels").fetchall()
    else:
        for r in csv.DictReader(open(f)):
            rows.append((r["Kernel_Name"], int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), int(r["Start_Timestamp"])))
rows.sort(key=lambda r: r[3])
g = OrderedDict()
for n, wg, d, _ in rows:
    if flt in n:
        g.setdefault((n.replace("(anonymous namespace)::", "").replace("void ", "")[:60], wg), []).append(d)
for (n, wg), v in g.items():
    v.sort()
    print(f"{n:60s} grid {wg:6d}  x{len(v):4d}  median {v[len(v) // 2] / 1e3:8.1f} us  min {v[0] / 1e3:8.1f}")
