"""Print a rocprofv3 kernel_stats.csv: python scripts/kstats.py FILE [substring ...] (top 30 by time without substrings)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
sel = [r for r in rows if not sys.argv[2:] or any(s in r["Name"] for s in sys.argv[2:])]
print(f"total {tot / 1e6:.2f} ms over {len(rows)} kernels")
for r in sorted(sel, key=lambda r: -float(r["TotalDurationNs"]))[:None if sys.argv[2:] else 30]:
    print(f'{r["Name"][:84]:84s} calls {int(r["Calls"]):5d} avg {float(r["AverageNs"]) / 1e3:8.1f} us  {100 * float(r["TotalDurationNs"]) / tot:5.1f}%')
