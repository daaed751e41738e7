"""Per-step summary of a rocprofv3 `*_kernel_stats.csv`:  python tools/kstats.py <csv> <steps incl. warm-up> [rows]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"{len(rows)} kernels, {tot / 1e6 / steps:.3f} ms per step")
for r in rows[:top]:
    print(f"{float(r['TotalDurationNs']) / 1e6 / steps:8.3f} ms/step {int(r['Calls']) / steps:6.1f} calls  avg {float(r['AverageNs']) / 1e3:8.1f} us  {r['Name'][:120]}")
