"""Per-step kernel time table from a rocprofv3 --kernel-trace --stats run of bench.py:
   python tools/dev/kstats.py <dir> <steps incl. warm-up> [rows]"""
import csv, sys, glob
f = glob.glob(sys.argv[1] + '/*/*kernel_stats.csv')[0]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 4
rows = list(csv.DictReader(open(f)))
ours = [r for r in rows if "at::native" not in r["Name"] and "rocclr" not in r["Name"]]
tot = sum(float(r['TotalDurationNs']) for r in ours)
print(f"our kernels: {tot/1e6/steps:.3f} ms/step")
for r in ours[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    print(f"{r['Name'][:86]:86s} n/step={float(r['Calls'])/steps:5.1f} ms/step={float(r['TotalDurationNs'])/1e6/steps:7.3f} avg_us={float(r['AverageNs'])/1e3:8.1f}")
