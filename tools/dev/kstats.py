import csv, sys, glob
f = glob.glob(sys.argv[1] + '/*/*kernel_stats.csv')[0]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 4
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f"total {tot/1e6/steps:.2f} ms/step")
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 22]:
    print(f"{r['Name'][:80]:80s} calls/step={float(r['Calls'])/steps:6.1f} ms/step={float(r['TotalDurationNs'])/1e6/steps:7.3f} avg_us={float(r['AverageNs'])/1e3:9.1f} pct={float(r['Percentage']):5.1f}")
