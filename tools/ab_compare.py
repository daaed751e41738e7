"""Per-kernel comparison of two rocprofv3 kernel-stats files (tools/ab_libs.sh):  python tools/ab_compare.py a.csv b.csv [steps]"""
import csv
import sys

def load(p):
    return {r["Name"]: (float(r["TotalDurationNs"]), int(r["Calls"])) for r in csv.DictReader(open(p))}

a, b = load(sys.argv[1]), load(sys.argv[2])
steps = float(sys.argv[3]) if len(sys.argv) > 3 else 13.0
ta, tb = sum(v[0] for v in a.values()), sum(v[0] for v in b.values())
print(f"total per step: {ta / steps / 1e6:.3f} ms vs {tb / steps / 1e6:.3f} ms")
names = sorted(set(a) | set(b), key=lambda k: -(a.get(k, (0, 0))[0] + b.get(k, (0, 0))[0]))
for k in names[:40]:
    xa, xb = a.get(k, (0, 0)), b.get(k, (0, 0))
    if abs(xa[0] - xb[0]) / steps < 2e3 and xa[1] == xb[1]:
        continue
    print(f"{xa[0] / steps / 1e6:8.3f} ({xa[1] / steps:4.1f}) -> {xb[0] / steps / 1e6:8.3f} ({xb[1] / steps:4.1f}) ms/step  {k[:110]}")
