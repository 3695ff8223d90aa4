"""Print the top kernels (per-step time) from a rocprofv3 kernel_stats.csv.  usage: top_kernels.py stats.csv n_steps"""
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time per step: {tot / n / 1e6:.3f} ms, launches per step: {sum(int(r['Calls']) for r in rows) / n:.0f}")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:22]:
    name = r["Name"].split("(")[0].replace("void ", "")[:60]
    print(f"{name:60s} calls/step {int(r['Calls']) / n:7.1f}  avg {float(r['AverageNs']) / 1e3:8.1f} us  per-step {float(r['TotalDurationNs']) / n / 1e6:7.3f} ms  {float(r['Percentage']):5.1f}%")
