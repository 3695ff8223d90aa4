"""Per-kernel-group averages of rocprofv3 --pmc counters (counter_collection.csv): launches of kernels whose name contains <part>, in
groups of <per> consecutive dispatches, first <skip> of each group dropped.  Usage: python tools/probes/r6/pmc_avg.py <csv> <part> <per> [skip]"""
import csv
import sys
from collections import defaultdict
f, part, per = sys.argv[1], sys.argv[2], int(sys.argv[3])
skip = int(sys.argv[4]) if len(sys.argv) > 4 else 5
disp = defaultdict(dict)
order = []
for r in csv.DictReader(open(f)):
    if part not in r["Kernel_Name"]:
        continue
    d = int(r["Dispatch_Id"])
    if d not in disp:
        order.append(d)
    disp[d][r["Counter_Name"]] = disp[d].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
order.sort()
for k in range(len(order) // per):
    grp = order[k * per + skip:(k + 1) * per]
    names = sorted(disp[grp[0]])
    print(k, " ".join(f"{n}={sum(disp[d][n] for d in grp) / len(grp):.4g}" for n in names))
