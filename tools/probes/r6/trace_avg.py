"""Average duration of the launches of one kernel family in a rocprofv3 kernel trace, in groups of `per` consecutive launches (the first
`skip` of each group dropped).  Usage: python tools/probes/r6/trace_avg.py <kernel_trace.csv> <name part> <per> [skip]"""
import csv
import sys
f, part, per = sys.argv[1], sys.argv[2], int(sys.argv[3])
skip = int(sys.argv[4]) if len(sys.argv) > 4 else 5
rows = [r for r in csv.DictReader(open(f)) if part in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
for k in range(len(rows) // per):
    d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows[k * per + skip:(k + 1) * per]]
    print(k, rows[k * per]["Kernel_Name"][:40], f"{sum(d) / len(d) / 1e3:.2f} us")
