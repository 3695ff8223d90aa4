"""One captured policy train step as a flat timeline (from a rocprofv3 kernel trace csv): start offset, duration, queue, grid, kernel.
Usage: step_trace.py <kernel_trace.csv> [which step from the end, default 2] > step.txt"""
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:64],
                     r.get("Queue_Id", ""), r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", ""))))
rows.sort()
k = int(sys.argv[2]) if len(sys.argv) > 2 else 2
starts = [i for i, r in enumerate(rows) if r[2].startswith("replay_gather")]
a, b = starts[-k - 1], starts[-k]
seg = rows[a:b]
t0 = seg[0][0]
print(f"# step span {(max(r[1] for r in seg) - t0) / 1e3:.1f} us, {len(seg)} kernels")
for s, e, n, q, g, w in seg:
    try:
        wg = int(g) // max(int(w), 1)
    except Exception:
        wg = -1
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} q{q:>3s} wg{wg:6d} {n}")
