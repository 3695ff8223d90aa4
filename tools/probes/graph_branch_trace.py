"""Per-queue first-start / last-end offsets of the marked replays of graph_branch_probe.py (rocprofv3 kernel trace csv)."""
import csv
import sys
from collections import defaultdict

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")))
rows.sort()
marks = [i for i, r in enumerate(rows) if "fill" in r[2].lower() or "Fill" in r[2]]
marks = marks[-5:]
for tag, a, b in zip("abcde", marks, marks[1:] + [len(rows)]):
    seg = rows[a + 1:b]
    if not seg:
        continue
    t0 = min(r[0] for r in seg)
    q = defaultdict(lambda: [1e30, 0, 0])
    for s, e, n, qi in seg:
        v = q[qi]
        v[0] = min(v[0], s - t0); v[1] = max(v[1], e - t0); v[2] += 1
    print(tag, "span %.1f us" % ((max(r[1] for r in seg) - t0) / 1e3), {k: (round(v[0] / 1e3, 1), round(v[1] / 1e3, 1), v[2]) for k, v in q.items()})
