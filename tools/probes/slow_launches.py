"""List the slowest launches of selected kernels in a rocprofv3 kernel trace, with their position in the step and grid size.
usage: slow_launches.py <kernel_trace.csv> <name substring> [...]"""
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
pats = sys.argv[2:]
# step boundaries: mt_adamw_ema_kernel ends a step
step, pos = 0, 0
out = []
for r in rows:
    name = r["Kernel_Name"]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if any(p in name for p in pats):
        out.append((step, pos, d, name[:60], r.get("Grid_Size", r.get("Grid_Size_X", "?")), r.get("Workgroup_Size", r.get("Workgroup_Size_X", "?")),
                    r.get("LDS_Block_Size", "?")))
    pos += 1
    if "mt_adamw_ema" in name:
        step += 1
        pos = 0
last = max(s for s, *_ in out)
for s, p, d, n, g, w, l in out:
    if s == last - 1 and d > 40:
        print(f"step {s} pos {p:4d} {d:8.1f} us grid {g} wg {w} lds {l}  {n}")
