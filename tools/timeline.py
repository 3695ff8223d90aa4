"""Per-step timeline statistics from a rocprofv3 kernel trace: span of one hipGraph replay, union of busy time, sum of kernel
time (overlap across the parallel branches), and the top kernels by time inside the step.  Usage: timeline.py <kernel_trace.csv> [steps]"""
import csv
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:60], r.get("Queue_Id", "")))
rows.sort()
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
# steps are delimited by the optimiser kernel (one mt_adamw_ema_kernel per step)
ends = [i for i, r in enumerate(rows) if r[2].startswith("mt_adamw_ema")]
if len(ends) < nsteps + 1:
    print("not enough steps in trace", len(ends)); sys.exit(1)
lo, hi = ends[-nsteps - 1], ends[-1]
seg = rows[lo + 1:hi + 1]
span = (seg[-1][1] - seg[0][0]) / nsteps
tot = sum(e - s for s, e, _, _ in seg) / nsteps
# union of busy intervals
busy, cur_s, cur_e = 0, None, None
for s, e, _, _ in seg:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
busy /= nsteps
print(f"per step: span {span/1e6:.3f} ms | GPU busy (union) {busy/1e6:.3f} ms | idle gaps {(span-busy)/1e6:.3f} ms | sum of kernel time {tot/1e6:.3f} ms | launches {len(seg)/nsteps:.0f}")
agg = defaultdict(lambda: [0, 0])
for s, e, n, _ in seg:
    agg[n][0] += e - s
    agg[n][1] += 1
for n, (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:25]:
    print(f"  {n:60s} {c/nsteps:7.1f}/step  avg {t/c/1e3:8.1f} us  {t/nsteps/1e6:7.3f} ms/step")
