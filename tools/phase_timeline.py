"""Wall-clock phases of one captured policy train step from a rocprofv3 kernel trace (csv).  Phases are delimited by marker kernels:
replay_gather (step start) | sincos_embed (ConditionalUnet1D forward starts) | mse_loss | first spatial_softmax_bwd (image-encoder
backward starts) | mt_sumsq (optimiser) | end of the step.  Usage: phase_timeline.py <kernel_trace.csv> [steps]"""
import csv
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:48]))
rows.sort()
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
starts = [i for i, r in enumerate(rows) if r[2].startswith("replay_gather")]
starts = starts[-nsteps - 1:]
MARK = [("enc_fwd", "replay_gather"), ("unet_fwd", "sincos_embed"), ("unet_bwd", "mse_loss"), ("enc_bwd(+unet wgrad)", "spatial_softmax_bwd"),
        ("optimiser+pack", "mt_sumsq")]
acc = defaultdict(lambda: [0.0, 0.0, 0, 0.0])
tops = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for a, b in zip(starts[:-1], starts[1:]):
    seg = rows[a:b]
    cut = []
    for name, mk in MARK:
        idx = next(i for i, r in enumerate(seg) if r[2].startswith(mk))
        cut.append((name, idx))
    cut.append(("end", len(seg)))
    for (name, i0), (_, i1) in zip(cut[:-1], cut[1:]):
        part = seg[i0:i1]
        t0 = part[0][0]
        t1 = seg[i1][0] if i1 < len(seg) else max(r[1] for r in part)
        busy, ce = 0, None
        cs = None
        for s, e, _ in part:
            if ce is None or s > ce:
                if ce is not None:
                    busy += ce - cs
                cs, ce = s, e
            else:
                ce = max(ce, e)
        busy += ce - cs
        v = acc[name]
        v[0] += t1 - t0; v[1] += sum(e - s for s, e, _ in part); v[2] += len(part); v[3] += busy
        for s, e, n in part:
            tops[name][n][0] += e - s; tops[name][n][1] += 1
n = len(starts) - 1
tot = 0.0
for name, _ in MARK:
    w, k, c, b = acc[name]
    tot += w / n
    print(f"{name:24s} wall {w/n/1e6:7.3f} ms | kernel-time sum {k/n/1e6:7.3f} ms | busy {b/n/1e6:7.3f} ms | launches {c/n:6.1f}")
    for kn, (t, cnt) in sorted(tops[name].items(), key=lambda kv: -kv[1][0])[:7]:
        print(f"      {kn:48s} {cnt/n:6.1f}/step avg {t/cnt/1e3:7.1f} us  {t/n/1e6:6.3f} ms")
print(f"step wall {tot:.3f} ms")
