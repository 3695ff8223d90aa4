#!/bin/bash
# kernel trace of two eager UNet forwards of the bf16 sampler, summarised per (kernel, grid): launches per forward, average us, ms per forward
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/vtrace; rm -rf $OUT; mkdir -p $OUT
V2A_SAMPLER_GRAPH=0 rocprofv3 --kernel-trace --output-format csv -d $OUT/a -o v -- python $R/tools/video_only.py --storage ${STORAGE:-bf16} --steps 3 > $OUT/a.log 2>&1
cd $R
python - <<PY
import csv, glob, collections
f = glob.glob("$OUT/a/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
acc = collections.defaultdict(list)
for r in rows:
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    g = int(r["Grid_Size"]) if "Grid_Size" in r else int(r["Grid_Size_X"])
    acc[(k, g, int(r.get("Workgroup_Size", r.get("Workgroup_Size_X", 0))))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
nf = 8.0   # video_only: 1 warm-up + 1 timed call x (3 steps + 1)  -> forwards
import re
nf = max(1, sum(len(v) for (k, g, w), v in acc.items() if "video_denoise" in k))
tot = sum(sum(v) for v in acc.values())
print(f"forwards {nf}, kernel time per forward {tot/nf/1e3:.2f} ms")
for (k, g, w), v in sorted(acc.items(), key=lambda kv: -sum(kv[1]))[:45]:
    print(f"{k[:70]:70s} grid {g//max(w,1):7d}x{w:4d} n/fwd {len(v)/nf:6.1f} avg {sum(v)/len(v):8.1f} us  ms/fwd {sum(v)/nf/1e3:7.3f}")
PY
python - <<PY
import csv, glob
f = glob.glob("$OUT/a/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# launches of the LAST forward: everything after the second-to-last video_denoise launch
idx = [i for i, r in enumerate(rows) if "video_denoise" in r["Kernel_Name"]]
lo = idx[-2] + 1 if len(idx) >= 2 else 0
print("--- last forward, conv launches in order (us)")
for r in rows[lo:idx[-1]]:
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if k.startswith("conv_"):
        g = int(r.get("Grid_Size", r.get("Grid_Size_X", 0))) // max(int(r.get("Workgroup_Size", r.get("Workgroup_Size_X", 1))), 1)
        print(f"{k[:44]:44s} grid {g:6d} {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:8.1f}")
PY
rm -rf $OUT/a
