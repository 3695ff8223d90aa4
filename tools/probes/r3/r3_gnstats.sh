#!/bin/bash
# per-kernel time of the policy step's GroupNorm launches (kernel trace of the replayed graph; rocprofv3 serialises the queues)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/gnstats -o p -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-video --no-bf16-extra --no-predict --no-roofline-pass ${EXTRA:-} > $R/gpurun_out/gnstats.log 2>&1
python - <<PY
import csv, glob
f = glob.glob("$R/gpurun_out/gnstats/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows) / 13 / 1e3
gn = 0
for r in rows:
    nm = r["Name"].replace("void ", "")
    if nm.startswith("gn_"):
        t = float(r["TotalDurationNs"]) / 13 / 1e3
        gn += t
        print(f"{nm[:46]:46s} n/step {int(r['Calls'])/13:5.1f} avg {float(r['AverageNs'])/1e3:7.1f} us  us/step {t:7.1f}")
print("GN total us/step", round(gn, 1), "of", round(tot, 1))
PY
find $R/gpurun_out/gnstats -name "*kernel_trace.csv" -delete
