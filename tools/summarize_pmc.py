"""Per-kernel average FETCH_SIZE / WRITE_SIZE from rocprofv3 --pmc counter_collection CSVs -> JSON on stdout.
HBM bytes follow MI355X_MICROARCH.md section HBM: counters are in KiB; on gfx950 FETCH_SIZE reports exactly half of the bytes of a
wide coalesced streaming read (128-B requests tallied at 64 B) => fetched bytes = 2 * FETCH_SIZE * 1024 for such kernels."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]
out = {}
leg_totals = {}
FROM_SUMMARY = None
if os.path.isfile(root):          # re-derive roofline_traffic from a pmc_summary.json written earlier (the raw csv files do not travel)
    FROM_SUMMARY = json.load(open(root))


def _units(leg):
    """Steps (policy) / UNet forwards (sampler) the PMC command of this leg ran, from the JSON line in its log."""
    for ctr in ("fetch", "write"):
        try:
            lines = [l for l in open(os.path.join(root, f"pmc_{ctr}_{leg}.log")) if l.startswith('{"metric')]
            j = json.loads(lines[-1])
            return j["unet_forwards"] if "unet_forwards" in j else j["steps"] + j["warmup"]
        except Exception:
            continue
    return None


SETUP_NAMES = ("__amd_rocclr_", "at::native::", "pack_weight_kernel", "pack_weight_h_kernel")


def _steady(kernels, units):
    """HBM bytes per unit of the REPEATING part of a leg: a profiled run also holds the model set-up (parameter uploads, RNG fills,
    one-off single-tensor packs, the first build of the pack tables), which `hbm_bytes_per_unit` = whole run / units charges to the
    steps.  A kernel counts with floor(launches / units) launches per unit; kernels that only set up (by name) count zero."""
    tot = 0.0
    for k, v in kernels.items():
        if any(k.startswith(n) for n in SETUP_NAMES):
            continue
        tot += v.get("hbm_bytes_per_launch_corrected", 0.0) * (v.get("n", 0) // units)
    return tot


for leg in ("policy", "policy_bf16", "video", "video_bf16"):
    if FROM_SUMMARY is not None:
        if leg not in FROM_SUMMARY:
            out[leg] = {}
            continue
        out[leg] = FROM_SUMMARY[leg]
        if leg in FROM_SUMMARY.get("roofline_traffic", {}).get("legs", {}):
            leg_totals[leg] = dict(FROM_SUMMARY["roofline_traffic"]["legs"][leg])
            if "steady_hbm_bytes_per_unit" not in leg_totals[leg]:       # from the 40 largest kernels the summary kept
                leg_totals[leg]["steady_hbm_bytes_per_unit"] = _steady(out[leg], leg_totals[leg]["units_in_run"])
        continue
    res = defaultdict(lambda: {"n": 0})
    tot = {"fetch": 0.0, "write": 0.0}
    for ctr in ("fetch", "write"):
        files = glob.glob(os.path.join(root, f"pmc_{ctr}_{leg}", "**", "*counter_collection.csv"), recursive=True)
        acc = defaultdict(lambda: [0.0, 0])
        for f in files:
            for r in csv.DictReader(open(f)):
                name = r.get("Kernel_Name", "")
                short = name.split("(")[0].replace("void ", "")
                val = float(r.get("Counter_Value", 0) or 0)
                a = acc[short]
                a[0] += val
                a[1] += 1
                tot[ctr] += val
        for k, (s, n) in acc.items():
            res[k][f"{ctr}_kib_avg"] = s / max(n, 1)
            res[k]["n"] = max(res[k]["n"], n)
    for k, v in res.items():
        f, w = v.get("fetch_kib_avg", 0.0), v.get("write_kib_avg", 0.0)
        v["hbm_bytes_per_launch_corrected"] = (2.0 * f + w) * 1024.0
        v["hbm_bytes_per_launch_raw"] = (f + w) * 1024.0
    out[leg] = dict(sorted(res.items(), key=lambda kv: -kv[1].get("hbm_bytes_per_launch_corrected", 0) * kv[1]["n"])[:40])
    u = _units(leg)
    if u:       # whole-leg HBM bytes per step / per UNet forward: every kernel of the run, same correction
        leg_totals[leg] = {"units_in_run": u, "hbm_bytes_per_unit": (2.0 * tot["fetch"] + tot["write"]) * 1024.0 / u,
                           "fetch_bytes_per_unit": 2.0 * tot["fetch"] * 1024.0 / u, "write_bytes_per_unit": tot["write"] * 1024.0 / u,
                           "steady_hbm_bytes_per_unit": _steady(res, u)}
# bench.py reads profiles/roofline_traffic.json: {"policy"|"video": {"<kernel><BM,BN>": corrected HBM bytes per launch}}
import re
rt = {}
for leg_name in ("policy", "policy_bf16", "video", "video_bf16"):
    traffic = {}
    for k, v in out[leg_name].items():
        m = re.match(r"(conv_(?:igemm|wgrad)_(?:dma_f32|f32x3|f32p|f32|bf16))<(\d+), (\d+)", k)
        mx = re.match(r"conv_halo_x3<(\d+)>", k)                  # round 4: three-plane halo conv, key = bench.py's ops.last_kernel name
        if mx:
            a = traffic.setdefault(f"conv_halo_x3<{mx.group(1)}>", [0.0, 0])
            a[0] += v["hbm_bytes_per_launch_corrected"] * v["n"]
            a[1] += v["n"]
            continue
        mh = re.match(r"conv_igemm_h<(\d+), (\d+), (float|unsigned short|f16s)(?:, \d+)?>", k)
        # <WAVES_M, WAVES_N, TM, TN, SB[, GN][, F16]> -> BM x BN (the trailing bool is the fp16 flag of round 3)
        m3 = re.match(r"(conv_halo_h3|conv_igemm_h2)<(\d+), (\d+), (\d+), (\d+), \d+(?:, (\d+))?(?:, (?:true|false))?>", k)
        mf = re.match(r"conv_frames_h3<(\d+)(?:, (?:true|false))?>", k)
        mw = re.match(r"(conv_wgrad_multi(?:_halo|_x3h|_x3)?)_kernel", k)
        if mw:
            a = traffic.setdefault(mw.group(1), [0.0, 0])
            a[0] += v["hbm_bytes_per_launch_corrected"] * v["n"]
            a[1] += v["n"]
            continue
        if m or mh or m3 or mf:
            if m3:
                wm_, wn_, tm_, tn_ = (int(m3.group(i)) for i in (2, 3, 4, 5))
                key = f"{m3.group(1)}{'_gn' if m3.group(6) == '1' else ''}<{wm_ * tm_ * 32}x{wn_ * tn_ * 32}>"
            elif mf:
                key = f"conv_frames_h3<{int(mf.group(1)) * 64}x128>"
            else:
                key = (f"{m.group(1)}<{m.group(2)},{m.group(3)}>" if m else
                       f"conv_igemm_h<{mh.group(1)},{mh.group(2)},{'float' if mh.group(3) == 'float' else ('fp16' if mh.group(3) == 'f16s' else 'bf16')}>")
            a = traffic.setdefault(key, [0.0, 0])     # template variants sharing a tile: launch-count weighted mean
            a[0] += v["hbm_bytes_per_launch_corrected"] * v["n"]
            a[1] += v["n"]
    hx = [(k, a) for k, a in traffic.items() if k.startswith("conv_halo_x3<")]       # + the family key bench.py's roofline uses
    if hx:
        traffic["conv_halo_x3"] = [sum(a[0] for _, a in hx), sum(a[1] for _, a in hx)]
    rt[leg_name] = {k: a[0] / max(a[1], 1) for k, a in traffic.items()}
rt["legs"] = leg_totals
out["roofline_traffic"] = rt
print(json.dumps(out, indent=1))
