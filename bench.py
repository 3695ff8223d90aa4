#!/usr/bin/env python
"""Headline benchmark (driver contract: one JSON line on rank 0).

    python bench.py --gpus N --steps K --warmup W

Workload at N=1 = BASELINE.json configs[1]: Libero 8-task diffusion-policy train step, batch 64 per GPU, synthetic 128x128
start/goal images (uint8 replay store, 8 tasks x 50 episodes x 121 frames) + 7-DoF action chunks (horizon 16), random-init weights
of the reference architecture (87,219,143 parameters).  A "step" = SURVEY.md 8a rows R1..R9: bit-exact replay index draw + HBM
gather -> noise / timesteps -> forward + backward (hand-written HIP) -> [RCCL grad all-reduce] -> fused clip + AdamW + zero + EMA.
Compute dtype fp32 (exact-f32 MFMA): the parity configuration (north_star: outputs within 1e-4 of the fp32 CPU path).
Multi-GPU: weak scaling, one process per GPU (torch.distributed 'nccl' = RCCL over xGMI), value = batch-64 steps of ALL ranks / time.
Extra objects: `roofline` (dominant kernel, measured live with HIP events in an instrumented pass), `cpu_baseline` (the CPU oracle
timed on this box's host cores on a bounded sample), `video` (sampler frames/s, when --video).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "video-to-action-release_amd"))

FP32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md chip table (dense f32 matrix = vector peak)
HBM_PEAK_GBS = 8000.0
BF16_MFMA_PEAK_TFLOPS = 2500.0     # dense bf16 MFMA


def _traffic(leg, kernel):
    """HBM bytes per launch of `kernel` from the committed PMC summary (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes,
    FETCH doubled per MI355X_MICROARCH.md section HBM; tools/profile_round.sh + tools/summarize_pmc.py), or None."""
    tpath = os.path.join(ROOT, "profiles", "roofline_traffic.json")
    try:
        return json.load(open(tpath)).get(leg, {}).get(kernel)
    except Exception:
        return None


# SURVEY.md 8d algorithmic bytes: sampler per UNet forward at B = 16 = W + 16 * A * s (W = 201,087,649 weights, A = 1.816 G activation
# elements per sample, s bytes per element); policy per step at B = 64, fp32 = 64 B * 87,219,143 parameters + 3 * 3.326 M * 64 * 4
TRAFFIC_SOURCE = ("profiles/roofline_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of a separate run (tools/profile_round.sh), "
                  "NOT measured inside this bench run")
ALGORITHMIC_BYTES = {"video": 201087649 * 4 + 16 * 1.816e9 * 4, "video_bf16": 201087649 * 2 + 16 * 1.816e9 * 2, "policy": 8.13e9,
                     "policy_bf16": 8.13e9}          # the bf16-MFMA policy mode keeps fp32 tensors in HBM: same algorithmic bytes


def _leg_traffic(leg, batch_scale=1.0):
    """Whole-leg HBM bytes per step / UNet forward (every kernel of the leg, PMC) against SURVEY 8d's algorithmic bytes, or None."""
    tpath = os.path.join(ROOT, "profiles", "roofline_traffic.json")
    try:
        t = json.load(open(tpath)).get("legs", {}).get(leg)
        alg = ALGORITHMIC_BYTES[leg] * batch_scale
        # `steady` leaves out the set-up launches of the profiled run (parameter uploads, RNG fills, one-off packs: tools/summarize_pmc.py)
        steady = t.get("steady_hbm_bytes_per_unit", t["hbm_bytes_per_unit"])
        return {"hbm_bytes": steady, "algorithmic_bytes": alg, "ratio": steady / alg, "hbm_bytes_incl_setup_of_the_profiled_run": t["hbm_bytes_per_unit"],
                "per": "UNet forward (B=16)" if leg.startswith("video") else "train step (B=64)", "source": "profiles/roofline_traffic.json"}
    except Exception:
        return None


# ---------------------------------------------------------------------------------------------------------------- output line
# The driver keeps the last 8 KB of stdout: the FULL record (every note, every conv variant) goes to bench_full_last.json + stderr, the
# line on stdout is the same record with prose dropped, floats rounded to 5 significant digits and long strings cut -- so that the
# bf16 leg, the sampler legs, predict_action and every cpu_baseline survive in the driver's copy.
_DROP = {"note", "timing", "isolation", "traffic_source", "source", "settings", "all_conv_variants", "families", "hbm_bytes_incl_setup_of_the_profiled_run",
         "per", "cpu_model_detail", "image", "action"}
_KEEP_STR = 110


def _round5(x):
    if isinstance(x, float):
        if x != x or x in (float("inf"), float("-inf")):
            return None
        return float(f"{x:.5g}")
    return x


def _compact(o, depth=0):
    if isinstance(o, dict):
        out = {}
        for k, v in o.items():
            if k in _DROP:
                continue
            out[k] = _compact(v, depth + 1)
        return out
    if isinstance(o, (list, tuple)):
        return [_compact(v, depth + 1) for v in o][:12]
    if isinstance(o, str):
        return o if len(o) <= _KEEP_STR else o[:_KEEP_STR - 1] + "~"
    return _round5(o)


def _emit(out):
    full = json.dumps(out)
    try:
        with open(os.path.join(ROOT, "bench_full_last.json"), "w") as f:
            f.write(full + "\n")
    except Exception:
        pass
    print("[bench full record] " + full, file=sys.stderr)
    line = _compact(out)
    if isinstance(out.get("roofline"), dict) and "families" in out["roofline"]:      # the three largest source-kernel families, by in-graph time
        fams = out["roofline"]["families"]
        top = sorted(fams.items(), key=lambda kv: -(kv[1].get("ms_per_step_in_graph") or 0.0))[:3]
        line["roofline"]["top_families"] = {k: _compact(v) for k, v in top}
    for leg in ("policy_exact", "video_exact"):  # the strict-arithmetic legs: the numbers only (everything else is in the full record)
        if isinstance(line.get(leg), dict):
            line[leg] = {k: v for k, v in line[leg].items() if k in ("value", "unit", "ms_per_step", "seconds_per_sample_call", "f32_conv_mode",
                                                                     "whole_step_frac_of_f32_mfma_floor", "algorithmic_tflops", "error")}
    txt = json.dumps(line, separators=(",", ":"))
    if len(txt) > 7600:                          # last resort: shed the per-leg configs, then the per-leg rooflines of secondary legs
        for leg in list(line):
            if isinstance(line[leg], dict) and leg not in ("config", "roofline", "cpu_baseline"):
                line[leg].pop("config", None)
        txt = json.dumps(line, separators=(",", ":"))
    if len(txt) > 7600:
        for leg in ("video_bf16_ddpm100", "video_c5", "video_c5_fp16", "video_fp16", "video_round8"):
            if isinstance(line.get(leg), dict):
                line[leg] = {k: v for k, v in line[leg].items() if k in ("value", "unit", "seconds_per_sample_call", "algorithmic_tflops", "error")}
        txt = json.dumps(line, separators=(",", ":"))
    if len(txt) > 7600:                          # still too long: explanatory strings and per-call lists of every leg (all of it is in the full record)
        def shed(o):
            if isinstance(o, dict):
                return {k: shed(v) for k, v in o.items() if k not in ("peak_is", "output_range", "call_seconds", "cpu_model", "usable_cores",
                                                                      "estimate_assumes", "frac_of_f32_mfma_peak", "end_to_end_frac_of_f32_mfma_peak")}
            return o
        line = shed(line)
        txt = json.dumps(line, separators=(",", ":"))
    # the driver keeps the LAST 8 KB of stdout: a longer line would arrive cut.  Whole secondary legs go (least important first, named in
    # `dropped_for_length`; the full record on stderr / bench_full_last.json keeps them) until the line fits with room to spare.
    dropped = []
    for leg in ("video_train", "video_c5_fp16", "video_c5", "video_bf16_ddpm100", "video_fp16", "video_exact", "policy_exact", "video_round8",
                "dp_structure", "policy_b256", "video_b1"):
        if len(txt) <= 7900:
            break
        if leg in line:
            dropped.append(leg)
            del line[leg]
            line["dropped_for_length"] = dropped
            txt = json.dumps(line, separators=(",", ":"))
    print(txt)


def _family(name):
    """Source-kernel family of a kernel name: 'void conv_igemm_f32x3<64, 64, 2, 2, 2, false>(ConvDescH)' and the launcher's plan name
    'conv_igemm_f32x3<64,64>' -> 'conv_igemm_f32x3' (every instance of one template is one kernel: same source, same roofline)."""
    n = name.replace("void ", "").strip()
    for sep in ("<", "("):
        i = n.find(sep)
        if i > 0:
            n = n[:i]
    return n[:-len("_kernel")] if n.endswith("_kernel") else n


def _live_kernel_stats(precision, batch, steps=10, warmup=3, timeout=420):
    """Per-family launch statistics of the CAPTURED policy step, measured in THIS bench run: a child `rocprofv3 --kernel-trace --stats` of
    this script (policy leg only, no secondary legs) in a scratch directory; returns ({family: [calls, total_ns]}, csv_path) or (None, why).
    The same command as tools/profile_round.sh's first line, whose summary is committed under profiles/ per round."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    tmp = tempfile.mkdtemp(prefix="v2a_bench_prof_", dir="/tmp")
    cmd = [exe, "--kernel-trace", "--stats", "--output-format", "csv", "-d", tmp, "-o", "policy", "--", sys.executable, os.path.abspath(__file__),
           "--steps", str(steps), "--warmup", str(warmup), "--batch", str(batch), "--precision", precision, "--no-cpu-baseline", "--no-video",
           "--no-bf16-extra", "--no-predict", "--no-roofline-pass", "--no-video-train"]
    try:
        r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=timeout)
        files = glob.glob(os.path.join(tmp, "**", "*kernel_stats.csv"), recursive=True)
        if r.returncode != 0 or not files:
            return None, f"rocprofv3 child rc {r.returncode}: {r.stderr[-300:]}"
        fam = {}
        for row in csv.DictReader(open(files[0])):
            d = fam.setdefault(_family(row["Name"]), [0, 0.0])
            d[0] += int(row["Calls"])
            d[1] += float(row["TotalDurationNs"])
        keep = os.path.join(ROOT, "gpurun_out", "bench_live_policy_kernel_stats.csv")      # scratch copy for the round's profiles/ commit
        try:
            os.makedirs(os.path.dirname(keep), exist_ok=True)
            shutil.copyfile(files[0], keep)
        except Exception:
            keep = files[0]
        return fam, keep
    except Exception as e:
        return None, f"{type(e).__name__}: {e}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def _graph_avg_us(kernel_prefix, csv_name="r05_policy_kernel_stats.csv"):
    """Average duration (us) of the launches whose name starts with kernel_prefix in the committed rocprofv3 --kernel-trace --stats summary
    of the captured step (profiles/): the figure INSIDE the replayed graph, next to the eager-pass figure this run measures."""
    import csv
    try:
        tot, n = 0.0, 0
        key = kernel_prefix.replace(" ", "")
        for r in csv.DictReader(open(os.path.join(ROOT, "profiles", csv_name))):
            name = r["Name"].replace("void ", "").replace(" ", "")
            if name.startswith(key):
                tot += float(r["TotalDurationNs"])
                n += int(r["Calls"])
        return (tot / n / 1e3) if n else None
    except Exception:
        return None


def build_store(torch, device, batch, seed):
    from v2a_hip.replay import ReplayStore
    n_eps, ep_len = 8 * 50, 121
    store = ReplayStore(1200, 700, 30, image_hw=(128, 128), capacity_frames=n_eps * ep_len, device=device)
    g = torch.Generator(device=device).manual_seed(seed)
    # synthetic payload generated directly in HBM (uint8 frames like the HDF5 random-action file; actions U[-1,1))
    store.frames.copy_(torch.randint(0, 256, store.frames.shape, dtype=torch.uint8, device=device, generator=g))
    store.acts.copy_(torch.rand(store.acts.shape, device=device, generator=g) * 2 - 1)
    for e in range(n_eps):
        store.episodes.append((e * ep_len, ep_len, f"task{e % 8}", "agentview", e % 8))
    return store


def instrumented_pass(torch, trainer, steps):
    """Eager pass with a HIP-event pair around every conv launch: per tile-variant totals of algorithmic FLOPs and time."""
    from v2a_hip import ops
    from v2a_hip._lib import lib
    recs = []
    orig_fwd, orig_wg = ops.conv2d, ops.conv2d_wgrad
    # A ~40 us blocker kernel is queued in front of every measured launch so that the event pair and the kernel are already in the
    # queue when the GPU reaches them (otherwise the host's submission latency would be counted as kernel time for 10-us kernels).
    blk_a = torch.zeros(32 << 20, dtype=torch.float32, device=trainer.device)
    blk_b = torch.zeros_like(blk_a)

    def timed(kind, fn, flops_of):
        def wrapper(*a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            orig_axpy(blk_a, blk_b, 1.0, out=blk_b)
            e0.record()
            out = fn(*a, **k)
            e1.record()
            recs.append((kind, flops_of(a, k, out), e0, e1))
            return out
        return wrapper

    def f_fwd(a, k, out):
        x, cout, kh, kw = a[0], a[3], a[4], a[5]
        c2 = k.get("x2").shape[-1] if k.get("x2") is not None else 0
        y = out if k.get("y") is None else k["y"]
        y = y[0] if isinstance(y, tuple) else y          # conv2d(..., defer=True) returns (y, slabs)
        m = y.shape[0] * y.shape[1] * y.shape[2]
        K = kh * kw * (x.shape[-1] + c2)
        # a conv over a zero-interleaved input (idil = 2: data gradient of a stride-2 conv, transposed conv) has 1 / 4 (2-d) or 1 / 2 (1-d) of
        # its tap x pixel products on stored pixels: only those are algorithmic work (the kernels skip the rest since round 6)
        live = 1.0 if int(k.get("idil", 1) or 1) != 2 else (0.25 if y.shape[1] > 1 else 0.5)
        return (ops.last_kernel[0], 2.0 * m * cout * K * live, (m, cout, K, kh, kw, int(k.get("bmode", 0) or 0)))     # name: the launcher's plan

    def f_wg(a, k, out):
        x, dy, kh, kw = a[0], a[1], a[3], a[4]
        c2 = k.get("x2").shape[-1] if k.get("x2") is not None else 0
        m = dy.shape[0] * dy.shape[1] * dy.shape[2]
        cout = dy.shape[-1]
        K = kh * kw * (x.shape[-1] + c2)
        return (ops.last_kernel[0], 2.0 * m * cout * K, (m, cout, K, kh, kw, -1))

    orig_axpy = ops.axpy
    ops.conv2d = timed("fwd", orig_fwd, f_fwd)
    ops.conv2d_wgrad = timed("wgrad", orig_wg, f_wg)
    # grouped weight gradients (ops.WgradBatch -> conv_wgrad_multi_kernel / conv_wgrad_multi_halo_kernel): one event pair per launch family
    orig_fam = ops.WgradBatch._launch_family

    def fam(self, calls, target_wg):
        fl = 0.0
        for c in calls:
            x, dy = c["x"], c["dy"]
            c2 = c["x2"].shape[-1] if c["x2"] is not None else 0
            fl += 2.0 * dy.shape[0] * dy.shape[1] * dy.shape[2] * dy.shape[3] * c["KH"] * c["KW"] * (x.shape[-1] + c2)
        v0 = calls[0]["variant"]
        name = "conv_wgrad_multi_x3h" if v0 >= 8 else ("conv_wgrad_multi_x3" if v0 >= 6 else ("conv_wgrad_multi_halo" if v0 >= 3 else "conv_wgrad_multi"))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        orig_axpy(blk_a, blk_b, 1.0, out=blk_b)
        e0.record()
        orig_fam(self, calls, target_wg)
        e1.record()
        recs.append(("wgrad", (name, fl, (len(calls), 0, 0, 0, 0, -2)), e0, e1))

    ops.WgradBatch._launch_family = fam
    g_saved, e_saved = trainer.use_graph, trainer.eng.enc_streams
    trainer.use_graph = False
    trainer.eng.enc_streams = False          # measure every kernel alone on the main stream (the timed step overlaps the two camera encoders)
    d_saved, trainer.eng.defer_unet_wgrad = trainer.eng.defer_unet_wgrad, False
    try:
        for _ in range(steps):
            trainer.step()
        torch.cuda.synchronize()
    finally:
        ops.conv2d, ops.conv2d_wgrad = orig_fwd, orig_wg
        ops.WgradBatch._launch_family = orig_fam
        trainer.use_graph = g_saved
        trainer.eng.enc_streams = e_saved
        trainer.eng.defer_unet_wgrad = d_saved
    agg, shapes = {}, {}
    for kind, (name, fl, shape), e0, e1 in recs:
        dt = e0.elapsed_time(e1) * 1e-3
        for tab, key in ((agg, name), (shapes, (name,) + shape)):
            d = tab.setdefault(key, [0.0, 0.0, 0])
            d[0] += fl
            d[1] += dt
            d[2] += 1
    if os.environ.get("V2A_BENCH_SHAPES"):       # per-shape table for kernel tuning: (kernel, M, Cout, K, kh, kw, bmode|-1 = wgrad)
        rows = sorted(shapes.items(), key=lambda kv: -kv[1][1])
        with open(os.environ["V2A_BENCH_SHAPES"], "w") as f:
            for key, (fl, t, n) in rows:
                f.write(f"{key[0]:28s} M={key[1]:7d} Cout={key[2]:5d} K={key[3]:6d} k={key[4]}x{key[5]} mode={key[6]:2d} n/step={n / steps:5.1f} "
                        f"us={t / n * 1e6:8.1f} TF={fl / t / 1e12:6.1f} ms/step={t / steps * 1e3:6.3f}\n")
    return agg


def _cpu_baseline_worker(batch, threads, budget, max_steps=12):
    """Runs in a subprocess (bounded by a hard timeout): the CPU oracle train step on `threads` host threads."""
    import torch
    from oracle import policy as OP
    from oracle import optim as OO
    from diffuser.diffusion_policy.get_dp import build_policy, DEFAULT_CONF
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    pol = build_policy(DEFAULT_CONF)
    sd = {k: v.detach().clone() for k, v in pol.state_dict().items()}
    names = [n for n, p in pol.named_parameters() if p.dim() > 0 and p.numel() > 0]
    ps = [sd[n] for n in names]
    ms = [torch.zeros_like(p) for p in ps]
    vs = [torch.zeros_like(p) for p in ps]
    em = [p.clone() for p in ps]
    st = OO.EmaState(power=0.75)
    b = {"obs": {"img_obs_1": torch.rand(batch, 1, 3, 128, 128), "img_goal_1": torch.rand(batch, 1, 3, 128, 128)},
         "action": torch.rand(batch, 16, 7) * 2 - 1}
    n, t_used = 0, 0.0
    while t_used < budget and n < max_steps:
        t0 = time.time()
        noise = torch.randn(batch, 16, 7)
        ts = torch.randint(0, 100, (batch,))
        _, g = OP.loss_and_grads(sd, b, noise, ts, names=names)
        OO.train_tail(ps, [g[k] for k in names], ms, vs, em, n + 1, st)
        dt = time.time() - t0
        if n > 0:            # the first iteration warms the allocator / oneDNN primitives and is not counted
            t_used += dt
        else:
            first = dt
        n += 1
        print(json.dumps({"done": max(n - 1, 0), "t": t_used, "first": first}), flush=True)


def _cpu_video_worker(threads, budget, max_fwd=4, count_first=False):
    """Subprocess body: full-size Unet_Libero forward of the CPU oracle at B=1 (one denoise step of one sample).
    count_first: the first (cold) forward is the sample (one-thread setting, where a warm-up pass would double a minute of work)."""
    import torch
    from oracle.video_unet import unet_libero_forward, LIBERO_CFG
    from flowdiffusion.flowdiffusion.unet import Unet_Libero
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    sd = {k: v.detach() for k, v in Unet_Libero().state_dict().items()}
    x, t, te = torch.randn(1, 24, 128, 128), torch.tensor([50]), torch.randn(1, 10, 512)
    n, t_used = 0, 0.0
    with torch.no_grad():
        while t_used < budget and n < max_fwd:
            t0 = time.time()
            unet_libero_forward(sd, x, t, te, LIBERO_CFG)
            dt = time.time() - t0
            if n > 0 or count_first:
                t_used += dt
            n += 1
            print(json.dumps({"done": n if count_first else max(n - 1, 0), "t": t_used}), flush=True)


def _usable_cores():
    try:
        return len(os.sched_getaffinity(0))
    except Exception:
        return os.cpu_count() or 1


def _cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.lower().startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except Exception:
        pass
    import platform
    return platform.processor() or platform.machine()


def _cpu_spawn(worker_call, threads):
    """Start one CPU-oracle timing subprocess on `threads` host threads (GPU hidden from it)."""
    import subprocess
    code = (f"import sys, json, time; sys.path.insert(0, {ROOT!r}); sys.path.insert(0, {os.path.join(ROOT, 'video-to-action-release_amd')!r}); "
            f"import bench; bench.{worker_call}")
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads), HIP_VISIBLE_DEVICES="")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, env=env)


def _cpu_collect(proc, hard_timeout):
    """Last progress line a timing subprocess printed within `hard_timeout` seconds (it is killed afterwards), or None."""
    import subprocess
    try:
        out, _ = proc.communicate(timeout=hard_timeout)
    except subprocess.TimeoutExpired:
        proc.kill()
        out, _ = proc.communicate()
    last = None
    for ln in (out or "").strip().splitlines():
        try:
            last = json.loads(ln)
        except Exception:
            pass
    return last


class CpuBaselines:
    """The CPU oracle (pinned bit-exact against the reference: tests/test_oracle_golden.py) timed on this box's host cores, as
    SURVEY 8d asks: ONE thread (the reference's launch script pins OMP_NUM_THREADS=1, train_libero_dp.sh:11) and 32 threads (all
    256 SMT threads of this host do not finish one step in 100 s -- measured in round 2 -- so that setting is not run).  Every run
    is a subprocess with a hard timeout.  `run_all()` BLOCKS and is called before any GPU leg is timed (and after the GPU has gone
    idle): no CPU-oracle process is alive inside a GPU timed window.  The two one-thread runs share the wait with the 32-thread runs
    (34 of this host's cores busy at once)."""

    def __init__(self, batch, video_steps, video=True):
        self.batch, self.video_steps, self.video_on = batch, video_steps, video
        self.usable = _usable_cores()
        self.model = _cpu_model()
        self.results = {}

    def run_all(self):
        th = min(32, self.usable)
        t0 = time.time()
        bg = {"policy": _cpu_spawn("_cpu_baseline_worker(8, 1, 25.0, 4)", 1)}
        if self.video_on:
            bg["video"] = _cpu_spawn("_cpu_video_worker(1, 1.0, 1, True)", 1)
        pol = [self._policy_multi(th)]
        vid = [self._video_multi(th)] if self.video_on else []
        last = _cpu_collect(bg["policy"], max(5.0, 90.0 - (time.time() - t0)))
        if last and last["done"] >= 1:
            per8 = last["t"] / last["done"]
            pol.insert(0, {"threads": 1, "value": 1.0 / (per8 * self.batch / 8.0),
                           "sample": f"{last['done']} timed B=8 steps (BASELINE configs[0] size; {per8:.2f} s each, first untimed), scaled "
                                     f"linearly to B={self.batch} rows per step"})
        else:
            pol.insert(0, {"threads": 1, "value": None, "sample": "no B=8 step finished on one thread within the cap"})
        self.results["policy"] = self._pack(pol, "steps/s", f"CPU oracle of the same train step (fwd+bwd via torch-CPU autograd + "
                                                            f"clip/AdamW/EMA), fp32, B={self.batch}")
        if self.video_on:
            last = _cpu_collect(bg["video"], max(5.0, 150.0 - (time.time() - t0)))
            if last and last["done"] >= 1:
                per = last["t"] / last["done"]
                vid.insert(0, {"threads": 1, "value": 7.0 / (per * self.video_steps),
                               "sample": f"ONE full-size Unet_Libero forward at B=1, cold (no warm-up pass: {per:.1f} s)"})
            else:
                vid.insert(0, {"threads": 1, "value": None, "sample": "the one-thread forward did not finish within 150 s"})
            self.results["video"] = self._pack(vid, "predicted frames/s",
                                               f"CPU oracle UNet forwards, extrapolated linearly to {self.video_steps} denoise steps per sample "
                                               f"(the sampler costs B x steps forwards; the elementwise DDIM update is ignored)")
        self.results["wall_s"] = time.time() - t0

    def _policy_multi(self, th):
        last = _cpu_collect(_cpu_spawn(f"_cpu_baseline_worker({self.batch}, {th}, 12.0, 12)", th), 120.0)
        if last and last["done"] >= 1:
            return {"threads": th, "value": last["done"] / last["t"],
                    "sample": f"{last['done']} timed B={self.batch} fp32 train steps (first untimed), {last['t']:.1f} s"}
        if last and last.get("first"):
            return {"threads": th, "value": 1.0 / last["first"],
                    "sample": f"1 cold B={self.batch} fp32 train step ({last['first']:.1f} s, includes allocator / oneDNN warm-up)"}
        return {"threads": th, "value": None, "sample": "no step finished within the 120 s cap"}

    def _video_multi(self, th):
        last = _cpu_collect(_cpu_spawn(f"_cpu_video_worker({th}, 10.0, 4, False)", th), 90.0)
        if last and last["done"] >= 1:
            per = last["t"] / last["done"]
            return {"threads": th, "value": 7.0 / (per * self.video_steps),
                    "sample": f"{last['done']} timed full-size Unet_Libero forwards at B=1 ({per:.2f} s each, first untimed)"}
        return {"threads": th, "value": None, "sample": "no forward finished within the 90 s cap"}

    def policy(self):
        return self.results.get("policy")

    def video(self):
        return self.results.get("video")

    def _pack(self, settings, unit, what):
        good = [s for s in settings if s["value"]]
        best = max(good, key=lambda s: s["value"]) if good else None
        return {"value": best["value"] if best else None, "unit": unit, "cores": best["threads"] if best else 0, "kind": "port",
                "cpu_model": self.model, "usable_cores": self.usable,
                "sample": (what + "; best of the settings below: " + best["sample"]) if best else what + "; nothing finished",
                "settings": settings, "isolation": "run to completion before the first GPU leg is timed (no CPU-oracle process alive "
                                                   "inside a GPU timed window)"}


def _median(xs):
    xs = sorted(xs)
    n = len(xs)
    return xs[n // 2] if n % 2 else 0.5 * (xs[n // 2 - 1] + xs[n // 2])


# algorithmic FLOPs per sample per denoise step of Unet_Libero (FlopCounterMode on the reference)
VIDEO_TFLOP = {(128, 7): 2.130, (256, 15): 18.379}


def video_leg(torch, device, batch=16, sampling_steps=50, traffic_leg="video", size=128, frames=7, reps=3, roofline=True, workload=None):
    """AVDC sampler, Unet_Libero (201 M parameters, random init), `frames` predicted frames + 1 conditioning frame at size x size,
    CLIP-free synthetic task tokens [B,10,512].  sampling_steps < 100 = the reference's DDIM branch (goal_diffusion.py:405,647),
    100 = its released configuration (ancestral, config/libero/lb_tk8_65to72.py:40-47).  One untimed call first (weight packs,
    workspaces, hipGraph capture), then `reps` sample() calls, each bracketed by HIP events on the launch stream; value = B * frames /
    MEDIAN call time, inputs resident in HBM."""
    from flowdiffusion.flowdiffusion.unet import Unet_Libero
    from flowdiffusion.flowdiffusion.goal_diffusion import GoalGaussianDiffusion
    from v2a_hip import ops
    torch.manual_seed(0)
    unet = Unet_Libero().to(device).eval()
    C = 3 * frames
    d = GoalGaussianDiffusion(unet, image_size=(size, size), channels=C, timesteps=100, sampling_timesteps=sampling_steps, loss_type="l2",
                              objective="pred_v", beta_schedule="cosine", min_snr_loss_weight=True, guidance_weight=0).to(device)
    g = torch.Generator(device=device).manual_seed(1)
    x_cond = torch.rand(batch, 3, size, size, device=device, generator=g)
    te = torch.randn(batch, 10, 512, device=device, generator=g)
    out = d.sample(x_cond, te, batch_size=batch)                    # warm-up: packs, workspace, allocator, graph capture
    torch.cuda.synchronize()
    times = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = d.sample(x_cond, te, batch_size=batch)
        e1.record()
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1) * 1e-3)
    dt = _median(times)
    flops = VIDEO_TFLOP[(size, frames)] * 1e12 * batch * sampling_steps
    res = {"metric": "video_frames_per_sec", "value": batch * frames / dt, "unit": "predicted frames/s (x(f+1)/f incl. the conditioning frame)",
           "seconds_per_sample_call": dt, "timing": f"median of {reps} sample() calls, HIP events on the launch stream", "call_seconds": times,
           "config": {"workload": workload or f"AVDC sampler Unet_Libero {size}x{size}, 1+{frames} frames", "batch": batch,
                      "sampling_steps": sampling_steps, "sampler": "ddim" if sampling_steps < 100 else "ddpm (ancestral)", "guidance_weight": 0,
                      "hip_graph": True, "noise": "in-kernel Philox"},
           "dtype": "f32", "algorithmic_tflops": flops / dt / 1e12, "output_range": [float(out.min()), float(out.max())]}
    import v2a_hip as _v
    # UNet forwards this leg ran in total (profiling scripts divide whole-run counters by it): the untimed call (+ the one eager step
    # in front of the graph capture), the timed calls, the instrumented forward below
    res["unet_forwards"] = sampling_steps * (1 + reps) + (1 if _v.sampler_graphs_enabled() else 0) + (1 if roofline else 0)
    if not roofline:
        return res
    # instrumented single UNet forward: per-variant conv timing with HIP events
    recs = []
    orig = ops.conv2d

    def wrap(orig_fn):
        def timed(*a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ret = orig_fn(*a, **k)
            e1.record()
            y = ret[0] if isinstance(ret, tuple) else ret
            x, cout, kh, kw = a[0], a[3], a[4], a[5]
            c2 = k.get("x2").shape[-1] if k.get("x2") is not None else 0
            m = y.shape[0] * y.shape[1] * y.shape[2]
            recs.append((ops.last_kernel[0], 2.0 * m * cout * kh * kw * (x.shape[-1] + c2), e0, e1))
            return ret
        return timed

    orig_h = ops.conv2d_h
    timed, timed_h = wrap(orig), wrap(orig_h)
    ops.conv2d = timed
    ops.conv2d_h = timed_h
    orig_gn = ops.conv2d_x3p_gn                    # fp32 3x3 convs with GroupNorm folded into the loader: (pg, x4, w_packed, bias, Cout, fps)

    def timed_gn(pg, x4, wp, bias, cout, fps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        y = orig_gn(pg, x4, wp, bias, cout, fps)
        e1.record()
        recs.append((ops.last_kernel[0], 2.0 * y.shape[0] * y.shape[1] * y.shape[2] * cout * 9 * x4.shape[-1], e0, e1))
        return y

    ops.conv2d_x3p_gn = timed_gn
    orig_u4 = ops.conv2d_x3p_ups4                  # Upsample + 3x3 as four class convs: (x, w_ups4, bias, Cout); algorithmic FLOPs = nine taps

    def timed_u4(xs, w4, bias, cout):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        y = orig_u4(xs, w4, bias, cout)
        e1.record()
        recs.append((ops.last_kernel[0], 2.0 * y.shape[0] * y.shape[1] * y.shape[2] * cout * 9 * xs.shape[-1], e0, e1))
        return y

    ops.conv2d_x3p_ups4 = timed_u4
    orig_u4h = ops.conv2d_hp_ups4

    def timed_u4h(xs, w4, bias, cout):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        y = orig_u4h(xs, w4, bias, cout)
        e1.record()
        recs.append((ops.last_kernel[0], 2.0 * y.shape[0] * y.shape[1] * y.shape[2] * cout * 9 * xs.shape[-1], e0, e1))
        return y

    ops.conv2d_hp_ups4 = timed_u4h
    try:
        eng = unet._engine()
        lab = eng.label_embedding(te)
        xin = ops.video_pack2(torch.randn(batch, C, size, size, device=device), x_cond, frames, size, size)
        eng.forward_cl(xin, torch.full((batch,), 50, dtype=torch.long, device=device), lab)
        torch.cuda.synchronize()
    finally:
        ops.conv2d = orig
        ops.conv2d_h = orig_h
        ops.conv2d_x3p_gn = orig_gn
        ops.conv2d_x3p_ups4 = orig_u4
        ops.conv2d_hp_ups4 = orig_u4h
    agg = {}
    for name, fl, e0, e1 in recs:
        v = agg.setdefault(name, [0.0, 0.0, 0])
        v[0] += fl; v[1] += e0.elapsed_time(e1) * 1e-3; v[2] += 1
    famv = {}                                     # source-kernel families (every instance of one template is one kernel)
    for k, v in agg.items():
        d = famv.setdefault(_family(k), [0.0, 0.0, 0])
        d[0] += v[0]; d[1] += v[1]; d[2] += v[2]
    name, (fl, sec, cnt) = max(famv.items(), key=lambda kv: kv[1][1])
    ach = fl / sec / 1e12
    std = batch == 16 and size == 128
    x3 = "x3" in name                             # fp32 products from three bf16 planes: the kernel's peak is the bf16 matrix peak / 6
    pk = (BF16_MFMA_PEAK_TFLOPS / 6.0) if x3 else FP32_MFMA_PEAK_TFLOPS
    res["roofline"] = {"kernel": name, "bound": "mfma", "achieved": ach, "peak": pk, "unit": "TFLOP/s",
                       "frac": ach / pk, "frac_of_f32_mfma_peak": ach / FP32_MFMA_PEAK_TFLOPS,
                       "peak_is": ("dense bf16 MFMA peak 2500 / 6 plane products per fp32 product" if x3 else "dense MFMA peak of the dtype"),
                       "end_to_end_frac_of_f32_mfma_peak": flops / dt / 1e12 / FP32_MFMA_PEAK_TFLOPS,
                       "traffic": _traffic(traffic_leg, name) if std else None,
                       "traffic_source": TRAFFIC_SOURCE if std else None,
                       "traffic_vs_algorithmic": _leg_traffic(traffic_leg) if std else None, "launches": cnt, "avg_launch_us": sec / cnt * 1e6,
                       "share_of_conv_time": sec / sum(v[1] for v in agg.values()),
                       "end_to_end_frac": flops / dt / 1e12 / pk,
                       "frac_src": "HIP events around every launch of one eager UNet forward of this run (ms-scale launches: eager = in-graph)",
                       "families": {k: {"tflops": v[0] / v[1] / 1e12, "ms_per_unet_fwd": v[1] * 1e3, "launches": v[2]} for k, v in sorted(famv.items())},
                       "all_conv_variants": {k: {"tflops": v[0] / v[1] / 1e12, "ms_per_unet_fwd": v[1] * 1e3, "launches": v[2]}
                                             for k, v in sorted(agg.items())}}
    return res


def _as_fp16(vb, note):
    vb = _as_bf16(vb, note)
    vb["dtype"] = "fp16 (IEEE half) activations / weights in HBM, fp16 MFMA (v_mfma_f32_32x32x16_f16), f32 accumulate / norm statistics / softmax"
    return vb


def _as_bf16(vb, note):
    if "roofline" in vb:
        vb["roofline"]["peak"] = BF16_MFMA_PEAK_TFLOPS
        vb["roofline"]["frac"] = vb["roofline"]["achieved"] / BF16_MFMA_PEAK_TFLOPS
        vb["roofline"]["end_to_end_frac"] = vb["algorithmic_tflops"] / BF16_MFMA_PEAK_TFLOPS
    vb["dtype"] = "bf16 activations / weights in HBM, bf16 MFMA, f32 accumulate / norm statistics / softmax"
    vb["note"] = note
    return vb


def video_train_leg(torch, device, batch=2, steps=4, warmup=2):
    """One optimisation step of the video model: q_sample -> Unet_Libero forward with tape -> loss -> hand-written
    backward -> clip + Adam + EMA, fp32 parity configuration and the bf16-MFMA mode."""
    import copy
    import v2a_hip
    from flowdiffusion.flowdiffusion.unet import Unet_Libero
    from flowdiffusion.flowdiffusion.goal_diffusion import GoalGaussianDiffusion
    from v2a_hip.video_train import VideoTrainStep
    torch.manual_seed(0)
    m = Unet_Libero().to(device)
    d = GoalGaussianDiffusion(m, image_size=(128, 128), channels=21, timesteps=100, sampling_timesteps=100, loss_type="l2", objective="pred_v",
                              beta_schedule="cosine", min_snr_loss_weight=True, guidance_weight=0).to(device)
    ts = VideoTrainStep(d, copy.deepcopy(d).requires_grad_(False))
    img, cond = torch.rand(batch, 21, 128, 128, device=device), torch.rand(batch, 3, 128, 128, device=device)
    te = torch.randn(batch, 8, 512, device=device)
    res = {"workload": "Unet_Libero (201 M parameters) training step, 7 frames 128x128, loss_type l2 / pred_v / min-SNR", "batch": batch}
    old = "bf16" if v2a_hip.get_precision() == "bf16" else "fp32"
    try:
        for mode in ("fp32", "bf16"):
            v2a_hip.set_precision(mode)
            m.__dict__.pop("_train_eng", None)
            for _ in range(warmup):
                loss = ts.step(img, cond, te)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                loss = ts.step(img, cond, te)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps
            res[mode] = {"ms_per_step": dt * 1e3, "samples_per_sec": batch / dt, "loss": float(loss)}
    finally:
        v2a_hip.set_precision(old)
    res["note"] = "fp32 = parity configuration; bf16 = bf16 MFMA inputs (twins of every packed operand), fp32 storage / accumulate / optimiser"
    return res


def _timed_policy_steps(torch, tr, steps, barrier):
    """The driver contract: K steps bracketed by barrier + synchronize on both sides (wall clock), plus one HIP event per step on the
    launch stream -> the median step interval."""
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    barrier()
    t0 = time.perf_counter()
    for i in range(steps):
        evs[i].record()
        tr.step()
    evs[steps].record()
    barrier()
    dt = time.perf_counter() - t0
    iv = [evs[i].elapsed_time(evs[i + 1]) for i in range(steps)]
    return dt, _median(iv)


def _rccl_algo(path):
    """What RCCL logged about the all-reduce (NCCL_DEBUG=INFO, subsystems INIT + TUNING): the `-> Algo a proto p` decisions by message
    size and the ring / tree channel lines."""
    import glob
    import re
    names = {0: "tree", 1: "ring", 2: "collnet_direct", 3: "collnet_chain", 4: "nvls", 5: "nvls_tree"}
    out = {"log": path, "decisions": [], "channels": None}
    try:
        txt = ""
        for f in sorted(glob.glob(path.replace("%h", "*").replace("%p", "*")))[:1]:
            txt = open(f, errors="replace").read()
        for m in re.finditer(r"AllReduce:?\s*(\d+)\s*Bytes\s*->\s*Algo\s*(\d+)\s*proto\s*(\d+)", txt):
            d = {"bytes": int(m.group(1)), "algo": names.get(int(m.group(2)), m.group(2)), "proto": int(m.group(3))}
            if d not in out["decisions"]:
                out["decisions"].append(d)
        out["decisions"] = out["decisions"][:8]
        ch = re.findall(r"(\d+) coll channels", txt)
        out["channels"] = int(ch[0]) if ch else None
        out["rings_logged"] = len(re.findall(r"Ring \d+ :", txt)) or len(re.findall(r"Channel \d+/\d+ :", txt))
        out["trees_logged"] = len(re.findall(r"Trees? \[", txt))
        if not txt:
            out["note"] = "no RCCL log found (NCCL_DEBUG_FILE)"
    except Exception as e:
        out["note"] = f"{type(e).__name__}: {e}"
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--force-dp", action="store_true", help="one rank through the data-parallel step structure and RCCL (PolicyTrainer(force_dp=True))")
    ap.add_argument("--no-video", action="store_true", help="skip the video-sampler legs (BASELINE.json configs[2], released config, C5, B=1)")
    ap.add_argument("--no-video-train", action="store_true", help="skip the video-model training-step leg")
    ap.add_argument("--no-predict", action="store_true", help="skip the predict_action latency leg")
    ap.add_argument("--no-roofline-pass", action="store_true", help="skip the instrumented eager pass (profiling runs)")
    ap.add_argument("--precision", choices=["fp32", "bf16"], default="fp32",
                    help="MFMA precision of the contraction kernels: fp32 = exact-f32 (parity configuration, default); bf16 = bf16 inputs, "
                         "fp32 accumulate/storage (performance configuration)")
    ap.add_argument("--no-bf16-extra", action="store_true", help="skip the additional bf16-mode measurements appended to the fp32 run")
    ap.add_argument("--video-batch", type=int, default=16)
    ap.add_argument("--video-steps", type=int, default=50)
    ap.add_argument("--video-reps", type=int, default=3)
    ap.add_argument("--watchdog", type=int, default=0, help="seconds after which every rank dumps its Python stacks to stderr and exits (0: off)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher around us: become one.  One process per GPU under torch.distributed.run (rendezvous on 127.0.0.1); rank 0's
        # JSON line passes through on stdout.
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd, env=dict(os.environ, OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", "8"))))

    if args.watchdog > 0:                                  # in the ranks, not in the launcher above
        import faulthandler
        faulthandler.dump_traceback_later(args.watchdog, exit=True)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    force_dp = args.force_dp                              # one rank through the N > 1 step structure + RCCL (validation on a 1-GPU box)
    rccl_log = None
    if world > 1 or force_dp:
        # what RCCL decides for the gradient all-reduce is part of the result (comm.algo): have it log its tuning decisions to a file
        rccl_log = os.environ.setdefault("NCCL_DEBUG_FILE", f"/tmp/v2a_rccl_{os.getpid()}_%h_%p.log")
        os.environ.setdefault("NCCL_DEBUG", "INFO")
        os.environ.setdefault("NCCL_DEBUG_SUBSYS", "INIT,TUNING")

    # ---- CPU baseline FIRST (rank 0, N = 1): the GPU is idle and nothing is being timed on it while the CPU oracle runs
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = CpuBaselines(args.batch, args.video_steps, video=not args.no_video)
        cpu.run_all()

    import numpy as np
    import random
    import torch
    if args.gpus != world and world > 1:
        print(f"warning: --gpus {args.gpus} != WORLD_SIZE {world}", file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    ndev = torch.cuda.device_count()
    dev_index = local_rank % ndev                 # more ranks than GPUs (validation on a 1-GPU box): ranks share devices ...
    torch.cuda.set_device(dev_index)
    device = f"cuda:{dev_index}"
    pg = None
    backend = None
    if world > 1 or force_dp:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        backend = "nccl" if world <= ndev else "gloo"     # ... and RCCL refuses two ranks on one device: gloo moves the arena then
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(device))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        pg = dist.group.WORLD
        world = dist.get_world_size()                      # the ranks the communicator actually initialised

    from diffuser.diffusion_policy.get_dp import build_policy, DEFAULT_CONF
    from v2a_hip.trainer import PolicyTrainer
    import v2a_hip
    v2a_hip.set_precision(args.precision)
    torch.manual_seed(0)                       # identical replica init on every rank
    pol = build_policy(DEFAULT_CONF).to(device)
    np.random.seed(rank)
    random.seed(rank)
    store = build_store(torch, device, args.batch, seed=100 + rank)
    tr = PolicyTrainer(pol, store, batch_size=args.batch, seed=rank, use_graph=not args.no_graph, process_group=pg,
                       world_size=world, rank=rank, force_dp=force_dp)

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if world <= 1:
            return x
        import torch.distributed as dist
        t = torch.tensor([x], dtype=torch.float64, device="cpu" if backend == "gloo" else device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    for _ in range(max(args.warmup, 3)):       # >= 3: two eager steps + the capture step
        tr.step()

    # ---- which gradient exchange the timed steps run on (comm.algo).  The two slice all-reduces alone (nothing to hide under), blocking,
    # 5 rounds, through the process group (RCCL on a node); if that moves the 349 MB below DIRECT_BELOW_GBPS of bus bandwidth -- a ring on
    # the xGMI mesh is bound by ONE link per direction, ~153 GB/s peak, SURVEY.md section 5 -- the peer-pointer exchange (v2a_hip/dp.py
    # algo="direct", csrc/dp.hip) is connected and measured the same way, and the faster of the two carries the timed steps.  Two ranks on
    # two GPUs have ONE link between them whichever algorithm drives it: the direct exchange is tried from three ranks on (and when the
    # ranks share a GPU, where the process group is gloo through the host).  Every number a decision hangs on is a max over ranks, so all
    # ranks decide alike.
    DIRECT_BELOW_GBPS = 200.0
    choice = None

    def iso_ms(launch_all):
        ts = []
        for _ in range(5):
            barrier()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            launch_all()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        tr.arena.zero_()
        return max_over_ranks(_median(ts))

    def busbw(ms_):
        return tr.reducer.bytes_per_step() * 2.0 * (world - 1) / world / (ms_ * 1e-3) / 1e9

    def pg_all():
        import torch.distributed as dist
        for lo, hi in tr.reducer.slices:
            if hi > lo:
                dist.all_reduce(tr.arena[lo:hi], op=dist.ReduceOp.SUM, group=pg)

    def direct_all():
        tr.reducer.launch(0)
        tr.reducer.launch(1)
        tr.reducer.finish(lambda scale: None)

    iso_pg = iso_direct = None
    if tr.dp:
        iso_pg = iso_ms(pg_all)
        if world > 1:
            choice = {"threshold_busbw_GBps": DIRECT_BELOW_GBPS, "process_group_busbw_GBps": busbw(iso_pg), "picked": "rccl",
                      "rule": "process group below the threshold and (>= 3 ranks or ranks sharing a GPU) -> connect the peer-pointer exchange, "
                              "measure it alike, keep the faster"}
            if busbw(iso_pg) < DIRECT_BELOW_GBPS and (world >= 3 or backend == "gloo"):
                try:
                    tr.set_dp_algo("direct")                   # raises on every rank or on none
                    # a known answer before anything is timed on it: every rank contributes rank + 1 everywhere, the sum is W (W + 1) / 2
                    tr.arena.fill_(float(rank + 1))
                    direct_all()
                    torch.cuda.synchronize()
                    tr.reducer.check()
                    wrong = float((tr.arena != world * (world + 1) / 2.0).any().item())
                    tr.arena.zero_()
                    if max_over_ranks(wrong) != 0.0:
                        raise RuntimeError("the peer-pointer exchange returned a wrong sum on this node (known-answer check)")
                    for _ in range(2):
                        tr.step()
                    iso_direct = iso_ms(direct_all)
                    torch.cuda.synchronize()
                    tr.reducer.check()
                    choice["direct_busbw_GBps"] = busbw(iso_direct)
                    if iso_direct < iso_pg:
                        choice["picked"] = "direct"
                    else:
                        tr.set_dp_algo("rccl")
                except RuntimeError as e:
                    choice["direct_error"] = str(e)[:400]
                    if tr.reducer.algo != "rccl":
                        tr.set_dp_algo("rccl")
    if tr.dp:
        tr.comm_events = []
    dt, med_ms = _timed_policy_steps(torch, tr, args.steps, barrier)
    dt = max_over_ranks(dt)
    comm = None
    if tr.dp:
        if tr.reducer.algo == "direct":
            torch.cuda.synchronize()
            tr.reducer.check()
        # exposed communication: how long the compute stream sat in its wait on the exchange, per step (HIP events recorded on the launch
        # stream right before / after GradReducer.finish; with the direct exchange the wait is the join with the side stream of slice 0)
        exposed = [a.elapsed_time(b) for a, b in tr.comm_events]
        tr.comm_events = None
        nbytes = tr.reducer.bytes_per_step()
        iso_used = iso_direct if tr.reducer.algo == "direct" else iso_pg
        comm = {"algo": tr.reducer.algo,
                "backend": ("peer pointers (hipIpc), csrc/dp.hip; handles exchanged through " + backend) if tr.reducer.algo == "direct"
                else ("rccl (torch.distributed nccl)" if backend == "nccl" else "gloo (ranks share a GPU: RCCL needs one device per rank)"),
                "choice": choice,
                "rccl_ranks": world if backend == "nccl" else 0, "ranks": world, "physical_gpus": min(world, ndev),
                "allreduce_bytes_per_step": nbytes, "slices": [hi - lo for lo, hi in tr.reducer.slices],
                "allreduce_ms_isolated": iso_used,
                "allreduce_busbw_GBps": busbw(iso_used) if world > 1 else None,
                "exposed_comm_ms_per_step": sum(exposed) / max(len(exposed), 1),
                "rccl_algo": _rccl_algo(rccl_log) if (backend == "nccl" and rccl_log) else None,
                "note": "slice 0 (ConditionalUnet1D gradients) is launched after backward phase 1 and travels under the image-encoder "
                        "backward; exposed = compute-stream wait on the exchange, measured with HIP events inside the timed steps"}
    loss = float(tr.loss.item())

    out = None
    ms = dt / args.steps * 1e3
    if rank == 0:
        value = world * args.steps / dt
        P = 87219143
        flops_step = 8.722e9 * args.batch                  # fwd+bwd algorithmic FLOPs per sample
        out = {"metric": "policy_train_steps_per_sec", "value": value, "unit": f"steps/s (batch-{args.batch} steps, all ranks)", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None,
               "dtype": ("f32 (f32 storage / accumulate / optimiser; conv products from three bf16 planes per operand = fp32-equivalent accuracy, "
                         "V2A_F32_CONV=exact selects the exact-f32 MFMA kernels)") if args.precision == "fp32"
               else "bf16 (MFMA inputs; f32 accumulate, f32 storage/optimizer)", "data": "synthetic",
               "config": {"workload": "Libero 8-task diffusion-policy train step (BASELINE.json configs[1]): R1 replay gather -> "
                                      "compute_loss fwd/bwd -> clip -> AdamW -> EMA", "batch_per_gpu": args.batch,
                          "global_batch": args.batch * world, "image": "128x128x3 uint8 start+goal", "action": "16x7",
                          "params": P, "parallelism": f"dp{world}", "hip_graph": not args.no_graph},
               "ms_per_step_median_hip_events": med_ms,
               "samples_per_sec": value * args.batch, "final_loss": loss,
               "step_algorithmic_tflops": flops_step / (ms * 1e-3) / 1e12}
        if comm is not None:
            out["comm"] = comm

    # ---- N > 1: the sampler sharded by batch rows and the joint loop (BASELINE configs[3]); every rank takes part
    if world > 1 and not args.no_video:
        try:
            v2a_hip.set_video_storage("bf16")
            from v2a_hip.dp import shard_rows, shard_tasks
            lo_r, hi_r = shard_rows(args.video_batch, world, rank)
            rows = max(1, hi_r - lo_r)
            barrier()
            vs = video_leg(torch, device, rows, args.video_steps, traffic_leg="video_bf16", reps=args.video_reps, roofline=False,
                           workload="BASELINE configs[2] sharded by batch rows over the ranks")
            t_call = max_over_ranks(vs["seconds_per_sample_call"])
            # per-task exploration rollouts of the released config (lb_online_trainer_v7.py:871,888-891: 8 tasks, bs = 1, 100 ancestral
            # steps): tasks r, r + N, ... on rank r
            tasks = len(shard_tasks(8, world, rank))
            barrier()
            vr = video_leg(torch, device, 1, 100, reps=1, roofline=False) if tasks else None
            t_roll = max_over_ranks((vr["seconds_per_sample_call"] * tasks) if vr else 0.0)
            if rank == 0:
                out["video_bf16"] = _as_bf16({"metric": "video_frames_per_sec", "value": rows * world * 7 / t_call,
                                              "unit": "predicted frames/s, all ranks", "rows_per_rank": rows, "seconds_per_sample_call": t_call,
                                              "config": vs["config"], "timing": vs["timing"] + "; MAX over ranks",
                                              "algorithmic_tflops": 2.130 * rows * world * args.video_steps / t_call},
                                             "the B rows of one sample() call are split over the ranks (rows of a batch are independent): no collective")
                every = 200                                      # config/libero/lb_tk8_65to72.py:84-90: one exploration round per 200 steps
                out["joint"] = {"workload": "BASELINE configs[3]: policy train steps + one video-guided exploration round (8 tasks, bs 1, 100 "
                                            "ancestral steps, tasks split over the ranks) every 200 steps", "rollout_every_steps": every,
                                "sampler_seconds_per_round": t_roll, "tasks_per_rank_max": -(-8 // world),
                                "steps_per_sec_incl_sampling": world * every / (every * ms * 1e-3 + t_roll),
                                "note": "the policy rollouts inside a round run in the simulator (out of scope); only the video sampler's share is timed"}
            v2a_hip.set_video_storage("f32")
        except Exception as e:
            if rank == 0:
                out["video_bf16"] = {"error": f"{type(e).__name__}: {e}"}

    # ---- roofline of the dominant kernel (rank 0, N=1 only): instrumented eager pass
    if rank == 0 and world == 1:
        if not args.no_roofline_pass:
            agg = instrumented_pass(torch, tr, 3)
            tot = sum(v[1] for v in agg.values())
            # Kernel FAMILIES = source kernels (every instance of one template: conv_igemm_f32x3<BM, BN, ...>, conv_halo_x3<OW> ...); the
            # per-instance figures stay in all_conv_variants.  FLOPs per launch come from the instrumented eager pass (the launcher's
            # shapes); the DURATIONS that price them are the in-graph ones of the captured step, measured by a rocprofv3 --kernel-trace
            # --stats child of this very run (the eager figure stays on the line as avg_launch_us_eager).
            fam = {}
            for k, v in agg.items():
                d = fam.setdefault(_family(k), [0.0, 0.0, 0])
                d[0] += v[0]; d[1] += v[1]; d[2] += v[2]
            live, live_src = _live_kernel_stats(args.precision, args.batch)
            if live:
                have = {k: v for k, v in fam.items() if k in live}
                name = max(have, key=lambda k: live[k][1]) if have else max(fam, key=lambda k: fam[k][1])
            else:
                name = max(fam, key=lambda k: fam[k][1])
            fl, sec, cnt = fam[name]
            x3 = "x3" in name                        # fp32 products from three bf16 planes: six bf16 MFMAs per product block
            peak = (BF16_MFMA_PEAK_TFLOPS / 6.0) if x3 else (FP32_MFMA_PEAK_TFLOPS if args.precision == "fp32" else BF16_MFMA_PEAK_TFLOPS)
            eager_tf = fl / sec / 1e12
            if live and name in live:
                g_us = live[name][1] / live[name][0] / 1e3
                frac_source = "in-graph launch durations: rocprofv3 --kernel-trace --stats child of this bench run"
                conv_live = sum(v[1] for k, v in live.items() if k in fam)
                share = live[name][1] / conv_live if conv_live else None
            else:
                g_us = (_graph_avg_us(name + "<") or _graph_avg_us(name + "_kernel") or _graph_avg_us(name))
                frac_source = f"in-graph durations from the COMMITTED profiles/r05_policy_kernel_stats.csv ({live_src}); eager figure if absent"
                share = sec / tot
            achieved = ((fl / cnt) / (g_us * 1e-6) / 1e12) if g_us else eager_tf
            out["roofline"] = {"kernel": name, "bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                               "frac": achieved / peak, "frac_src": frac_source,
                               "peak_is": ("dense bf16 MFMA peak 2500 / 6 plane products per fp32 product (three-plane kernels)" if x3 else
                                           "dense MFMA peak of the dtype"),
                               "frac_of_f32_mfma_peak": achieved / FP32_MFMA_PEAK_TFLOPS if args.precision == "fp32" else None,
                               "frac_eager": eager_tf / peak, "avg_launch_us": g_us if g_us else sec / cnt * 1e6,
                               "avg_launch_us_eager": sec / cnt * 1e6, "live_stats_csv": live_src if live else None,
                               "traffic": _traffic("policy", name), "traffic_src": "committed profiles/roofline_traffic.json (PMC passes of a separate run)",
                               "traffic_source": TRAFFIC_SOURCE,
                               "traffic_vs_algorithmic": _leg_traffic("policy") if (args.batch == 64 and args.precision == "fp32") else None,
                               "launches": cnt, "launches_per_step": cnt / 3, "algorithmic_gflop_per_launch": fl / cnt / 1e9,
                               "share_of_conv_time": share, "whole_step_frac_of_mfma_floor": flops_step / (ms * 1e-3) / 1e12 / peak,
                               "whole_step_frac_of_f32_mfma_floor": flops_step / (ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS,
                               "families": {k: {"tflops_eager": v[0] / v[1] / 1e12, "launches_per_step": v[2] / 3,
                                                "tflops_in_graph": ((v[0] / v[2]) / (live[k][1] / live[k][0] * 1e-9) / 1e12) if (live and k in live) else None,
                                                "ms_per_step_in_graph": (live[k][1] / live[k][0] * (v[2] / 3) * 1e-6) if (live and k in live) else None}
                                            for k, v in sorted(fam.items())},
                               "all_conv_variants": {k: {"tflops": v[0] / v[1] / 1e12, "ms_per_step": v[1] / 3 * 1e3, "launches_per_step": v[2] / 3}
                                                     for k, v in sorted(agg.items())}}
        if cpu is not None:
            out["cpu_baseline"] = cpu.policy()
        if args.precision == "fp32" and not args.no_bf16_extra:
            # the same step in the bf16-MFMA performance configuration (fresh trainer: new hipGraph), reported beside the parity run.
            # BASELINE configs[1] names bf16: this leg is HBM-bound (64 B x 87.2 M parameters + 3 x 3.326 M x B x 4 B of
            # activations = 8.13 GB per step against a 0.22 ms MFMA floor), so its roofline is bytes / time against 8 TB/s.
            v2a_hip.set_precision("bf16")
            torch.manual_seed(0)
            pol2 = build_policy(DEFAULT_CONF).to(device)
            tr2 = PolicyTrainer(pol2, store, batch_size=args.batch, seed=rank, use_graph=not args.no_graph)
            for _ in range(max(args.warmup, 3)):
                tr2.step()
            dt2, med2 = _timed_policy_steps(torch, tr2, args.steps, barrier)
            ms2 = dt2 / args.steps * 1e3
            alg_bytes = 64.0 * 87219143 + 3 * 3.326e6 * args.batch * 4
            out["bf16"] = {"metric": "policy_train_steps_per_sec", "value": args.steps / dt2, "ms_per_step": ms2,
                           "ms_per_step_median_hip_events": med2,
                           "dtype": "bf16 MFMA inputs, f32 accumulate/storage/optimizer", "final_loss": float(tr2.loss.item()),
                           "roofline": {"kernel": "whole captured step (HBM-bound leg: no single kernel dominates)", "bound": "hbm",
                                        "achieved": alg_bytes / (ms2 * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                        "frac": alg_bytes / (ms2 * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_step": alg_bytes,
                                        "traffic": (_leg_traffic("policy_bf16") or {}).get("hbm_bytes") if args.batch == 64 else None,
                                        "traffic_source": TRAFFIC_SOURCE if (args.batch == 64 and _leg_traffic("policy_bf16")) else None,
                                        "traffic_vs_algorithmic": _leg_traffic("policy_bf16") if args.batch == 64 else None,
                                        "mfma_floor_ms": 8.722e9 * args.batch / (BF16_MFMA_PEAK_TFLOPS * 1e12) * 1e3,
                                        "hbm_floor_ms": alg_bytes / (HBM_PEAK_GBS * 1e9) * 1e3},
                           "note": "performance configuration (what BASELINE configs[1] names); parity (1e-4) is claimed for the fp32 run only"}
            del tr2, pol2
            v2a_hip.set_precision("fp32")
        # predict_action latency (B=1, DDIM-8, EMA-style replica) under hipGraph replay
        try:
            if args.no_predict:
                raise KeyboardInterrupt
            from v2a_hip.inference import GraphedPredictAction
            polq = tr.ema_for_inference()
            obs1 = {k: torch.rand(1, 1, 3, 128, 128, device=device) for k in ("img_obs_1", "img_goal_1")}

            def pa_latency(persistent):
                gp = GraphedPredictAction(polq, batch_size=1, use_ddim=True, persistent=persistent)
                for _ in range(3):
                    gp(obs1)
                lat = []
                for _ in range(50):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    gp(obs1)
                    e1.record()
                    torch.cuda.synchronize()
                    lat.append(e0.elapsed_time(e1))
                return _median(lat), gp

            ms_p, gp_p = pa_latency(True)
            ms_l, _ = pa_latency(False)
            out["predict_action"] = {"latency_ms": ms_p, "batch": 1, "sampler": "ddim-8", "timing": "median of 50 HIP-event intervals",
                                     "path": "persistent denoiser (default at batch 1): csrc/policy_persist.hip",
                                     "layer_by_layer_ms": ms_l, "grid_barriers_per_call": gp_p.pp.n_barriers, "workgroups": gp_p.pp.nwg,
                                     "note": "encoders + 8 ConditionalUnet1D steps + scheduler updates + unnormalise, one hipGraph replay per "
                                             "call; the 8 steps are ONE persistent launch (exact fp32 FMA), layer_by_layer_ms = the same call on "
                                             "the training path's kernels; reference CPU path 110 ms (SURVEY section 6)"}
        except KeyboardInterrupt:
            pass
        except Exception as e:                      # never let the secondary leg break the headline line
            out["predict_action"] = {"error": f"{type(e).__name__}: {e}"}
        # ---- BASELINE configs[4], policy half at one GPU's share: B = 256 (fp32 and the 16-bit modes), with its rooflines
        def policy_leg(prec, batch, steps=6, warm=3, dp=False):
            v2a_hip.set_precision(prec)
            torch.manual_seed(0)
            polx = build_policy(DEFAULT_CONF).to(device)
            stx = build_store(torch, device, batch, seed=100) if batch != args.batch else store
            kw = {}
            if dp:
                import torch.distributed as dist
                if not dist.is_initialized():
                    import socket
                    with socket.socket() as sk:
                        sk.bind(("127.0.0.1", 0))
                        port = sk.getsockname()[1]
                    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device(device))
                kw = dict(process_group=dist.group.WORLD, world_size=1, force_dp=True)
            trx = PolicyTrainer(polx, stx, batch_size=batch, seed=0, use_graph=not args.no_graph, **kw)
            for _ in range(max(warm, 3)):
                trx.step()
            if dp:
                trx.phase_events, trx.comm_events = [], []
            dtx, medx = _timed_policy_steps(torch, trx, steps, barrier)
            msx = dtx / steps * 1e3
            r = {"ms_per_step": msx, "ms_per_step_median_hip_events": medx, "value": steps / dtx, "unit": f"steps/s (batch-{batch})",
                 "samples_per_sec": steps / dtx * batch, "batch": batch, "precision": prec, "final_loss": float(trx.loss.item())}
            flx = 8.722e9 * batch
            if prec == "fp32":
                r["whole_step_frac_of_f32_mfma_floor"] = flx / (msx * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS
            else:
                albytes = 64.0 * 87219143 + 3 * 3.326e6 * batch * 4
                r["roofline"] = {"bound": "hbm", "achieved": albytes / (msx * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": albytes / (msx * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_step": albytes}
            if getattr(trx, "loss_scaling", False):
                ls_, _, _, nskip_ = trx.opt.scaler()
                r["loss_scale"], r["skipped_steps"] = ls_, nskip_      # GradScaler calibration: the first steps at 65536 overflow and are skipped
            if dp:
                torch.cuda.synchronize()
                ph = [[a.elapsed_time(b) for a, b in zip(pe[:-1], pe[1:])] for pe in trx.phase_events]
                med = [_median([p_[i] for p_ in ph]) for i in range(4)] if ph else [None] * 4
                sl = [hi - lo for lo, hi in trx.reducer.slices]
                r.update({"graph_phase1_ms": med[0], "graph_encoder_backward_ms": med[1], "reduce_wait_ms": med[2], "graph_optimiser_ms": med[3],
                          "slice_elems": sl,
                          "slice0_window_ms": med[1],
                          "exposed_comm_estimate_ms_at_8_gpus": {
                              w: max(0.0, 2 * 7 / 8 * sl[0] * b / 153e9 * 1e3 - (med[1] or 0.0)) + 2 * 7 / 8 * sl[1] * b / 153e9 * 1e3
                              for w, b in (("fp32_wire", 4), ("bf16_wire", 2))},
                          "estimate_assumes": "ring all-reduce bound by ONE xGMI link at 153 GB/s (2 (N-1)/N x bytes / link); slice 0 hides under the "
                                              "encoder-backward graph, slice 1 is exposed"})
            del trx, polx
            v2a_hip.set_precision("fp32")
            return r

        def leg0(key, fn):
            try:
                torch.cuda.empty_cache()
                out[key] = fn()
            except Exception as e:
                out[key] = {"error": f"{type(e).__name__}: {e}"}

        def exact_mode(fn):
            """Run fn with the exact-f32 MFMA conv kernels (V2A_F32_CONV=exact / v2a_set_f32_conv_mode(0)): the configuration that is bit-equal
            to an fmaf chain per accumulator -- the cost of strict fp32 arithmetic next to the three-plane headline."""
            from v2a_hip._lib import lib as _lib
            old_mode = _lib.v2a_set_f32_conv_mode(0)
            try:
                r = fn()
            finally:
                _lib.v2a_set_f32_conv_mode(old_mode)
            r["f32_conv_mode"] = "exact (v_mfma_f32_32x32x2_f32)"
            return r

        if not args.no_bf16_extra and args.batch == 64:
            leg0("policy_exact", lambda: exact_mode(lambda: dict(policy_leg("fp32", args.batch, steps=10),
                                                                 workload="the headline step on the exact-f32 MFMA conv kernels")))
            leg0("policy_b256", lambda: {"workload": "BASELINE configs[4] policy half at one GPU's share: batch 256",
                                         "fp32": policy_leg("fp32", 256), "bf16": policy_leg("bf16", 256),
                                         **({"fp16": policy_leg("fp16", 256)} if hasattr(v2a_hip, "set_policy_half") else {})})
            # the data-parallel step STRUCTURE on one rank (three graphs + two slice all-reduces through RCCL with one rank): its cost
            # per GPU against the one-graph step above, and the window the encoder backward leaves for slice 0
            leg0("dp_structure", lambda: dict(policy_leg("fp32", args.batch, steps=10, dp=True), one_graph_ms_per_step=ms,
                                              workload="force_dp step structure, one rank, RCCL"))
        if not args.no_video:
            del tr, pol, store
            torch.cuda.empty_cache()
            R = args.video_reps

            def leg(key, fn):
                try:
                    torch.cuda.empty_cache()
                    out[key] = fn()
                except Exception as e:
                    out[key] = {"error": f"{type(e).__name__}: {e}"}

            leg("video", lambda: video_leg(torch, device, args.video_batch, args.video_steps, reps=R,
                                           workload="AVDC sampler Unet_Libero 128x128, 1+7 frames (BASELINE.json configs[2])"))
            if cpu is not None and "error" not in out["video"]:
                out["video"]["cpu_baseline"] = cpu.video()
            if args.precision == "fp32" and not args.no_bf16_extra:
                leg("video_exact", lambda: exact_mode(lambda: video_leg(
                    torch, device, args.video_batch, args.video_steps, reps=1, roofline=False,
                    workload="AVDC sampler (BASELINE.json configs[2]) on the exact-f32 MFMA conv kernels")))
            if args.precision == "fp32" and not args.no_bf16_extra:
                note16 = ("performance configuration (counterpart of the reference's fp16-autocast GPU path); 0.8 % relative L2 "
                          "deviation from the fp32 parity path per UNet forward (tests/test_video_gpu.py)")
                v2a_hip.set_video_storage("bf16")
                leg("video_bf16", lambda: _as_bf16(video_leg(torch, device, args.video_batch, args.video_steps, traffic_leg="video_bf16", reps=R,
                                                             workload="AVDC sampler Unet_Libero 128x128, 1+7 frames (BASELINE.json configs[2])"), note16))
                # the RELEASED sampler configuration (SURVEY 8d "also report the reference-config variant"): 100 ancestral DDPM steps
                leg("video_bf16_ddpm100", lambda: _as_bf16(video_leg(
                    torch, device, args.video_batch, 100, traffic_leg="video_bf16", reps=R, roofline=False,
                    workload="released sampler config (config/libero/lb_tk8_65to72.py:40-47): 100 ancestral steps, B=16"), note16))
                # what the trainer actually issues: ONE task at bs = 1 (lb_online_trainer_v7.py:871,888-891) -> latency per rollout
                def b1():
                    r = _as_bf16(video_leg(torch, device, 1, 100, reps=R, roofline=False,
                                           workload="per-task exploration rollout: bs = 1, 100 ancestral steps (lb_online_trainer_v7.py:871-891)"), note16)
                    r["latency_ms_per_rollout"] = r["seconds_per_sample_call"] * 1e3
                    r["ms_per_denoise_step"] = r["seconds_per_sample_call"] * 1e3 / 100
                    return r
                leg("video_b1", b1)
                # an exploration round = 8 per-task rollouts (lb_online_trainer_v7.py:871,888-891): one after another (what the reference
                # issues) against ONE B = 8 call of the same sampler
                def round8():
                    r = _as_bf16(video_leg(torch, device, 8, 100, reps=2, roofline=False,
                                           workload="exploration round as ONE call: bs = 8, 100 ancestral steps"), note16)
                    one = out.get("video_b1", {}).get("seconds_per_sample_call")
                    r["seconds_8_rollouts_as_one_b8_call"] = r["seconds_per_sample_call"]
                    r["seconds_8_rollouts_one_after_another"] = 8 * one if one else None
                    return r
                leg("video_round8", round8)
                try:                                     # BASELINE configs[3] arithmetic at ONE GPU: policy steps + one round per 200 steps
                    every = 200
                    t8 = out["video_round8"]["seconds_per_sample_call"]
                    t1 = out["video_b1"]["seconds_per_sample_call"] * 8
                    out["joint"] = {"workload": "BASELINE configs[3] at one GPU: policy train steps + one video-guided exploration round (8 tasks, "
                                                "100 ancestral steps) every 200 steps", "rollout_every_steps": every,
                                    "steps_per_sec": every / (every * ms * 1e-3 + t8),
                                    "product_path": "LB_Online_Trainer_V7.video_guided_explore samples a round as ONE batched call (explore_batched, default)",
                                    "steps_per_sec_incl_sampling_rollouts_one_by_one": every / (every * ms * 1e-3 + t1),
                                    "steps_per_sec_incl_sampling_one_b8_call": every / (every * ms * 1e-3 + t8),
                                    "sampler_seconds_per_round_one_by_one": t1, "sampler_seconds_per_round_b8": t8}
                except Exception as e:
                    out["joint"] = {"error": f"{type(e).__name__}: {e}"}
                # BASELINE configs[4], video half at a 1-GPU slice: 256x256, 16-frame sequence (1 cond + 15 predicted), per-GPU batch 2
                leg("video_c5", lambda: _as_bf16(video_leg(
                    torch, device, 2, args.video_steps, size=256, frames=15, reps=R, roofline=False,
                    workload="BASELINE configs[4] video half: 256x256, 1+15 frames, per-GPU sampler batch 2 (16 over 8 GPUs)"), note16))
                # BASELINE configs[4] says fp16 (and the reference's GPU path is fp16 autocast): the IEEE-half instances of the same kernels
                v2a_hip.set_video_storage("fp16")
                note_h = ("fp16 storage = the reference's own 16-bit type (lb_online_trainer_v7.py:72-76,889); 3 more mantissa bits than bf16, "
                          "same MFMA rate; tests/test_video_gpu.py measures its deviation from the fp32 parity path")
                leg("video_fp16", lambda: _as_fp16(video_leg(torch, device, args.video_batch, args.video_steps, reps=R, roofline=False,
                                                             workload="AVDC sampler Unet_Libero 128x128, 1+7 frames (BASELINE.json configs[2]), fp16 storage"), note_h))
                leg("video_c5_fp16", lambda: _as_fp16(video_leg(
                    torch, device, 2, args.video_steps, size=256, frames=15, reps=R, roofline=False,
                    workload="BASELINE configs[4] video half in fp16: 256x256, 1+15 frames, per-GPU sampler batch 2 (16 over 8 GPUs)"), note_h))
                v2a_hip.set_video_storage("f32")
            if not args.no_video_train:
                try:
                    torch.cuda.empty_cache()
                    out["video_train"] = video_train_leg(torch, device)
                except Exception as e:
                    out["video_train"] = {"error": f"{type(e).__name__}: {e}"}
    if rank == 0:
        _emit(out)
    sys.stdout.flush()
    try:
        import torch.distributed as _d
        made_pg = _d.is_available() and _d.is_initialized()
    except Exception:
        made_pg = False
    if world > 1 or force_dp or made_pg:
        import torch.distributed as dist
        try:                                   # the result line is out: never let communicator teardown turn into a failure
            torch.cuda.synchronize()
            dist.barrier()
            dist.destroy_process_group()
        except Exception as e:
            print(f"[bench] process-group teardown: {type(e).__name__}: {e}", file=sys.stderr)
        sys.stderr.flush()
        os._exit(0)                            # skip interpreter teardown (RCCL watchdog threads vs. graph / allocator destructors)


if __name__ == "__main__":
    main()
