"""Video-sampler leg only (for profiling): prints bench.video_leg's JSON object."""
import argparse
import json
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "video-to-action-release_amd"))
import torch
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=50)
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--precision", default="fp32")
ap.add_argument("--storage", default="f32", choices=["f32", "bf16"])
ap.add_argument("--no-graph", action="store_true", help="eager launches (PMC passes: rocprofv3 counters do not see kernels inside a replayed hipGraph)")
a = ap.parse_args()
import v2a_hip
if a.no_graph:
    v2a_hip.set_sampler_graphs(False)
v2a_hip.set_precision(a.precision)
v2a_hip.set_video_storage(a.storage)
print(json.dumps(bench.video_leg(torch, "cuda:0", a.batch, a.steps, traffic_leg="video_bf16" if a.storage == "bf16" else "video", reps=1)))
