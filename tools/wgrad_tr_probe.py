"""One twin-fed bf16 weight gradient at a video-training shape (run under rocprofv3 --pmc to read LDS bank conflicts / MFMA busy)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-to-action-release_amd"))
sys.path.insert(0, ROOT)
import torch
import v2a_hip
from v2a_hip import ops

v2a_hip.set_precision("bf16")
N, H, W, C, Co = 14, 64, 64, 256, 256
x = torch.randn(N, H, W, C, device="cuda:0")
dy = torch.randn(N, H, W, Co, device="cuda:0")
xh, dyh = ops.cast_h(x), ops.cast_h(dy)
f = lambda: ops.conv2d_wgrad(x, dy, (Co, C, 3, 3), 3, 3, (1, 1), (1, 1), x_h=xh, dy_h=dyh)
f(); f()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    f()
e1.record(); e1.synchronize()
t = e0.elapsed_time(e1) / 5 * 1e-3
print(f"{ops.last_kernel[0]}  {t*1e6:.1f} us  {2.0*N*H*W*C*9*Co/t/1e12:.1f} TF (incl. split reduce)")
