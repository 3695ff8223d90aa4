import os, sys
ROOT = "/root/repo"
sys.path.insert(0, os.path.join(ROOT, "video-to-action-release_amd"))
import torch
from v2a_hip import ops
from v2a_hip._lib import lib
dev = torch.device("cuda:0")
for C in (32, 64, 128, 256):
    N, S, Co = 64, 32, 64
    x = torch.randn(N, S, S, C, device=dev)
    w = torch.randn(Co, C, 3, 3, device=dev) * 0.05
    wp = ops.pack_weight(w, 0)
    for _ in range(20):
        y = ops.conv2d(x, wp, None, Co, 3, 3, (1, 1), (1, 1))
    torch.cuda.synchronize()
    print(C, ops.last_kernel[0])
