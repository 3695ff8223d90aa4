import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "video-to-action-release_amd"))
import numpy as np, torch, torch.nn.functional as F
from v2a_hip import ops
g = np.load(f"{ROOT}/tests/golden/policy.npz", allow_pickle=True)
dev = "cuda:0"
for nm in ("img_obs", "img_goal"):
    img = torch.from_numpy(g[nm]).reshape(-1, 3, 128, 128)
    N = img.shape[0]
    print(nm, img.shape, img.dtype, float(img.min()), float(img.max()))
    for rep in range(3):
        torch.manual_seed(rep)
        dy = torch.randn(N, 64, 64, 64) * torch.rand(N, 64, 64, 64).mul(8).exp2()
        xp = torch.zeros((N, 134, 134, 4), device=dev)
        ops.nchw_to_nhwc4p(img.to(dev), xp, 3, normalize=True)
        dw4 = ops.conv2d_wgrad(xp, dy.to(dev), (64, 4, 7, 7), 7, 7, (2, 2), (0, 0))
        wd = torch.zeros(64, 3, 7, 7, dtype=torch.float64, requires_grad=True)
        F.conv2d((img.double() * 2 - 1), wd, stride=2, padding=3).backward(dy.permute(0, 3, 1, 2).double())
        e = (dw4[:, :3].double().cpu() - wd.grad).abs().max().item() / wd.grad.abs().max().item()
        print("  rep", rep, "rel err", f"{e:.3e}", "ch3 max", float(dw4[:, 3].abs().max()))
