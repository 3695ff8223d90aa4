"""numpy <-> torch helpers of the trainer (reference: diffuser/utils/arrays.py:11-31,39-58,96-108)."""
import numpy as np
import torch

__all__ = ["DTYPE", "torch_stack", "to_device_tp", "to_torch_tp", "number_by_ratio", "to_np", "to_torch", "report_parameters"]

DTYPE = torch.float


def torch_stack(*args, dim):
    return (torch.stack(a, dim=dim) for a in args)


def to_device_tp(*args, device):
    return tuple(a.to(device) for a in args)


def to_torch(x, dtype=None, device="cpu"):
    dtype = dtype or DTYPE
    if isinstance(x, dict):
        return {k: to_torch(v, dtype, device) for k, v in x.items()}
    if torch.is_tensor(x):
        return x.to(device).type(dtype)
    return torch.tensor(x, dtype=dtype, device=device)


def to_torch_tp(*args, dtype=torch.float32, device="cpu"):
    return (to_torch(a, dtype=dtype, device=device) for a in args)


def to_np(x):
    return x.detach().cpu().numpy() if torch.is_tensor(x) else x


def number_by_ratio(num, ratio):
    """num=10, ratio=[0.2, 0.8] -> [2, 8]: round(ratio * num) per entry; the ratios must sum to 1 and the parts to num."""
    ratio = np.array(ratio)
    assert np.isclose(sum(ratio), 1), "Ratios must sum to 1"
    out = ratio * num
    assert np.isclose(out.sum(), num)
    return np.round(out).astype(np.int32).tolist()


def report_parameters(model, topk=10):
    """Parameter count of a module and its `topk` largest tensors."""
    counts = {k: p.numel() for k, p in model.named_parameters()}
    n = sum(counts.values())
    print(f"[ utils/arrays ] Total parameters: {n / 1e6:.2f} M")
    for k in sorted(counts, key=lambda x: -counts[x])[:topk]:
        print(f"        {k}: {counts[k] / 1e6:.2f} M")
    return n
