import torch


def wsum(sd):
    """Order-independent weight checksum used by tools/make_golden.py (float64 sum of |w|)."""
    return float(sum(v.double().abs().sum() for k, v in sorted(sd.items()) if torch.is_floating_point(v)))
