"""Parameter shells of the 1-D building blocks (reference model/conv1d_components.py:7-40).  Nothing here computes: the classes only
own torch parameters under the reference's attribute names (`conv`, `block.0`, `block.1`) so that checkpoints load with strict=True;
the arithmetic runs in v2a_hip.policy_engine (implicit-GEMM conv, fused GroupNorm + Mish + FiLM kernels)."""
import torch.nn as nn


def _param_shell(name, doc, build):
    """Class factory: `build(*ctor_args)` returns {attribute: submodule}; the resulting nn.Module subclass has no forward."""

    def __init__(self, *args, **kwargs):
        nn.Module.__init__(self)
        for attr, sub in build(*args, **kwargs).items():
            setattr(self, attr, sub)

    cls = type(name, (nn.Module,), {"__init__": __init__, "__doc__": doc, "__module__": __name__})
    return cls


Downsample1d = _param_shell(
    "Downsample1d", "stride-2 Conv1d(k=3, p=1) over the horizon axis",
    lambda dim: {"conv": nn.Conv1d(dim, dim, kernel_size=3, stride=2, padding=1)})

Upsample1d = _param_shell(
    "Upsample1d", "stride-2 ConvTranspose1d(k=4, p=1) over the horizon axis",
    lambda dim: {"conv": nn.ConvTranspose1d(dim, dim, kernel_size=4, stride=2, padding=1)})

Conv1dBlock = _param_shell(
    "Conv1dBlock", "Conv1d(k, same padding) -> GroupNorm(n_groups) -> Mish; parameters live in `block.0` and `block.1`",
    lambda inp_channels, out_channels, kernel_size, n_groups=8: {
        "block": nn.Sequential(nn.Conv1d(inp_channels, out_channels, kernel_size, padding=kernel_size // 2),
                               nn.GroupNorm(n_groups, out_channels), nn.Mish())})
