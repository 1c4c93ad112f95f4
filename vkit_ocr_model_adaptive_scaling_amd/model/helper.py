"""Building blocks shared by the backbone, necks and heads (mirror of vkit_open_model/model/helper.py).

The reference's factories return stock torch modules that *compute*; here the same names return
parameter holders with the same parameter names / shapes / default initialisation (so state dicts are
interchangeable), while all arithmetic goes through the HIP ops in ``..ops``.  The layout
permutes of the reference (helper.py:76-93) do not exist: activations are NHWC end to end; ``Slot`` keeps
their positions inside ``nn.Sequential`` containers so that parameter indices (``block.0``, ``block.2``,
``block.3``, ``block.5`` ...) stay identical.
"""
from typing import Optional, Sequence

import torch
from torch import nn
from torch.nn import functional as F

from .. import ops


class Slot(nn.Module):
    """Parameter-free placeholder occupying the index a permute / GELU module has in the reference."""

    def __init__(self, what: str = ''):
        super().__init__()
        self.what = what

    def extra_repr(self):
        return self.what

    @torch.jit.unused
    def forward(self, x):  # never part of the compute path
        raise RuntimeError('Slot modules only keep parameter indices aligned with the reference')


def conv1x1(in_channels: int, out_channels: int):
    """helper.py:18-22 — nn.Linear(in, out) applied on the channel axis."""
    return nn.Linear(in_channels, out_channels)


def conv3x3(in_channels: int, out_channels: int):
    """helper.py:25-31"""
    return nn.Conv2d(in_channels, out_channels, kernel_size=3, padding=1)


def conv5x5(in_channels: int, out_channels: int):
    """helper.py:34-40 (only reachable with upsampling_factor in (2, 4], fpn.py:170-174; not on the HIP path)."""
    return nn.Conv2d(in_channels, out_channels, kernel_size=5, padding=2)


def pconv2x2(in_channels: int, out_channels: int):
    """helper.py:43-49"""
    return nn.Conv2d(in_channels, out_channels, kernel_size=2, stride=2)


def pconv4x4(in_channels: int, out_channels: int):
    """helper.py:52-58"""
    return nn.Conv2d(in_channels, out_channels, kernel_size=4, stride=4)


def dconv7x7(in_channels: int, out_channels: Optional[int] = None):
    """helper.py:61-73"""
    if out_channels is None:
        out_channels = in_channels
    else:
        assert in_channels % out_channels == 0
    return nn.Conv2d(in_channels, out_channels, kernel_size=7, padding=3, groups=in_channels)


def ln(in_channels: int):
    """helper.py:96-97"""
    return nn.LayerNorm(in_channels, eps=1E-6)


def gelu():
    """helper.py:100-101 (fused into the producing kernel here)."""
    return Slot('gelu')


def permute_bchw_to_bhwc():
    return Slot('bchw->bhwc')


def permute_bhwc_to_bchw():
    return Slot('bhwc->bchw')


class Permutation(Slot):
    """helper.py:76-86 — kept for API compatibility; a no-op marker in the NHWC-native design."""

    def __init__(self, dims: Sequence[int]):
        super().__init__(f'permute{tuple(dims)}')
        self.dims = tuple(dims)


# ---------------------------------------------------------------------------------------------------
# NCHW <-> NHWC activation at the module API boundary (zero-copy whenever the caller hands over a
# channels-last tensor with a multiple of 8 channels, which is what this package's own modules produce)
# ---------------------------------------------------------------------------------------------------
def nchw_to_act(x: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    if x.dim() != 4:
        raise ValueError(f'expected a (B,C,H,W) tensor, got {tuple(x.shape)}')
    if not x.is_cuda:
        raise RuntimeError('this model runs on the MI355X only; move the input to the GPU (no CPU fallback)')
    a = x.permute(0, 2, 3, 1)
    c = a.shape[3]
    if c % 8 != 0:
        a = F.pad(a, (0, 8 - c % 8))
    if a.dtype != dtype:
        a = a.to(dtype)
    return a if ops.act_ok(a) else a.contiguous()


def act_to_nchw(a: torch.Tensor, channels: int) -> torch.Tensor:
    if a.shape[3] != channels:
        a = a[..., :channels]
    return a.permute(0, 3, 1, 2)


def conv_block(x, conv: nn.Module, norm: nn.LayerNorm, stride: int = 1, pad: int = 0):
    """Conv/Linear -> LN -> GELU: upernext.py:21-45, fpn.py:21-48."""
    y = ops.Conv.apply(x, conv.weight, conv.bias, stride, pad)
    return ops.LayerNorm.apply(y, norm.weight, norm.bias, True)


def set_compute_dtype(module: nn.Module, dtype: torch.dtype) -> nn.Module:
    """Select the activation storage type of every module of this package below ``module``:
    torch.bfloat16 (MFMA path, default), torch.float16 (MFMA path; the reference's fp16 inference, BASELINE.json
    configs[4]) or torch.float32 (exact-fp32 parity mode)."""
    if dtype not in (torch.bfloat16, torch.float16, torch.float32):
        raise ValueError('compute dtype must be torch.bfloat16, torch.float16 or torch.float32')
    for m in module.modules():
        if hasattr(m, 'compute_dtype'):
            m.compute_dtype = dtype
        if hasattr(m, '_refresh_script_spec'):  # AdaptiveScaling: the recipe torch.jit.script(model) bakes in
            m._refresh_script_spec()
    return module
