"""FPN neck / head (mirror of vkit_open_model/model/fpn.py) on the HIP ops."""
from typing import List, Optional, Sequence

import torch
from torch import nn

from . import helper, scripting
from .. import ops

NEAREST = 1


def build_conv1x1_block(in_channels: int, out_channels: int):
    """fpn.py:21-28"""
    return nn.Sequential(helper.permute_bchw_to_bhwc(), helper.conv1x1(in_channels, out_channels),
                         helper.ln(out_channels), helper.permute_bhwc_to_bchw(), helper.gelu())


def build_conv3x3_block(in_channels: int, out_channels: int):
    """fpn.py:31-38"""
    return nn.Sequential(helper.conv3x3(in_channels, out_channels), helper.permute_bchw_to_bhwc(),
                         helper.ln(out_channels), helper.permute_bhwc_to_bchw(), helper.gelu())


def _init_kaiming(module: nn.Module):
    for m in module.modules():  # fpn.py:104-108,185-189
        if isinstance(m, (nn.Conv2d, nn.Linear)):
            nn.init.kaiming_normal_(m.weight)
            if m.bias is not None:
                nn.init.zeros_(m.bias)


class FpnNeck(nn.Module):
    """fpn.py:51-146: 1x1 laterals to out_channels, top-down nearest add, 3x3 blocks to out/levels, nearest to
    level 0, concat."""
    _script_params: List[torch.Tensor]  # what the compiled forward hands to vkas::module_forward (model/scripting.py)
    _script_spec: str

    @classmethod
    def build_step1_conv_blocks(cls, in_channels_group: Sequence[int], out_channels: int):
        return nn.ModuleList([build_conv1x1_block(c, out_channels) for c in in_channels_group])

    @classmethod
    def build_step2_conv_blocks(cls, in_channels_group: Sequence[int], out_channels: int):
        assert out_channels % len(in_channels_group) == 0
        inner = out_channels // len(in_channels_group)
        return nn.ModuleList([build_conv3x3_block(out_channels, inner) for _ in in_channels_group])

    def __init__(self, in_channels_group: Sequence[int], out_channels: int) -> None:
        super().__init__()
        assert len(in_channels_group) > 1
        assert out_channels % len(in_channels_group) == 0  # fpn.py:75
        self.out_channels = out_channels
        self.inner_channels = out_channels // len(in_channels_group)
        self.step1_conv_blocks = self.build_step1_conv_blocks(in_channels_group, out_channels)
        self.step2_conv_blocks = self.build_step2_conv_blocks(in_channels_group, out_channels)
        self.compute_dtype = torch.bfloat16
        _init_kaiming(self)
        scripting.init_script_state(self, {'in_channels_group': [int(c) for c in in_channels_group],
                                           'out_channels': out_channels})

    def _refresh_script_spec(self):
        scripting.refresh_module_spec(self)

    def forward_act(self, feats: Sequence[torch.Tensor]) -> torch.Tensor:
        n = len(feats)
        assert n == len(self.step1_conv_blocks)
        outs = [helper.conv_block(feats[i], blk[1], blk[2]) for i, blk in enumerate(self.step1_conv_blocks)]
        for i in range(n - 1, 0, -1):
            outs[i - 1] = ops.ResizeAdd.apply(outs[i - 1], outs[i], NEAREST)
        outs = [helper.conv_block(o, blk[0], blk[2], 1, 1) for o, blk in zip(outs, self.step2_conv_blocks)]
        # widths that are not multiples of 8 (e.g. out_channels = 400, tests/test_fpn.py:16-28) take ResizeCat's compact path
        return ops.ResizeCat.apply(NEAREST, [self.inner_channels] * n, *outs)  # every level resized to the finest one and concatenated (fpn.py:131-144)

    def forward(self, features: List[torch.Tensor]) -> torch.Tensor:
        """fpn.py:110-146; scripted (tests/test_fpn.py:30): one call of vkas::module_forward."""
        if torch.jit.is_scripting():
            return torch.ops.vkas.module_forward(features, self._script_params, self._script_spec, self.training)[0]
        else:
            return self._forward_eager(features)

    @torch.jit.unused
    def _forward_eager(self, features: List[torch.Tensor]) -> torch.Tensor:
        acts = [helper.nchw_to_act(f, self.compute_dtype) for f in features]
        return helper.act_to_nchw(self.forward_act(acts), self.out_channels)

    @torch.jit.unused
    def _script_call(self, inputs: List[torch.Tensor]) -> List[torch.Tensor]:
        return [self._forward_eager(inputs)]


class FpnHead(nn.Module):
    """fpn.py:149-208"""
    _script_params: List[torch.Tensor]
    _script_spec: str

    def __init__(self, in_channels: int, out_channels: int, upsampling_factor: int = 1,
                 init_output_bias: float = 0.0):
        super().__init__()
        self.upsampling_factor = upsampling_factor
        self.out_channels = out_channels
        inner_channels = (in_channels + out_channels) // 2
        if 1 <= upsampling_factor <= 2:
            self.step1_conv = build_conv3x3_block(in_channels, inner_channels)
        else:
            # fpn.py:170-176: 5x5 smoothing for factors in (2, 4] is unreachable at the reference's defaults
            # (adaptive_scaling.py:45,47) and has no HIP kernel.
            raise NotImplementedError()
        self.step2_conv = nn.Sequential(helper.permute_bchw_to_bhwc(), helper.conv1x1(inner_channels, out_channels),
                                        helper.permute_bhwc_to_bchw())
        self.compute_dtype = torch.bfloat16
        _init_kaiming(self)
        nn.init.constant_(self.step2_conv[1].bias, init_output_bias)  # fpn.py:191
        scripting.init_script_state(self, {'in_channels': in_channels, 'out_channels': out_channels,
                                           'upsampling_factor': upsampling_factor,
                                           'init_output_bias': float(init_output_bias)})

    def _refresh_script_spec(self):
        scripting.refresh_module_spec(self)

    def conv_norm_proj(self):
        """(3x3 conv, its LayerNorm, the 1x1 projection) parameter holders."""
        return self.step1_conv[0], self.step1_conv[2], self.step2_conv[1]

    def upsample_act(self, x: torch.Tensor) -> torch.Tensor:
        if self.upsampling_factor > 1:
            f = self.upsampling_factor
            return ops.Resize.apply(x, (x.shape[1] * f, x.shape[2] * f), NEAREST)
        return x

    def forward_act(self, x: torch.Tensor, upsampled: Optional[torch.Tensor] = None) -> torch.Tensor:
        x = self.upsample_act(x) if upsampled is None else upsampled
        x = helper.conv_block(x, self.step1_conv[0], self.step1_conv[2], 1, 1)
        proj = self.step2_conv[1]
        y = ops.Conv.apply(x, proj.weight, proj.bias, 1, 0)
        return ops.ToNchw.apply(y, self.out_channels)

    def forward(self, fpn_neck_feature: torch.Tensor) -> torch.Tensor:
        if torch.jit.is_scripting():
            return torch.ops.vkas.module_forward([fpn_neck_feature], self._script_params, self._script_spec,
                                                 self.training)[0]
        else:
            return self._forward_eager(fpn_neck_feature)

    @torch.jit.unused
    def _forward_eager(self, fpn_neck_feature: torch.Tensor) -> torch.Tensor:
        return self.forward_act(helper.nchw_to_act(fpn_neck_feature, self.compute_dtype))

    @torch.jit.unused
    def _script_call(self, inputs: List[torch.Tensor]) -> List[torch.Tensor]:
        return [self._forward_eager(inputs[0])]
