"""UPerNext neck / head (mirror of vkit_open_model/model/upernext.py) on the HIP ops."""
from typing import List, Optional, Sequence, Tuple

import torch
from torch import nn

from . import helper, scripting
from .. import ops

BILINEAR = 0


def build_conv1x1_block(in_channels: int, out_channels: int, no_ln: bool = False):
    """upernext.py:21-36 (parameter slots 1 = Linear, 2 = LayerNorm)."""
    mods: List[nn.Module] = [helper.permute_bchw_to_bhwc(), helper.conv1x1(in_channels, out_channels)]
    if not no_ln:
        mods.append(helper.ln(out_channels))
    mods.extend([helper.permute_bhwc_to_bchw(), helper.gelu()])
    return nn.Sequential(*mods)


def build_conv3x3_block(in_channels: int, out_channels: int):
    """upernext.py:39-45 (parameter slots 0 = Conv2d, 2 = LayerNorm)."""
    return nn.Sequential(helper.conv3x3(in_channels, out_channels), helper.permute_bchw_to_bhwc(),
                         helper.ln(out_channels), helper.permute_bhwc_to_bchw(), helper.gelu())


def _init_trunc_normal(module: nn.Module):
    for m in module.modules():  # upernext.py:157-161,225-229
        if isinstance(m, (nn.Conv2d, nn.Linear)):
            nn.init.trunc_normal_(m.weight, std=0.02)
            if m.bias is not None:
                nn.init.zeros_(m.bias)


class PpmBlock(nn.Module):
    """upernext.py:48-84: pooled 1x1 branches, bilinear back to the input size, concat, 3x3 block."""
    _script_params: List[torch.Tensor]  # what the compiled forward hands to vkas::module_forward (model/scripting.py)
    _script_spec: str

    def __init__(self, ppm_scales: Sequence[int], in_channels: int, out_channels: int) -> None:
        super().__init__()
        self.ppm_scales = tuple(ppm_scales)
        self.in_channels, self.branch_channels = in_channels, out_channels
        self.ap_conv_blocks = nn.ModuleList([
            nn.Sequential(nn.AdaptiveAvgPool2d(s), build_conv1x1_block(in_channels, out_channels)) for s in ppm_scales
        ])
        self.final_conv_block = build_conv3x3_block(in_channels + len(ppm_scales) * out_channels, out_channels)
        self.compute_dtype = torch.bfloat16
        scripting.init_script_state(self, {'ppm_scales': [int(v) for v in ppm_scales], 'in_channels': in_channels,
                                           'out_channels': out_channels})

    def _refresh_script_spec(self):
        scripting.refresh_module_spec(self)

    def forward_act(self, x: torch.Tensor) -> torch.Tensor:
        pooled = ops.AdaptiveAvgPools.apply(x, *self.ppm_scales)  # one node: its backward sums the branches' gradients
        parts = [helper.conv_block(f, branch[1][1], branch[1][2]) for f, branch in zip(pooled, self.ap_conv_blocks)]
        widths = [self.in_channels] + [self.branch_channels] * len(parts)
        cat = ops.ResizeCat.apply(BILINEAR, widths, x, *parts)  # the pooled branches resized to x and concatenated behind it
        return helper.conv_block(cat, self.final_conv_block[0], self.final_conv_block[2], 1, 1)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if torch.jit.is_scripting():
            return torch.ops.vkas.module_forward([x], self._script_params, self._script_spec, self.training)[0]
        else:
            return self._forward_eager(x)

    @torch.jit.unused
    def _forward_eager(self, x: torch.Tensor) -> torch.Tensor:
        out_c = self.final_conv_block[0].out_channels
        return helper.act_to_nchw(self.forward_act(helper.nchw_to_act(x, self.compute_dtype)), out_c)

    @torch.jit.unused
    def _script_call(self, inputs: List[torch.Tensor]) -> List[torch.Tensor]:
        return [self._forward_eager(inputs[0])]


class UperNextNeck(nn.Module):
    """upernext.py:87-198"""
    _script_params: List[torch.Tensor]
    _script_spec: str

    @classmethod
    def build_step1_conv_blocks(cls, in_channels_group: Sequence[int], ppm_scales: Sequence[int], inner_channels: int):
        blocks: List[nn.Module] = [build_conv1x1_block(c, inner_channels) for c in in_channels_group[:-1]]
        blocks.append(PpmBlock(ppm_scales, in_channels_group[-1], inner_channels))
        return nn.ModuleList(blocks)

    @classmethod
    def build_step2_conv_blocks(cls, num_step1_conv_blocks: int, inner_channels: int):
        # the last level already went through the PPM's 3x3 block (upernext.py:125-133)
        return nn.ModuleList([build_conv3x3_block(inner_channels, inner_channels)
                              for _ in range(num_step1_conv_blocks - 1)])

    def __init__(self, in_channels_group: Sequence[int], out_channels: int,
                 ppm_scales: Sequence[int] = (1, 2, 3, 6)) -> None:
        super().__init__()
        assert len(in_channels_group) > 1
        assert out_channels % len(in_channels_group) == 0
        inner_channels = out_channels // len(in_channels_group)
        self.inner_channels = inner_channels
        self.out_channels = out_channels
        self.step1_conv_blocks = self.build_step1_conv_blocks(in_channels_group, ppm_scales, inner_channels)
        self.step2_conv_blocks = self.build_step2_conv_blocks(len(self.step1_conv_blocks), inner_channels)
        self.compute_dtype = torch.bfloat16
        _init_trunc_normal(self)
        scripting.init_script_state(self, {'in_channels_group': [int(c) for c in in_channels_group],
                                           'out_channels': out_channels, 'ppm_scales': [int(v) for v in ppm_scales]})

    def _refresh_script_spec(self):
        scripting.refresh_module_spec(self)

    def forward_act(self, feats: Sequence[torch.Tensor]) -> torch.Tensor:
        n = len(feats)
        assert n == len(self.step1_conv_blocks)
        outs = [helper.conv_block(feats[i], self.step1_conv_blocks[i][1], self.step1_conv_blocks[i][2])
                for i in range(n - 1)]
        outs.append(self.step1_conv_blocks[n - 1].forward_act(feats[n - 1]))
        for i in range(n - 1, 0, -1):  # top-down: outs[i-1] += bilinear(outs[i])  (upernext.py:174-182)
            outs[i - 1] = ops.ResizeAdd.apply(outs[i - 1], outs[i], BILINEAR)
        for i, blk in enumerate(self.step2_conv_blocks):
            outs[i] = helper.conv_block(outs[i], blk[0], blk[2], 1, 1)
        # every level resized to the finest one and concatenated (upernext.py:184-197): one op, the resize kernels write
        # into their channel slices
        return ops.ResizeCat.apply(BILINEAR, [self.inner_channels] * n, *outs)

    def forward(self, features: List[torch.Tensor]) -> torch.Tensor:
        """upernext.py:163-198: the backbone's NCHW features -> (B, out_channels, H/4, W/4)."""
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


class UperNextHead(nn.Module):
    """upernext.py:201-248: x`factor` bilinear -> 3x3 block to (in+out)//2 -> Linear to out_channels."""
    _script_params: List[torch.Tensor]
    _script_spec: str

    def __init__(self, in_channels: int, out_channels: int, upsampling_factor: int = 1,
                 init_output_bias: float = 0.0):
        super().__init__()
        self.upsampling_factor = upsampling_factor
        self.out_channels = out_channels
        inner_channels = (in_channels + out_channels) // 2
        self.step1_conv3x3 = build_conv3x3_block(in_channels, inner_channels)
        self.step2_conv1x1 = nn.Sequential(helper.permute_bchw_to_bhwc(), helper.conv1x1(inner_channels, out_channels),
                                           helper.permute_bhwc_to_bchw())
        self.compute_dtype = torch.bfloat16
        _init_trunc_normal(self)
        nn.init.constant_(self.step2_conv1x1[1].bias, init_output_bias)  # upernext.py:231
        scripting.init_script_state(self, {'in_channels': in_channels, 'out_channels': out_channels,
                                           'upsampling_factor': upsampling_factor,
                                           'init_output_bias': float(init_output_bias)})

    def _refresh_script_spec(self):
        scripting.refresh_module_spec(self)

    def conv_norm_proj(self):
        """(3x3 conv, its LayerNorm, the 1x1 projection) parameter holders."""
        return self.step1_conv3x3[0], self.step1_conv3x3[2], self.step2_conv1x1[1]

    def upsample_act(self, x: torch.Tensor) -> torch.Tensor:
        if self.upsampling_factor > 1:
            f = self.upsampling_factor
            return ops.Resize.apply(x, (x.shape[1] * f, x.shape[2] * f), BILINEAR)
        return x

    def forward_act(self, x: torch.Tensor, upsampled: Optional[torch.Tensor] = None) -> torch.Tensor:
        """NHWC activation in, (B, out_channels, H, W) fp32 NCHW out.  ``upsampled`` lets heads that read the
        same neck feature share one upsampled tensor (the reference recomputes it per head)."""
        x = self.upsample_act(x) if upsampled is None else upsampled
        x = helper.conv_block(x, self.step1_conv3x3[0], self.step1_conv3x3[2], 1, 1)
        proj = self.step2_conv1x1[1]
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
