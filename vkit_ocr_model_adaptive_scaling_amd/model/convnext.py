"""ConvNeXt backbone (mirror of vkit_open_model/model/convnext.py) on the HIP ops.

Same constructor signatures, attribute names and state-dict keys as the reference; the forward path is
NHWC-native: one fused autograd op per block layer (ops.ConvNextLayer: dw7x7, LN, two MFMA GEMMs with the
GELU and layer-scale/stochastic-depth/residual epilogues), LN and patchify-conv GEMMs between stages.
"""
from typing import List, Optional, Sequence, Tuple

import torch
from torch import nn

from . import helper, scripting
from .. import ops


class ConvNextBlockLayer(nn.Module):
    """convnext.py:20-59"""
    _script_params: List[torch.Tensor]  # what the compiled forward hands to vkas::module_forward (model/scripting.py)
    _script_spec: str

    def __init__(self, in_channels: int, prob_bypass: float = 0.0) -> None:
        super().__init__()
        self.block = nn.Sequential(
            helper.dconv7x7(in_channels),
            helper.permute_bchw_to_bhwc(),
            helper.ln(in_channels),
            helper.conv1x1(in_channels, 4 * in_channels),
            helper.gelu(),
            helper.conv1x1(4 * in_channels, in_channels),
            helper.permute_bhwc_to_bchw(),
        )
        self.block_scale = nn.Parameter(torch.full((in_channels, 1, 1), 1E-6))
        self.prob_bypass = prob_bypass
        self.compute_dtype = torch.bfloat16
        scripting.init_script_state(self, {'in_channels': in_channels, 'prob_bypass': prob_bypass})

    def _refresh_script_spec(self):
        scripting.refresh_module_spec(self)

    def stochastic_depth_mask(self, batch: int, device) -> Optional[torch.Tensor]:
        """convnext.py:41-53: per-sample Bernoulli(keep) / keep in training mode, None otherwise."""
        if not self.training or self.prob_bypass == 0.0:
            return None
        keep = 1.0 - self.prob_bypass
        mask = torch.empty((batch,), dtype=torch.float32, device=device).bernoulli_(keep)
        if keep > 0.0:
            mask.div_(keep)
        return mask

    def forward_act(self, x: torch.Tensor, mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        if mask is None:
            mask = self.stochastic_depth_mask(x.shape[0], x.device)
        dw, norm, fc1, fc2 = self.block[0], self.block[2], self.block[3], self.block[5]
        return ops.ConvNextLayer.apply(x, dw.weight, dw.bias, norm.weight, norm.bias, fc1.weight, fc1.bias,
                                       fc2.weight, fc2.bias, self.block_scale, mask, torch.is_grad_enabled())

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """NCHW in, NCHW out (convnext.py:55-59); scripted: one call of vkas::module_forward (model/scripting.py)."""
        if torch.jit.is_scripting():
            return torch.ops.vkas.module_forward([x], self._script_params, self._script_spec, self.training)[0]
        else:
            return self._forward_eager(x)

    @torch.jit.unused
    def _forward_eager(self, x: torch.Tensor) -> torch.Tensor:
        c = x.shape[1]
        return helper.act_to_nchw(self.forward_act(helper.nchw_to_act(x, self.compute_dtype)), c)

    @torch.jit.unused
    def _script_call(self, inputs: List[torch.Tensor]) -> List[torch.Tensor]:
        return [self._forward_eager(inputs[0])]


class ConvNextBlock(nn.Module):
    """convnext.py:62-101: N layers, an extra LayerNorm whose output is the emitted feature, optional 2x2/2 conv."""
    _script_params: List[torch.Tensor]
    _script_spec: str

    def __init__(self, layer_idx_begin: int, layer_idx_end: int, in_channels: int, num_layers: int,
                 out_channels: Optional[int]) -> None:
        super().__init__()
        self.layers = nn.Sequential(*[
            ConvNextBlockLayer(in_channels, prob_bypass=0.1 * (layer_idx_begin + i) / layer_idx_end)
            for i in range(num_layers)
        ])
        self.ln = nn.Sequential(helper.permute_bchw_to_bhwc(), helper.ln(in_channels), helper.permute_bhwc_to_bchw())
        self.pconv2x2: Optional[nn.Module] = None
        if out_channels:
            self.pconv2x2 = helper.pconv2x2(in_channels, out_channels)
        self.compute_dtype = torch.bfloat16
        scripting.init_script_state(self, {'layer_idx_begin': layer_idx_begin, 'layer_idx_end': layer_idx_end,
                                           'in_channels': in_channels, 'num_layers': num_layers,
                                           'out_channels': out_channels})

    def _refresh_script_spec(self):
        scripting.refresh_module_spec(self)

    def forward_act(self, x: torch.Tensor, masks: Optional[Sequence[Optional[torch.Tensor]]] = None):
        for i, layer in enumerate(self.layers):
            x = layer.forward_act(x, None if masks is None else masks[i])
        feature = ops.LayerNorm.apply(x, self.ln[1].weight, self.ln[1].bias, False)
        x = feature
        if self.pconv2x2 is not None:
            x = ops.Conv.apply(feature, self.pconv2x2.weight, self.pconv2x2.bias, 2, 0)
        return feature, x

    def forward(self, x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """convnext.py:93-101: (emitted feature, input of the next stage), NCHW."""
        if torch.jit.is_scripting():
            outs = torch.ops.vkas.module_forward([x], self._script_params, self._script_spec, self.training)
            return outs[0], outs[1]
        else:
            return self._forward_eager(x)

    @torch.jit.unused
    def _forward_eager(self, x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        c = x.shape[1]
        feature, y = self.forward_act(helper.nchw_to_act(x, self.compute_dtype))
        c_out = self.pconv2x2.out_channels if self.pconv2x2 is not None else c
        return helper.act_to_nchw(feature, c), helper.act_to_nchw(y, c_out)

    @torch.jit.unused
    def _script_call(self, inputs: List[torch.Tensor]) -> List[torch.Tensor]:
        return list(self._forward_eager(inputs[0]))


class ConvNext(nn.Module):
    """convnext.py:104-235"""
    _script_params: List[torch.Tensor]
    _script_spec: str

    @classmethod
    def build_stem(cls, stem_in_channels: int, block_in_channels: int, use_pconv2x2: bool):
        pconv = (helper.pconv2x2 if use_pconv2x2 else helper.pconv4x4)(stem_in_channels, block_in_channels)
        return nn.Sequential(pconv, helper.permute_bchw_to_bhwc(), helper.ln(block_in_channels),
                             helper.permute_bhwc_to_bchw())

    @classmethod
    def build_blocks(cls, block_in_channels_and_num_layers: Sequence[Tuple[int, int]]):
        plan = list(block_in_channels_and_num_layers)
        total = sum(n for _, n in plan)
        blocks: List[ConvNextBlock] = []
        begin = 0
        for idx, (channels, num_layers) in enumerate(plan):
            nxt = plan[idx + 1][0] if idx + 1 < len(plan) else None
            blocks.append(ConvNextBlock(begin, total - 1, channels, num_layers, nxt))
            begin += num_layers
        return nn.ModuleList(blocks), [c for c, _ in plan]

    def __init__(self, stem_in_channels: int, block_in_channels_and_num_layers: Sequence[Tuple[int, int]],
                 stem_use_pconv2x2: bool):
        super().__init__()
        if stem_in_channels > 8:
            raise NotImplementedError('the HIP stem packs the image into 8 channels')
        self.stem = self.build_stem(stem_in_channels, block_in_channels_and_num_layers[0][0], stem_use_pconv2x2)
        self.blocks, self.in_channels_group = self.build_blocks(block_in_channels_and_num_layers)
        self.compute_dtype = torch.bfloat16
        for module in self.modules():  # convnext.py:169-173
            if isinstance(module, (nn.Conv2d, nn.Linear)):
                nn.init.trunc_normal_(module.weight, std=0.02)
                if module.bias is not None:
                    nn.init.zeros_(module.bias)
        scripting.init_script_state(self, {'stem_in_channels': stem_in_channels,
                                           'block_in_channels_and_num_layers':
                                               [[int(c), int(n)] for c, n in block_in_channels_and_num_layers],
                                           'stem_use_pconv2x2': bool(stem_use_pconv2x2)})

    def _refresh_script_spec(self):
        scripting.refresh_module_spec(self)

    @classmethod
    def create_tiny(cls, stem_use_pconv2x2: bool = False):
        return ConvNext(3, ((96, 3), (192, 3), (384, 9), (768, 3)), stem_use_pconv2x2)

    @classmethod
    def create_small(cls, stem_use_pconv2x2: bool = False):
        return ConvNext(3, ((96, 3), (192, 3), (384, 27), (768, 3)), stem_use_pconv2x2)

    @classmethod
    def create_base(cls, stem_use_pconv2x2: bool = False):
        return ConvNext(3, ((128, 3), (256, 3), (512, 27), (1024, 3)), stem_use_pconv2x2)

    @classmethod
    def create_large(cls, stem_use_pconv2x2: bool = False):
        return ConvNext(3, ((192, 3), (384, 3), (768, 27), (1536, 3)), stem_use_pconv2x2)

    def draw_stochastic_depth_masks(self, batch: int, device) -> Optional[List[Optional[torch.Tensor]]]:
        """convnext.py:41-53 for every block layer at once: per-sample Bernoulli(keep) / keep in training mode (one
        random draw for the whole backbone instead of one per layer), None in eval mode / for layers that never drop."""
        layers = [layer for block in self.blocks for layer in block.layers]
        if not self.training or all(layer.prob_bypass == 0.0 for layer in layers):
            return None
        keep = getattr(self, '_keep_probs', None)
        if keep is None or keep.device != device or keep.numel() != len(layers):
            keep = torch.tensor([1.0 - layer.prob_bypass for layer in layers], dtype=torch.float32, device=device)
            self._keep_probs = keep
            self._inv_keep = 1.0 / keep[:, None].clamp_min(1e-12)
        # {0, 1} * (1 / keep) is bit for bit {0, 1} / keep: three launches per step instead of five
        m = (torch.rand((len(layers), batch), dtype=torch.float32, device=device) < keep[:, None]) * self._inv_keep
        return [None if layer.prob_bypass == 0.0 else m[i] for i, layer in enumerate(layers)]

    def forward_act(self, x: torch.Tensor, masks: Optional[Sequence[Optional[torch.Tensor]]] = None):
        """x: (B, 3, H, W) raw pixels (fp32 NCHW), or a tuple of such batches of one spatial size (run as one batch)
        -> list of NHWC activations at /4, /8, /16, /32.
        ``masks``: optional per-layer stochastic-depth keep masks (one (B,) tensor or None per block layer)."""
        parts = list(x) if isinstance(x, (tuple, list)) else None  # several image batches run as one (forward_both)
        batch = sum(t.shape[0] for t in parts) if parts is not None else x.shape[0]
        device = parts[0].device if parts is not None else x.device
        if masks is None:
            masks = self.draw_stochastic_depth_masks(batch, device)
        pconv, norm = self.stem[0], self.stem[2]
        k = pconv.kernel_size[0]
        want_dx = any(t.requires_grad for t in parts) if parts is not None else x.requires_grad
        a = ops.images_to_act(parts, self.compute_dtype) if parts is not None else ops.ImageToAct.apply(x, self.compute_dtype)
        # the gradient w.r.t. the image exists only when the caller asks for it (x.requires_grad): training feeds data
        a = ops.Conv.apply(a, pconv.weight, pconv.bias, k, 0, bool(want_dx and torch.is_grad_enabled()))
        a = ops.LayerNorm.apply(a, norm.weight, norm.bias, False)
        feats = []
        li = 0
        for block in self.blocks:
            n = len(block.layers)
            feature, a = block.forward_act(a, None if masks is None else masks[li:li + n])
            li += n
            feats.append(feature)
        return feats

    def forward(self, x: torch.Tensor) -> List[torch.Tensor]:
        """convnext.py:216-235: (B,3,H,W) -> the four stage features, NCHW.  Scripted (tests/test_convnext.py:53-63 scripts
        and calls the backbone on its own): one call of vkas::module_forward (model/scripting.py)."""
        if torch.jit.is_scripting():
            return torch.ops.vkas.module_forward([x], self._script_params, self._script_spec, self.training)
        else:
            return self._forward_eager(x)

    @torch.jit.unused
    def _forward_eager(self, x: torch.Tensor) -> List[torch.Tensor]:
        feats = self.forward_act(x)
        return [helper.act_to_nchw(f, c) for f, c in zip(feats, self.in_channels_group)]

    @torch.jit.unused
    def _script_call(self, inputs: List[torch.Tensor]) -> List[torch.Tensor]:
        return self._forward_eager(inputs[0])
