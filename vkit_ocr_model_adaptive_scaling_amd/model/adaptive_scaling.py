"""AdaptiveScaling model assembly (mirror of vkit_open_model/model/adaptive_scaling.py) on the HIP ops."""
import logging
from enum import Enum, unique
from typing import Dict, List, Mapping, Optional, Sequence, Tuple

import attrs
import os

import torch
from torch import nn
from torch.nn import functional as F

from .convnext import ConvNext
from .fpn import FpnNeck, FpnHead
from .upernext import UperNextNeck, UperNextHead
from .helper import set_compute_dtype
from . import scripting
from .. import ops

logger = logging.getLogger(__name__)


@unique
class AdaptiveScalingSize(Enum):
    TINY = 'tiny'
    SMALL = 'small'
    BASE = 'base'
    LARGE = 'large'


@unique
class AdaptiveScalingNeckHeadType(Enum):
    FPN = 'fpn'
    UPERNEXT = 'upernext'


@attrs.define
class AdaptiveScalingConfig:
    """adaptive_scaling.py:41-48 (same fields, same defaults)."""
    size: AdaptiveScalingSize = AdaptiveScalingSize.SMALL
    neck_head_type: AdaptiveScalingNeckHeadType = AdaptiveScalingNeckHeadType.FPN
    rough_upsampling_factor: int = 2
    rough_init_char_height_output_bias: float = 8.0
    precise_upsampling_factor: int = 2
    precise_enable_char_mask_head: bool = False


class _SoftplusSlot(nn.Module):
    """Occupies index 1 of the Softplus-wrapped heads (adaptive_scaling.py:93-102,133-141); applied via ops.Softplus."""

    @torch.jit.unused
    def forward(self, x):
        return ops.Softplus.apply(x)


_TWO_STREAMS = os.environ.get('VKAS_TWO_STREAMS', '0') == '1'
_SIDE = {}


def _side_stream(device):
    if device not in _SIDE:
        _SIDE[device] = torch.cuda.Stream(device=device)
    return _SIDE[device]


_BACKBONES = {
    AdaptiveScalingSize.TINY: ConvNext.create_tiny,
    AdaptiveScalingSize.SMALL: ConvNext.create_small,
    AdaptiveScalingSize.BASE: ConvNext.create_base,
    AdaptiveScalingSize.LARGE: ConvNext.create_large,
}


class AdaptiveScaling(nn.Module):
    # what torch.jit.script(model) compiles forward_rough / forward_precise against (model/scripting.py): the parameter
    # tensors in named_parameters() order and the construction recipe
    _script_params: List[torch.Tensor]
    _script_spec: str

    def __init__(self, config: AdaptiveScalingConfig, compute_dtype: torch.dtype = torch.bfloat16):
        super().__init__()
        if config.size not in _BACKBONES:
            raise NotImplementedError()
        self.backbone = _BACKBONES[config.size]()
        if config.neck_head_type == AdaptiveScalingNeckHeadType.FPN:
            neck_cls, head_cls = FpnNeck, FpnHead
        elif config.neck_head_type == AdaptiveScalingNeckHeadType.UPERNEXT:
            neck_cls, head_cls = UperNextNeck, UperNextHead
        else:
            raise NotImplementedError()
        width = self.backbone.in_channels_group[-2]  # adaptive_scaling.py:79

        def head(out_channels, factor, bias=0.0):
            return head_cls(in_channels=width, out_channels=out_channels, upsampling_factor=factor,
                            init_output_bias=bias)

        rf, pf = config.rough_upsampling_factor, config.precise_upsampling_factor
        self.rough_neck = neck_cls(in_channels_group=self.backbone.in_channels_group, out_channels=width)
        self.rough_char_mask_head = head(1, rf)
        self.rough_char_height_head = nn.Sequential(head(1, rf, config.rough_init_char_height_output_bias),
                                                    _SoftplusSlot())
        self.precise_neck = neck_cls(in_channels_group=self.backbone.in_channels_group, out_channels=width)
        self.precise_char_mask_head = head(1, pf) if config.precise_enable_char_mask_head else None
        self.precise_char_prob_head = head(1, pf)
        self.precise_char_up_left_corner_offset_head = head(2, pf)
        self.precise_char_corner_angle_head = head(4, pf)
        self.precise_char_corner_distance_head = nn.Sequential(head(4, pf), _SoftplusSlot())
        self.config = config
        self.compute_dtype = compute_dtype
        self._script_params = [p for _, p in self.named_parameters()]
        self._script_spec = ''
        set_compute_dtype(self, compute_dtype)

    def set_compute_dtype(self, dtype: torch.dtype):
        return set_compute_dtype(self, dtype)

    def _refresh_script_spec(self):
        self._script_spec = scripting.make_spec(self.config, self.compute_dtype)

    @torch.jit.unused
    def _run_heads(self, neck_feature: torch.Tensor, heads: Sequence[nn.Module], label_points=None, n_dense: int = 1):
        """All heads of a pass read the same neck feature (adaptive_scaling.py:150-152,163-170): upsample it once and
        run their 3x3 convolutions as ONE implicit GEMM (output channels of the heads side by side, each padded to a
        multiple of 8), then per-head LayerNorm+GELU on channel slices, projection, NCHW, Softplus."""
        plain = [h[0] if isinstance(h, nn.Sequential) else h for h in heads]
        up = plain[0].upsample_act(neck_feature)
        convs, norms, projs = zip(*[h.conv_norm_proj() for h in plain])
        if ops.HeadsFused.eligible(up, [c.out_channels for c in convs], [hp.out_channels for hp in plain]):
            # LayerNorm + GELU + projection run in the conv's epilogue: the per-head activations never reach HBM; the
            # heads' weights are packed side by side by the pack kernel (no torch.cat of parameters on the hot path)
            fused = []
            for cv, nm, proj in zip(convs, norms, projs):
                fused.extend([cv.weight, cv.bias, nm.weight, nm.bias, proj.weight, proj.bias])
            if label_points is not None and torch.is_grad_enabled() and 0 < n_dense < len(heads):
                # opt-in (TwoPassStep(label_point_forward=True)): the heads behind the first n_dense are evaluated at the
                # label points only - their maps are zero elsewhere, which is all a label-point loss reads
                py, px = label_points
                ys = (ops.HeadsFused.apply(up, True, *fused[:6 * n_dense])
                      + ops.HeadsAtPoints.apply(up, py, px, *fused[6 * n_dense:]))
            else:
                ys = ops.HeadsFused.apply(up, torch.is_grad_enabled(), *fused)
            outs = []
            for h, hp, y in zip(heads, plain, ys):
                y = ops.ToNchw.apply(y, hp.out_channels)
                outs.append(h[1](y) if isinstance(h, nn.Sequential) else y)
            return tuple(outs)
        w_parts, b_parts = [], []
        for c in convs:
            pad = ops.rup8(c.out_channels) - c.out_channels
            w_parts.append(F.pad(c.weight, (0, 0, 0, 0, 0, 0, 0, pad)) if pad else c.weight)
            b_parts.append(F.pad(c.bias, (0, pad)) if pad else c.bias)
        w_cat, b_cat = torch.cat(w_parts, 0), torch.cat(b_parts, 0)
        z = ops.Conv.apply(up, w_cat, b_cat, 1, 1)
        affine = []
        for nm in norms:
            affine.extend([nm.weight, nm.bias])
        acts = ops.MultiLayerNorm.apply(z, True, *affine)
        outs = []
        for h, hp, a, proj in zip(heads, plain, acts, projs):
            y = ops.ToNchw.apply(ops.Conv.apply(a, proj.weight, proj.bias, 1, 0), hp.out_channels)
            outs.append(h[1](y) if isinstance(h, nn.Sequential) else y)
        return tuple(outs)

    @torch.jit.export
    def forward_rough(self, x: torch.Tensor,
                      drop_masks: Optional[List[Optional[torch.Tensor]]] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """adaptive_scaling.py:143-154.  Scripted (train.py:278): one call of vkas::adaptive_scaling_forward, which runs
        _forward_rough_eager on the same parameter tensors (model/scripting.py)."""
        if torch.jit.is_scripting():
            assert drop_masks is None, 'explicit stochastic-depth masks are an eager-mode extension'
            outs = torch.ops.vkas.adaptive_scaling_forward(x, self._script_params, self._script_spec, 0, self.training)
            return outs[0], outs[1]
        else:
            return self._forward_rough_eager(x, drop_masks)

    @torch.jit.export
    def forward_precise(self, x: torch.Tensor, drop_masks: Optional[List[Optional[torch.Tensor]]] = None,
                        label_points: Optional[Tuple[torch.Tensor, torch.Tensor]] = None
                        ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
        """adaptive_scaling.py:156-177; see forward_rough and _forward_precise_eager."""
        if torch.jit.is_scripting():
            assert drop_masks is None and label_points is None, 'eager-mode extensions'
            outs = torch.ops.vkas.adaptive_scaling_forward(x, self._script_params, self._script_spec, 1, self.training)
            return outs[0], outs[1], outs[2], outs[3]
        else:
            return self._forward_precise_eager(x, drop_masks, label_points)

    @torch.jit.unused
    def _forward_rough_eager(self, x: torch.Tensor, drop_masks=None) -> Tuple[torch.Tensor, torch.Tensor]:
        """adaptive_scaling.py:143-154"""
        feats = self.backbone.forward_act(x, drop_masks)
        neck = self.rough_neck.forward_act(feats)
        return self._run_heads(neck, (self.rough_char_mask_head, self.rough_char_height_head))  # type: ignore

    @torch.jit.unused
    def _forward_precise_eager(self, x: torch.Tensor, drop_masks=None,
                               label_points=None) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
        """adaptive_scaling.py:156-177.  ``label_points`` = (y, x) (B,P) int64 is an extension for training steps only (see
        ops.HeadsAtPoints): the offset / angle / distance maps are then valid at those points and zero elsewhere."""
        feats = self.backbone.forward_act(x, drop_masks)
        neck = self.precise_neck.forward_act(feats)
        return self._run_heads(neck, (self.precise_char_prob_head, self.precise_char_up_left_corner_offset_head,
                                      self.precise_char_corner_angle_head,
                                      self.precise_char_corner_distance_head), label_points)  # type: ignore

    @torch.jit.unused
    def forward_both(self, x_rough: torch.Tensor, x_precise: torch.Tensor, drop_masks=None, precise_label_points=None):
        """forward_rough(x_rough) and forward_precise(x_precise) with ONE backbone pass over the concatenated batch.
        The reference's step (train.py:397-478) runs the two passes back to back and lets the gradients accumulate;
        no layer couples samples of a batch (LayerNorm is per pixel, stochastic depth per sample), so the outputs are
        those of the two separate calls and d(loss_r + loss_p) equals the accumulated gradient.  Halves the number of
        backbone launches and doubles their size (the small stage-3/4 GEMMs fill the 256 CUs better)."""
        b0 = x_rough.shape[0]
        feats = self.backbone.forward_act((x_rough, x_precise), drop_masks)  # no torch.cat of the images: ops.images_to_act
        halves = [ops.SplitBatch.apply(f, b0) for f in feats]

        def rough_branch():
            return self._run_heads(self.rough_neck.forward_act([h[0] for h in halves]),
                                   (self.rough_char_mask_head, self.rough_char_height_head))

        def precise_branch():
            return self._run_heads(self.precise_neck.forward_act([h[1] for h in halves]),
                                   (self.precise_char_prob_head, self.precise_char_up_left_corner_offset_head,
                                    self.precise_char_corner_angle_head, self.precise_char_corner_distance_head),
                                   precise_label_points)
        if not _TWO_STREAMS:
            return rough_branch(), precise_branch()
        # experiment (VKAS_TWO_STREAMS=1): the two neck + head branches are independent until the losses are summed
        main = torch.cuda.current_stream()
        side = _side_stream(x_rough.device)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            precise = precise_branch()
        rough = rough_branch()
        main.wait_stream(side)
        for t in precise:
            t.record_stream(main)
        return rough, precise

    # ---- gradient inspection helpers (adaptive_scaling.py:179-237) ---------------------------------------
    @classmethod
    def debug_get_rough_name_to_grad(cls, model: nn.Module) -> Dict[str, torch.Tensor]:
        return {n: p.grad.cpu().clone() for n, p in model.named_parameters() if p.grad is not None}

    @classmethod
    def debug_get_precise_name_to_grad(cls, model: nn.Module, rough_name_to_grad: Mapping[str, torch.Tensor]):
        return {n: p.grad.cpu() - rough_name_to_grad[n] for n, p in model.named_parameters()
                if p.grad is not None and n in rough_name_to_grad}

    @classmethod
    def debug_inspect_name_to_grad(cls, rough_name_to_grad: Mapping[str, torch.Tensor],
                                   precise_name_to_grad: Mapping[str, torch.Tensor]):
        names = sorted(set(rough_name_to_grad) & set(precise_name_to_grad))
        stats = {}
        for tag, table in (('rough', rough_name_to_grad), ('precise', precise_name_to_grad)):
            g = torch.abs(torch.cat([table[n].reshape(-1) for n in names]))
            stats[tag] = (float(g.mean()), float(g.std()))
            logger.info(f'{tag}_abs_grads_mean = {stats[tag][0]}, {tag}_abs_grads_std = {stats[tag][1]}')
        logger.info(f'rough_abs_grads_mean / precise_abs_grads_mean = {stats["rough"][0] / (stats["precise"][0] + 1E-15)}')
        return stats
