"""Rough / precise losses of the adaptive-scaling model (mirror of
vkit_open_model/loss_function/adaptive_scaling.py) as single fused HIP ops.

Same config classes (including the reference's spelling ``...LossFunctionConifg``), same call signatures.
Only the terms that are active under the reference's default factors have HIP kernels: focal + dice + masked
log-space smooth-L1 (rough); masked L2 x2, smooth-L1 offsets, distance regulariser, soft-target cross entropy,
corner distances (precise).  Enabling one of the default-off terms (weighted BCE, precise mask focal, prob L1,
weight-adaptive heatmap regression) raises NotImplementedError instead of silently computing something else.
"""
from typing import Any, Optional, Tuple

import attrs
import torch

from .. import ops
from .._lib import RoughLossCfg, PreciseLossCfg


@attrs.define
class Box:
    """Stand-in for ``vkit.element.Box`` (a third-party record the reference only reads ``up/down/left/right``
    from, inclusive bounds: loss_function/adaptive_scaling.py:77-86).  Any object with these attributes works."""
    up: int
    down: int
    left: int
    right: int


@attrs.define
class AdaptiveScalingRoughLossFunctionConifg:
    bce_negative_ratio: float = 3.0
    bce_factor: float = 0.0
    focal_factor: float = 5.0
    dice_factor: float = 1.0
    l1_factor: float = 1.0
    downsampled_score_map_min: float = 1.1
    char_height_feature_min: float = 1.1


class AdaptiveScalingRoughLossFunction:
    """loss_function/adaptive_scaling.py:38-131"""

    def __init__(self, config: AdaptiveScalingRoughLossFunctionConifg, focal_alpha: float = 0.25,
                 focal_gamma: float = 2.0):
        if config.bce_factor > 0.0:
            raise NotImplementedError('weighted BCE (bce_factor > 0) is off by default and has no HIP kernel')
        self.config = config
        self.focal_alpha = focal_alpha  # focal_with_logits.py:21-23
        self.focal_gamma = focal_gamma

    def __call__(self, rough_char_mask_feature: torch.Tensor, rough_char_height_feature: torch.Tensor,
                 downsampled_mask: torch.Tensor, downsampled_score_map: torch.Tensor,
                 downsampled_shape: Tuple[int, int], downsampled_core_box: Any, scale: float = 1.0) -> torch.Tensor:
        assert rough_char_mask_feature.shape == rough_char_height_feature.shape
        assert tuple(rough_char_mask_feature.shape[1:]) == (1, *downsampled_shape)
        box = downsampled_core_box
        assert tuple(downsampled_mask.shape[1:]) == (box.down - box.up + 1, box.right - box.left + 1)
        c = self.config
        cfg = RoughLossCfg(c.focal_factor, c.dice_factor, c.l1_factor, c.downsampled_score_map_min,
                           c.char_height_feature_min, self.focal_alpha, self.focal_gamma, scale)
        return ops.RoughLoss.apply(rough_char_mask_feature, rough_char_height_feature, downsampled_mask,
                                   downsampled_score_map, int(box.up), int(box.left), cfg)


@attrs.define
class AdaptiveScalingPreciseLossFunctionConifg:
    char_mask_focal_factor: float = 0.0
    char_prob_l1_factor: float = 0.0
    char_prob_pos_l2_factor: float = 2.0
    char_prob_neg_l2_factor: float = 1.0
    char_prob_wahr_factor: float = 0.0
    char_up_left_offset_l1_factor: float = 1.0
    char_up_left_distance_regulation_l1_factor: float = 1.0
    char_corner_angle_cross_entropy_factor: float = 5.0
    char_corner_distance_l1_factor: float = 1.0
    loss_factor: float = 0.15


class AdaptiveScalingPreciseLossFunction:
    """loss_function/adaptive_scaling.py:148-346"""

    def __init__(self, config: AdaptiveScalingPreciseLossFunctionConifg, smooth_beta: float = 2.5):
        if config.char_mask_focal_factor > 0 or config.char_prob_l1_factor > 0 or config.char_prob_wahr_factor > 0:
            raise NotImplementedError('mask focal / prob L1 / WAHR terms are off by default and have no HIP kernel')
        self.config = config
        self.smooth_beta = smooth_beta  # :159-165

    @classmethod
    def get_label_point_feature(cls, feature: torch.Tensor, label_point_y: torch.Tensor, label_point_x: torch.Tensor):
        """(B, C, H, W) -> (B, P, C) (loss_function/adaptive_scaling.py:167-179); host-side helper for inspection —
        the training path gathers inside the fused kernel."""
        batch = feature.shape[0]
        assert batch == label_point_y.shape[0] == label_point_x.shape[0]
        idx = torch.arange(batch, device=feature.device)[:, None]
        return feature[idx, :, label_point_y, label_point_x]

    def __call__(self, precise_char_mask_feature: Optional[torch.Tensor], precise_char_prob_feature: torch.Tensor,
                 precise_char_up_left_corner_offset_feature: torch.Tensor,
                 precise_char_corner_angle_feature: torch.Tensor, precise_char_corner_distance_feature: torch.Tensor,
                 downsampled_char_prob_score_map: torch.Tensor, downsampled_char_mask: torch.Tensor,
                 downsampled_shape: Tuple[int, int], downsampled_core_box: Any,
                 downsampled_label_point_y: torch.Tensor, downsampled_label_point_x: torch.Tensor,
                 char_up_left_offsets: torch.Tensor, char_corner_angles: torch.Tensor,
                 char_corner_distances: torch.Tensor, scale: float = 1.0) -> torch.Tensor:
        assert tuple(precise_char_prob_feature.shape[1:]) == (1, *downsampled_shape)
        box = downsampled_core_box
        assert tuple(downsampled_char_mask.shape[1:]) == (box.down - box.up + 1, box.right - box.left + 1)
        c = self.config
        cfg = PreciseLossCfg(c.char_prob_pos_l2_factor, c.char_prob_neg_l2_factor, c.char_up_left_offset_l1_factor,
                             c.char_up_left_distance_regulation_l1_factor, c.char_corner_angle_cross_entropy_factor,
                             c.char_corner_distance_l1_factor, c.loss_factor, self.smooth_beta, scale)
        return ops.PreciseLoss.apply(precise_char_prob_feature, precise_char_up_left_corner_offset_feature,
                                     precise_char_corner_angle_feature, precise_char_corner_distance_feature,
                                     downsampled_char_prob_score_map, downsampled_char_mask,
                                     downsampled_label_point_y, downsampled_label_point_x, char_up_left_offsets,
                                     char_corner_angles, char_corner_distances, int(box.up), int(box.left), cfg)
