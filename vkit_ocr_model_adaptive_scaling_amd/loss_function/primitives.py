"""Primitive loss callables (mirror of the classes vkit_open_model/loss_function/__init__.py:12-18 exports), each a
small HIP reduction kernel behind the reference's constructor / call signature.  Inputs live on the GPU (there is no
CPU fallback); the result is a 0-d fp32 tensor that back-propagates into ``pred``.

Differences from the reference, all deliberate: ``gt`` is never modified (dice.py:28-30 multiplies the caller's
tensor by the mask in place, SURVEY.md App. A "aliasing bugs"), and gradients are produced for ``pred`` only.
"""
from typing import Optional

import torch

from .. import ops
from .._lib import LOSS_FOCAL, LOSS_DICE, LOSS_L1, LOSS_SMOOTH_L1, LOSS_L2


class FocalWithLogitsLossFunction:
    """focal_with_logits.py:18-47 (torchvision.ops.sigmoid_focal_loss closed form: alpha_t * (1 - p_t)^gamma * BCE)."""

    def __init__(self, alpha: float = 0.25, gamma: float = 2, eps: float = 1E-6):
        self.alpha = alpha
        self.gamma = gamma
        self.eps = eps

    def __call__(self, pred: torch.Tensor, gt: torch.Tensor, mask: Optional[torch.Tensor] = None):
        return ops.ElementwiseLoss.apply(pred, gt, mask, LOSS_FOCAL, float(self.alpha), float(self.gamma), float(self.eps))


class DiceLossFunction:
    """dice.py:17-35: ``pred`` are probabilities."""

    def __init__(self, eps: float = 1E-6):
        self.eps = eps

    def __call__(self, pred: torch.Tensor, gt: torch.Tensor, mask: Optional[torch.Tensor] = None):
        return ops.ElementwiseLoss.apply(pred, gt, mask, LOSS_DICE, 0.0, 0.0, float(self.eps))


class L1LossFunction:
    """l1.py:19-47"""

    def __init__(self, eps: float = 1E-6, smooth: bool = False, smooth_beta: float = 1.0):
        self.smooth = smooth
        self.smooth_beta = smooth_beta
        self.eps = eps

    def __call__(self, pred: torch.Tensor, gt: torch.Tensor, mask: Optional[torch.Tensor] = None):
        if self.smooth and not self.smooth_beta > 0:
            raise ValueError('smooth_beta must be positive')  # F.smooth_l1_loss(beta=0) is plain L1: use smooth=False
        kind = LOSS_SMOOTH_L1 if self.smooth else LOSS_L1
        return ops.ElementwiseLoss.apply(pred, gt, mask, kind, float(self.smooth_beta), 0.0, float(self.eps))


class L2LossFunction:
    """l2.py:18-34"""

    def __init__(self, eps: float = 1E-6):
        self.eps = eps

    def __call__(self, pred: torch.Tensor, gt: torch.Tensor, mask: Optional[torch.Tensor] = None):
        return ops.ElementwiseLoss.apply(pred, gt, mask, LOSS_L2, 0.0, 0.0, float(self.eps))


class CrossEntropyWithLogitsLossFunction:
    """cross_entropy_with_logits.py:16-19 (F.cross_entropy, mean reduction): ``pred`` (N, C) or (N, C, d1, ...) logits
    with the class axis at dim 1 - the adaptive-scaling corner angles arrive as (B, 4, P),
    loss_function/adaptive_scaling.py:248-251; ``gt`` class probabilities of the same shape or int64 class indices
    (N, d1, ...).  The class axis is moved last (a view + one copy) and the rows go through one HIP kernel."""

    def __call__(self, pred: torch.Tensor, gt: torch.Tensor):
        if pred.dim() < 2:
            raise ValueError(f'cross entropy: logits must be (N, C, ...), got {tuple(pred.shape)}')
        classes = pred.shape[1]
        soft = gt.is_floating_point()
        if soft and gt.shape != pred.shape:
            raise ValueError(f'cross entropy: probability target {tuple(gt.shape)} != logits {tuple(pred.shape)}')
        if pred.dim() > 2:
            pred = pred.movedim(1, -1).reshape(-1, classes)
            gt = gt.movedim(1, -1).reshape(-1, classes) if soft else gt.reshape(-1)
        return ops.CrossEntropy.apply(pred, gt)


class WeightedBceWithLogitsLossFunction:
    """weighted_bce_with_logits.py — inactive under the default factors (bce_factor = 0); no HIP kernel."""

    def __init__(self, *args, **kwargs):
        raise NotImplementedError('weighted BCE is off by default on the adaptive-scaling path and has no HIP kernel')


class WeightAdaptiveHeatmapRegressionLossFunction:
    """weight_adaptive_heatmap_regression.py — inactive under the default factors (wahr_factor = 0); no HIP kernel."""

    def __init__(self, *args, **kwargs):
        raise NotImplementedError('WAHR is off by default on the adaptive-scaling path and has no HIP kernel')
