"""ORACLE — test infrastructure only.  NOT part of the product path.

CPU restatement (plain PyTorch tensor algebra, fp32 or fp64, NCHW) of the reference's
adaptive-scaling forward path and losses.  It exists to check the HIP path; only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it.  The product package (``vkit_ocr_model_adaptive_scaling_amd``) never does.

Every function works on a *state dict* (name -> tensor with the reference's schema,
SURVEY.md §8(b)) instead of on nn.Modules, so that the same seeded parameter set can be
pushed through the reference (when generating goldens), through this oracle and through
the HIP modules.  Gradients come from torch autograd on these formulas.

Pinning: ``tests/golden/make_golden.py`` runs the imported reference (``/root/reference``)
in the build container and stores its outputs / gradients; ``tests/test_oracle_golden.py``
checks this file against those fixtures.  The focal term follows torchvision's published
``sigmoid_focal_loss`` formula (torchvision is not installed anywhere in this pipeline and
the reference holds no known-answer test for it): that single term is "parity unpinned"
against third-party code, pinned only through the closed form used when generating goldens.

Reference citations are relative to /root/reference/vkit_open_model/.
"""
import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
from torch.nn import functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]


# ----------------------------------------------------------------------------------------
# storage-rounding mode (a model of the 16-bit storage FORMAT, not of the reference)
# ----------------------------------------------------------------------------------------
# The reference computes in fp32.  The HIP path keeps activations (and the matrix operands made from weights) in
# bf16 / fp16 between kernels and accumulates in fp32.  ``storage_rounding(dtype)`` makes this oracle round at the
# same places - every value an op hands to the next op, forward and backward - while all arithmetic stays in the
# tensors' own precision (fp64 in the tests).  Its distance from the plain oracle is the error the storage format
# alone causes; tests bound the kernels' error against it (tests/test_gpu_model.py::test_bf16_flat_gradient_*).
_STORAGE = [None]


class _RoundBoth(torch.autograd.Function):
    """Round an activation to the storage type; the gradient that flows back through the same edge is stored in that
    type too."""

    @staticmethod
    def forward(ctx, x, dtype):
        ctx.dtype = dtype
        return x.to(dtype).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g.to(ctx.dtype).to(g.dtype), None


class _RoundFwd(torch.autograd.Function):
    """Round a weight to the storage type of the matrix operand; its gradient is accumulated in fp32 (not rounded)."""

    @staticmethod
    def forward(ctx, w, dtype):
        return w.to(dtype).to(w.dtype)

    @staticmethod
    def backward(ctx, g):
        return g, None


class storage_rounding:
    def __init__(self, dtype):
        self.dtype = dtype

    def __enter__(self):
        self.old = _STORAGE[0]
        _STORAGE[0] = self.dtype
        return self

    def __exit__(self, *exc):
        _STORAGE[0] = self.old
        return False


def _q(x: Tensor) -> Tensor:
    """Activation storage point (no-op unless inside storage_rounding)."""
    return x if _STORAGE[0] is None else _RoundBoth.apply(x, _STORAGE[0])


def _qw(w: Tensor) -> Tensor:
    """Weight operand of a matrix product / depthwise convolution."""
    return w if _STORAGE[0] is None else _RoundFwd.apply(w, _STORAGE[0])


# ----------------------------------------------------------------------------------------
# primitive ops (model/helper.py)
# ----------------------------------------------------------------------------------------
def layer_norm_nchw(x: Tensor, g: Tensor, b: Tensor, eps: float = 1e-6) -> Tensor:
    """helper.ln (model/helper.py:96-97) applied between the two permutes the reference wraps
    around it: statistics over C for every (b, y, x), biased variance."""
    mu = x.mean(dim=1, keepdim=True)
    xc = x - mu
    var = (xc * xc).mean(dim=1, keepdim=True)
    y = xc * torch.rsqrt(var + eps)
    return y * g.view(1, -1, 1, 1) + b.view(1, -1, 1, 1)


def gelu(x: Tensor) -> Tensor:
    """helper.gelu (model/helper.py:100-101): exact erf form."""
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def softplus(x: Tensor) -> Tensor:
    """nn.Softplus() defaults (model/adaptive_scaling.py:101,140): beta 1, threshold 20."""
    return torch.where(x > 20.0, x, torch.log1p(torch.exp(torch.clamp(x, max=20.0))))


def linear_nchw(x: Tensor, w: Tensor, b: Optional[Tensor]) -> Tensor:
    """helper.conv1x1 = nn.Linear on the NHWC view (model/helper.py:18-22)."""
    y = torch.einsum('bchw,oc->bohw', x, _qw(w))
    if b is not None:
        y = y + b.view(1, -1, 1, 1)
    return y


def bilinear_axis(n_in: int, n_out: int, dtype) -> Tuple[Tensor, Tensor, Tensor]:
    """Source indices / weight for F.interpolate(mode='bilinear', align_corners=False) with an
    explicit output size (model/upernext.py:79,178-182,191-195,237-244):
    src = max((dst + 0.5) * in/out - 0.5, 0); i0 = floor(src); i1 = min(i0 + 1, in - 1)."""
    scale = n_in / n_out
    dst = torch.arange(n_out, dtype=torch.float64)
    src = torch.clamp((dst + 0.5) * scale - 0.5, min=0.0)
    i0 = torch.floor(src).to(torch.int64)
    i0 = torch.clamp(i0, max=n_in - 1)
    i1 = torch.clamp(i0 + 1, max=n_in - 1)
    w1 = (src - i0.to(torch.float64)).to(dtype)
    return i0, i1, w1


def resize_bilinear(x: Tensor, size: Tuple[int, int]) -> Tensor:
    hi, wi = x.shape[-2:]
    ho, wo = size
    y0, y1, wy = bilinear_axis(hi, ho, x.dtype)
    x0, x1, wx = bilinear_axis(wi, wo, x.dtype)
    rows = x.index_select(2, y0) * (1 - wy).view(1, 1, -1, 1) + x.index_select(2, y1) * wy.view(1, 1, -1, 1)
    return rows.index_select(3, x0) * (1 - wx).view(1, 1, 1, -1) + rows.index_select(3, x1) * wx.view(1, 1, 1, -1)


def resize_nearest(x: Tensor, size: Tuple[int, int]) -> Tensor:
    """F.interpolate(mode='nearest') (model/fpn.py:125-129,138-142,197-204): src = floor(dst * in/out)."""
    hi, wi = x.shape[-2:]
    ho, wo = size
    # integer form of floor(dst * in / out); exact for the sizes used here
    iy = torch.clamp((torch.arange(ho, dtype=torch.int64) * hi) // ho, max=hi - 1)
    ix = torch.clamp((torch.arange(wo, dtype=torch.int64) * wi) // wo, max=wi - 1)
    return x.index_select(2, iy).index_select(3, ix)


def adaptive_avg_pool(x: Tensor, s: int) -> Tensor:
    """nn.AdaptiveAvgPool2d(s) (model/upernext.py:62): bin i = [floor(i*H/s), ceil((i+1)*H/s))."""
    h, w = x.shape[-2:]
    rows = []
    for i in range(s):
        y0, y1 = (i * h) // s, -((-(i + 1) * h) // s)
        cols = []
        for j in range(s):
            x0, x1 = (j * w) // s, -((-(j + 1) * w) // s)
            cols.append(x[:, :, y0:y1, x0:x1].mean(dim=(2, 3)))
        rows.append(torch.stack(cols, dim=-1))
    return torch.stack(rows, dim=-2)


# ----------------------------------------------------------------------------------------
# ConvNeXt backbone (model/convnext.py)
# ----------------------------------------------------------------------------------------
def _count(sd: SD, prefix: str, fmt: str) -> int:
    n = 0
    while any(k.startswith(prefix + fmt.format(n)) for k in sd):
        n += 1
    return n


def convnext_layer(sd: SD, p: str, x: Tensor, drop_mask: Optional[Tensor]) -> Tensor:
    """ConvNextBlockLayer.forward (model/convnext.py:29-59): dw7x7 -> LN -> Linear(C,4C) -> GELU ->
    Linear(4C,C) -> * block_scale -> stochastic-depth mask -> + x.  ``drop_mask`` is the already
    divided (B,1,1,1) keep mask of apply_stochastic_depth (:41-53) or None."""
    c = x.shape[1]
    y = _q(F.conv2d(x, _qw(sd[p + 'block.0.weight']), sd[p + 'block.0.bias'], padding=3, groups=c))
    y = _q(layer_norm_nchw(y, sd[p + 'block.2.weight'], sd[p + 'block.2.bias']))
    y = _q(linear_nchw(y, sd[p + 'block.3.weight'], sd[p + 'block.3.bias']))
    y = _q(gelu(y))
    y = _q(linear_nchw(y, sd[p + 'block.5.weight'], sd[p + 'block.5.bias']))
    y = sd[p + 'block_scale'].view(1, -1, 1, 1) * y
    if drop_mask is not None:
        y = drop_mask * y
    return _q(y + x)


def convnext_forward(sd: SD, x: Tensor, prefix: str = '',
                     drop_masks: Optional[Sequence[Optional[Tensor]]] = None) -> List[Tensor]:
    """ConvNext.forward (model/convnext.py:227-235) incl. stem (:106-123) and ConvNextBlock (:93-101)."""
    wstem = sd[prefix + 'stem.0.weight']
    k = wstem.shape[-1]
    x = _q(F.conv2d(x, _qw(wstem), sd[prefix + 'stem.0.bias'], stride=k))
    x = _q(layer_norm_nchw(x, sd[prefix + 'stem.2.weight'], sd[prefix + 'stem.2.bias']))
    feats = []
    n_blocks = _count(sd, prefix, 'blocks.{}.')
    li = 0
    for bi in range(n_blocks):
        bp = f'{prefix}blocks.{bi}.'
        for l in range(_count(sd, bp, 'layers.{}.')):
            m = None if drop_masks is None else drop_masks[li]
            x = convnext_layer(sd, f'{bp}layers.{l}.', x, m)
            li += 1
        x = _q(layer_norm_nchw(x, sd[bp + 'ln.1.weight'], sd[bp + 'ln.1.bias']))
        feats.append(x)
        if bp + 'pconv2x2.weight' in sd:
            x = _q(F.conv2d(x, _qw(sd[bp + 'pconv2x2.weight']), sd[bp + 'pconv2x2.bias'], stride=2))
    return feats


def stochastic_depth_probs(num_layers: Sequence[int]) -> List[float]:
    """prob_bypass = 0.1 * idx / (total - 1) (model/convnext.py:76,130-132)."""
    total = sum(num_layers)
    return [0.1 * i / (total - 1) for i in range(total)]


# ----------------------------------------------------------------------------------------
# necks / heads (model/upernext.py, model/fpn.py)
# ----------------------------------------------------------------------------------------
def conv1x1_block(sd: SD, p: str, x: Tensor) -> Tensor:
    """build_conv1x1_block (model/upernext.py:21-36, model/fpn.py:21-28): Linear -> LN -> GELU."""
    y = _q(linear_nchw(x, sd[p + '1.weight'], sd[p + '1.bias']))
    y = layer_norm_nchw(y, sd[p + '2.weight'], sd[p + '2.bias'])
    return _q(gelu(y))


def convkxk_block(sd: SD, p: str, x: Tensor) -> Tensor:
    """build_conv3x3_block / build_conv5x5_block (model/upernext.py:39-45, model/fpn.py:31-48)."""
    w = sd[p + '0.weight']
    y = _q(F.conv2d(x, _qw(w), sd[p + '0.bias'], padding=w.shape[-1] // 2))
    y = layer_norm_nchw(y, sd[p + '2.weight'], sd[p + '2.bias'])
    return _q(gelu(y))


def ppm_forward(sd: SD, p: str, x: Tensor, ppm_scales: Sequence[int]) -> Tensor:
    """PpmBlock.forward (model/upernext.py:73-84)."""
    size = (x.shape[-2], x.shape[-1])
    feats = [x]
    for i, s in enumerate(ppm_scales):
        f = _q(adaptive_avg_pool(x, s))
        f = conv1x1_block(sd, f'{p}ap_conv_blocks.{i}.1.', f)
        feats.append(_q(resize_bilinear(f, size)))
    return convkxk_block(sd, p + 'final_conv_block.', torch.cat(feats, dim=1))


def upernext_neck_forward(sd: SD, feats: Sequence[Tensor], prefix: str = '',
                          ppm_scales: Sequence[int] = (1, 2, 3, 6)) -> Tensor:
    """UperNextNeck.forward (model/upernext.py:163-198)."""
    n = len(feats)
    outs = [conv1x1_block(sd, f'{prefix}step1_conv_blocks.{i}.', feats[i]) for i in range(n - 1)]
    outs.append(ppm_forward(sd, f'{prefix}step1_conv_blocks.{n - 1}.', feats[n - 1], ppm_scales))
    for i in range(n - 1, 0, -1):
        outs[i - 1] = _q(outs[i - 1] + resize_bilinear(outs[i], outs[i - 1].shape[-2:]))
    for i in range(n - 1):
        outs[i] = convkxk_block(sd, f'{prefix}step2_conv_blocks.{i}.', outs[i])
    size0 = feats[0].shape[-2:]
    for i in range(1, n):
        outs[i] = _q(resize_bilinear(outs[i], size0))
    return torch.cat(outs, dim=1)


def upernext_head_forward(sd: SD, x: Tensor, prefix: str, upsampling_factor: int) -> Tensor:
    """UperNextHead.forward (model/upernext.py:233-248)."""
    if upsampling_factor > 1:
        x = _q(resize_bilinear(x, (x.shape[-2] * upsampling_factor, x.shape[-1] * upsampling_factor)))
    x = convkxk_block(sd, prefix + 'step1_conv3x3.', x)
    return linear_nchw(x, sd[prefix + 'step2_conv1x1.1.weight'], sd[prefix + 'step2_conv1x1.1.bias'])


def fpn_neck_forward(sd: SD, feats: Sequence[Tensor], prefix: str = '') -> Tensor:
    """FpnNeck.forward (model/fpn.py:110-146)."""
    n = len(feats)
    outs = [conv1x1_block(sd, f'{prefix}step1_conv_blocks.{i}.', feats[i]) for i in range(n)]
    for i in range(n - 1, 0, -1):
        outs[i - 1] = _q(outs[i - 1] + resize_nearest(outs[i], outs[i - 1].shape[-2:]))
    for i in range(n):
        outs[i] = convkxk_block(sd, f'{prefix}step2_conv_blocks.{i}.', outs[i])
    size0 = feats[0].shape[-2:]
    for i in range(1, n):
        outs[i] = resize_nearest(outs[i], size0)
    return torch.cat(outs, dim=1)


def fpn_head_forward(sd: SD, x: Tensor, prefix: str, upsampling_factor: int) -> Tensor:
    """FpnHead.forward (model/fpn.py:193-208); step1_conv is 3x3 for factor<=2, 5x5 for <=4 (:165-176)."""
    if upsampling_factor > 1:
        x = resize_nearest(x, (x.shape[-2] * upsampling_factor, x.shape[-1] * upsampling_factor))
    x = convkxk_block(sd, prefix + 'step1_conv.', x)
    return linear_nchw(x, sd[prefix + 'step2_conv.1.weight'], sd[prefix + 'step2_conv.1.bias'])


# ----------------------------------------------------------------------------------------
# AdaptiveScaling (model/adaptive_scaling.py)
# ----------------------------------------------------------------------------------------
def _neck(sd, feats, prefix, kind):
    return upernext_neck_forward(sd, feats, prefix) if kind == 'upernext' else fpn_neck_forward(sd, feats, prefix)


def _head(sd, x, prefix, kind, factor):
    return upernext_head_forward(sd, x, prefix, factor) if kind == 'upernext' else fpn_head_forward(sd, x, prefix, factor)


def forward_rough(sd: SD, x: Tensor, kind: str = 'upernext', upsampling_factor: int = 2,
                  drop_masks=None) -> Tuple[Tensor, Tensor]:
    """AdaptiveScaling.forward_rough (model/adaptive_scaling.py:143-154)."""
    feats = convnext_forward(sd, x, 'backbone.', drop_masks)
    nf = _neck(sd, feats, 'rough_neck.', kind)
    mask = _head(sd, nf, 'rough_char_mask_head.', kind, upsampling_factor)
    height = softplus(_head(sd, nf, 'rough_char_height_head.0.', kind, upsampling_factor))
    return mask, height


def forward_precise(sd: SD, x: Tensor, kind: str = 'upernext', upsampling_factor: int = 2,
                    drop_masks=None) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """AdaptiveScaling.forward_precise (model/adaptive_scaling.py:156-177)."""
    feats = convnext_forward(sd, x, 'backbone.', drop_masks)
    nf = _neck(sd, feats, 'precise_neck.', kind)
    prob = _head(sd, nf, 'precise_char_prob_head.', kind, upsampling_factor)
    offset = _head(sd, nf, 'precise_char_up_left_corner_offset_head.', kind, upsampling_factor)
    angle = _head(sd, nf, 'precise_char_corner_angle_head.', kind, upsampling_factor)
    dist = softplus(_head(sd, nf, 'precise_char_corner_distance_head.0.', kind, upsampling_factor))
    return prob, offset, angle, dist


# ----------------------------------------------------------------------------------------
# losses (loss_function/*.py)
# ----------------------------------------------------------------------------------------
def sigmoid_focal_mean(x: Tensor, t: Tensor, alpha: float = 0.25, gamma: float = 2.0) -> Tensor:
    """FocalWithLogitsLossFunction (loss_function/focal_with_logits.py:18-47) -> torchvision
    sigmoid_focal_loss, reduction='mean' (closed form; see module docstring)."""
    p = torch.sigmoid(x)
    ce = torch.clamp(x, min=0) - x * t + torch.log1p(torch.exp(-torch.abs(x)))
    p_t = p * t + (1 - p) * (1 - t)
    loss = ce * (1 - p_t) ** gamma
    a_t = alpha * t + (1 - alpha) * (1 - t)
    return (a_t * loss).mean()


def dice(pred: Tensor, gt: Tensor, eps: float = 1e-6) -> Tensor:
    """DiceLossFunction (loss_function/dice.py:32-34)."""
    return 1 - 2.0 * (pred * gt).sum() / (pred.sum() + gt.sum() + eps)


def smooth_l1_elem(d: Tensor, beta: float) -> Tensor:
    a = d.abs()
    return torch.where(a < beta, 0.5 * a * a / beta, a - 0.5 * beta)


def smooth_l1(pred: Tensor, gt: Tensor, beta: float = 1.0, mask: Optional[Tensor] = None,
              eps: float = 1e-6) -> Tensor:
    """L1LossFunction(smooth=True) (loss_function/l1.py:30-47)."""
    e = smooth_l1_elem(pred - gt, beta)
    if mask is None:
        return e.mean()
    return (e * mask).sum() / (mask.sum() + eps)


def l2(pred: Tensor, gt: Tensor, mask: Optional[Tensor] = None, eps: float = 1e-6) -> Tensor:
    """L2LossFunction (loss_function/l2.py:23-34)."""
    e = (pred - gt) ** 2
    if mask is None:
        return e.mean()
    return (e * mask).sum() / (mask.sum() + eps)


def soft_cross_entropy(logits: Tensor, target: Tensor) -> Tensor:
    """F.cross_entropy with probability targets of shape (B, C, P)
    (loss_function/cross_entropy_with_logits.py:16-19): mean over B*P of -sum_c t*log_softmax."""
    lsm = logits - torch.logsumexp(logits, dim=1, keepdim=True)
    return -(target * lsm).sum(dim=1).mean()


def rough_loss(mask_feat: Tensor, height_feat: Tensor, gt_mask: Tensor, gt_score: Tensor,
               core_box: Tuple[int, int, int, int], focal_factor: float = 5.0, dice_factor: float = 1.0,
               l1_factor: float = 1.0, score_min: float = 1.1, height_min: float = 1.1) -> Tensor:
    """AdaptiveScalingRoughLossFunction.__call__ (loss_function/adaptive_scaling.py:53-131) with the
    default-active terms (bce_factor = 0).  core_box = (up, down, left, right), inclusive."""
    up, down, left, right = core_box
    m = mask_feat[:, 0, up:down + 1, left:right + 1]
    h = height_feat[:, 0, up:down + 1, left:right + 1]
    loss = focal_factor * sigmoid_focal_mean(m, gt_mask)
    loss = loss + dice_factor * dice(torch.sigmoid(m), gt_mask)
    l1_mask = ((h > height_min) & (gt_score > score_min) & gt_mask.bool()).to(h.dtype)  # :112-114, before clamp
    hl = torch.log(torch.clamp(h, min=height_min))
    sl = torch.log(torch.clamp(gt_score, min=score_min))
    return loss + l1_factor * smooth_l1(hl, sl, 1.0, l1_mask)


def gather_points(feat: Tensor, py: Tensor, px: Tensor) -> Tensor:
    """get_label_point_feature (loss_function/adaptive_scaling.py:167-179): (B,C,H,W) -> (B,P,C)."""
    b = feat.shape[0]
    return feat[torch.arange(b)[:, None], :, py, px]


def precise_loss(prob: Tensor, offset: Tensor, angle: Tensor, dist: Tensor, gt_score: Tensor, gt_mask: Tensor,
                 core_box: Tuple[int, int, int, int], py: Tensor, px: Tensor, gt_offsets: Tensor,
                 gt_angles: Tensor, gt_dists: Tensor, pos_l2: float = 2.0, neg_l2: float = 1.0,
                 offset_l1: float = 1.0, reg_l1: float = 1.0, angle_ce: float = 5.0, dist_l1: float = 1.0,
                 loss_factor: float = 0.15) -> Tensor:
    """AdaptiveScalingPreciseLossFunction.__call__ (loss_function/adaptive_scaling.py:181-346), default-active terms."""
    up, down, left, right = core_box
    p = torch.sigmoid(prob[:, 0, up:down + 1, left:right + 1])
    off = gather_points(offset, py, px)
    ang = gather_points(angle, py, px).transpose(1, 2)
    dst = gather_points(dist, py, px)
    loss = pos_l2 * l2(p, gt_score, gt_mask) + neg_l2 * l2(p, gt_score, 1 - gt_mask)
    loss = loss + offset_l1 * smooth_l1(off, gt_offsets, 2.5)
    loss = loss + reg_l1 * smooth_l1(torch.linalg.norm(off, dim=2), dst[:, :, 0], 2.5)
    loss = loss + angle_ce * soft_cross_entropy(ang, gt_angles.transpose(1, 2))
    loss = loss + dist_l1 * smooth_l1(dst[:, :, 1:], gt_dists, 2.5)
    return loss * loss_factor
