"""Properties at BASELINE.json's full sizes (config #3: ConvNeXt-T + UPerNext, 1024x1024, bf16) where the oracle is
too slow to run: size-independent invariants of the path (linearity of the convolutions, constants through the
resampling ops, per-pixel LayerNorm statistics, identity depthwise kernel, directional-derivative check of the whole
backward pass, run-to-run reproducibility) plus one complete train step against the oracle + torch.optim.AdamW at a
size the oracle finishes in seconds."""
import math

import numpy as np

import pytest
import torch

pytestmark = pytest.mark.gpu


def ops_mod():
    from vkit_ocr_model_adaptive_scaling_amd import ops
    return ops


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm())


def test_head_conv_linearity_fullsize():
    """3x3 implicit GEMM at the head size (M = 2*512*512 here, Cin 384, N 192): conv(a*x1 + x2) = a*conv(x1) + conv(x2)
    with zero bias, and a spatially constant input gives a constant interior (zero padding only touches the border)."""
    ops = ops_mod()
    g = torch.Generator(device='cuda').manual_seed(1)
    B, H, W, C, N = 2, 512, 512, 384, 192
    w = torch.randn((N, C, 3, 3), generator=g, device='cuda') * 0.02
    x1 = torch.randn((B, H, W, C), generator=g, device='cuda').bfloat16()
    x2 = torch.randn((B, H, W, C), generator=g, device='cuda').bfloat16()
    xs = (2.0 * x1.float() + x2.float()).bfloat16()  # exactly representable sums are not guaranteed: compare in tolerance
    y1 = ops.Conv.apply(x1, w, None, 1, 1).float()
    y2 = ops.Conv.apply(x2, w, None, 1, 1).float()
    ys = ops.Conv.apply(xs, w, None, 1, 1).float()
    lin = 2.0 * y1 + y2
    assert rel(ys, lin) < 1.5e-2
    const = torch.ones((1, H, W, C), device='cuda', dtype=torch.bfloat16)
    yc = ops.Conv.apply(const, w, None, 1, 1).float()
    interior = yc[0, 1:-1, 1:-1]
    ref = w.bfloat16().float().sum(dim=(1, 2, 3))
    assert float((interior - ref).abs().max()) <= 2e-2 * float(ref.abs().max()) + 1e-2
    # border pixels see fewer taps: top-left corner = sum over the 2x2 lower-right taps
    corner = w.bfloat16().float()[:, :, 1:, 1:].sum(dim=(1, 2, 3))
    assert float((yc[0, 0, 0] - corner).abs().max()) <= 2e-2 * float(corner.abs().max()) + 1e-2


def test_resize_pool_constants_and_adjoint_fullsize():
    """Bilinear / nearest resampling and adaptive pooling reproduce constants; backward is the exact adjoint:
    <R x, y> == <x, R^T y> (checked in fp32 at the head upsample size 256^2 -> 512^2, 384 channels)."""
    ops = ops_mod()
    g = torch.Generator(device='cuda').manual_seed(2)
    x = torch.full((2, 256, 256, 384), 3.0, device='cuda', dtype=torch.bfloat16)
    for mode in (0, 1):
        y = ops.Resize.apply(x, (512, 512), mode)
        assert float((y.float() - 3.0).abs().max()) == 0.0
    p = ops.AdaptiveAvgPool.apply(torch.full((2, 32, 32, 768), -1.5, device='cuda', dtype=torch.bfloat16), 6)
    assert float((p.float() + 1.5).abs().max()) == 0.0
    xf = torch.randn((1, 256, 256, 384), generator=g, device='cuda').requires_grad_(True)
    yv = torch.randn((1, 512, 512, 384), generator=g, device='cuda')
    for mode in (0, 1):
        xf.grad = None
        out = ops.Resize.apply(xf, (512, 512), mode)
        lhs = float((out.double() * yv.double()).sum())
        out.backward(yv)
        rhs = float((xf.detach().double() * xf.grad.double()).sum())
        assert abs(lhs - rhs) <= 1e-5 * max(abs(lhs), 1.0), (mode, lhs, rhs)


def test_layernorm_statistics_fullsize():
    """Per-pixel statistics of LN outputs at the stage-0 size (8 x 256 x 256 x 96): mean 0, variance 1 (gamma 1, beta 0)."""
    ops = ops_mod()
    g = torch.Generator(device='cuda').manual_seed(3)
    x = (torch.randn((8, 256, 256, 96), generator=g, device='cuda') * 7 + 3).bfloat16()
    y = ops.LayerNorm.apply(x, torch.ones(96, device='cuda'), torch.zeros(96, device='cuda'), False).float()
    assert float(y.mean(dim=-1).abs().max()) < 2e-2
    assert float((y.var(dim=-1, unbiased=False) - 1).abs().max()) < 3e-2


def test_convnext_layer_identity_kernel_fullsize():
    """A delta depthwise kernel, zero MLP output weights: the layer must return its input exactly (residual path),
    whatever the layer scale and the stochastic-depth mask are; with the mask at zero the branch is dropped too."""
    ops = ops_mod()
    g = torch.Generator(device='cuda').manual_seed(4)
    B, H, W, C = 8, 256, 256, 96
    x = torch.randn((B, H, W, C), generator=g, device='cuda').bfloat16()
    dw = torch.zeros((C, 1, 7, 7), device='cuda')
    dw[:, 0, 3, 3] = 1.0
    z = lambda *s: torch.zeros(*s, device='cuda')
    w1 = torch.randn((4 * C, C), generator=g, device='cuda') * 0.1
    mask = torch.tensor([1.0, 0.0, 1.25, 1.25, 0.0, 1.0, 1.0, 1.25], device='cuda')
    y = ops.ConvNextLayer.apply(x, dw, z(C), torch.ones(C, device='cuda'), z(C), w1, z(4 * C), z(C, 4 * C), z(C),
                                torch.ones((C, 1, 1), device='cuda'), mask)
    assert torch.equal(y, x)
    w2 = torch.randn((C, 4 * C), generator=g, device='cuda') * 0.1
    y2 = ops.ConvNextLayer.apply(x, dw, z(C), torch.ones(C, device='cuda'), z(C), w1, z(4 * C), w2, z(C),
                                 torch.ones((C, 1, 1), device='cuda'), mask)
    dropped = mask == 0
    assert torch.equal(y2[dropped], x[dropped]) and not torch.equal(y2[~dropped], x[~dropped])


def _tiny_upernext(dtype, size, batch, seed=5):
    from vkit_ocr_model_adaptive_scaling_amd.model import (AdaptiveScaling, AdaptiveScalingConfig, AdaptiveScalingSize,
                                                           AdaptiveScalingNeckHeadType)
    from tests.test_gpu_model import seed_module
    model = AdaptiveScaling(AdaptiveScalingConfig(AdaptiveScalingSize.TINY, AdaptiveScalingNeckHeadType.UPERNEXT),
                            compute_dtype=dtype)
    seed_module(model, seed, 0.05)
    return model.cuda().eval()


def _batches(batch, size, seed):
    import bench
    return bench.synthetic_batches(batch, (size, size), torch.device('cuda'), seed)


def test_directional_derivative_fullsize_fp32():
    """Whole backward pass at 1024x1024 (fp32 mode, B=1): <grad, v> matches the central finite difference of the rough
    loss along a random parameter direction v."""
    from vkit_ocr_model_adaptive_scaling_amd.loss_function import (AdaptiveScalingRoughLossFunction,
                                                                   AdaptiveScalingRoughLossFunctionConifg)
    model = _tiny_upernext(torch.float32, 1024, 1)
    rough, _ = _batches(1, 1024, 11)
    loss_fn = AdaptiveScalingRoughLossFunction(AdaptiveScalingRoughLossFunctionConifg())

    def loss():
        m, h = model.forward_rough(rough['image'])
        return loss_fn(m, h, rough['downsampled_mask'], rough['downsampled_score_map'], rough['downsampled_shape'],
                       rough['downsampled_core_box'])

    names = ['rough_char_mask_head.step1_conv3x3.0.weight', 'rough_neck.step2_conv_blocks.0.0.weight',
             'backbone.blocks.1.layers.0.block.3.weight', 'backbone.blocks.0.layers.1.block.0.weight',
             'backbone.blocks.2.ln.1.weight']
    params = dict(model.named_parameters())
    model.zero_grad()
    loss().backward()
    g = torch.Generator(device='cuda').manual_seed(6)
    dirs = {n: torch.randn(params[n].shape, generator=g, device='cuda') for n in names}
    analytic = sum(float((params[n].grad.double() * dirs[n].double()).sum()) for n in names)
    scale = sum(float(dirs[n].double().pow(2).sum()) for n in names) ** 0.5
    eps = 2e-2 / scale * sum(float(params[n].double().pow(2).sum()) for n in names) ** 0.5
    with torch.no_grad():
        for n in names:
            params[n].add_(eps * dirs[n])
        lp = float(loss())
        for n in names:
            params[n].add_(-2 * eps * dirs[n])
        lm = float(loss())
        for n in names:
            params[n].add_(eps * dirs[n])
    fd = (lp - lm) / (2 * eps)
    assert abs(fd - analytic) <= 3e-2 * max(abs(analytic), abs(fd)) + 1e-6, (fd, analytic)


def test_fullsize_bf16_step_is_finite_and_reproducible_forward():
    """Config #3 shapes (B=2 to bound the test time): both passes run, losses are finite, the forward outputs are
    bit-identical across two runs (no float atomics on the forward path)."""
    model = _tiny_upernext(torch.bfloat16, 1024, 2)
    rough, precise = _batches(2, 1024, 12)
    with torch.no_grad():
        a = model.forward_rough(rough['image'])
        b = model.forward_rough(rough['image'])
        p = model.forward_precise(precise['image'])
    assert all(torch.equal(x, y) for x, y in zip(a, b))
    assert [tuple(t.shape) for t in p] == [(2, 1, 512, 512), (2, 2, 512, 512), (2, 4, 512, 512), (2, 4, 512, 512)]
    assert all(torch.isfinite(t).all() for t in a + p)
    assert float(a[1].min()) >= 0.0 and float(p[3].min()) >= 0.0  # Softplus heads


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16], ids=['f32', 'bf16'])
def test_train_step_matches_oracle_and_torch_adamw(dtype):
    """One complete reference-semantics step (train.py:397-478): rough fwd/loss/bwd, precise fwd/loss/bwd with
    accumulation, clip_grad_norm_(2.5), AdamW — TwoPassStep + FlatAdamW on the GPU vs the oracle + torch.optim on the host."""
    from oracle import torch_oracle as O
    from vkit_ocr_model_adaptive_scaling_amd.loss_function import (
        AdaptiveScalingRoughLossFunction, AdaptiveScalingRoughLossFunctionConifg, AdaptiveScalingPreciseLossFunction,
        AdaptiveScalingPreciseLossFunctionConifg)
    from vkit_ocr_model_adaptive_scaling_amd.training import FlatBuffers, FlatAdamW, TwoPassStep
    size, batch = 128, 2
    model = _tiny_upernext(dtype, size, batch, seed=7)
    rough, precise = _batches(batch, size, 13)
    sd0 = {k: v.detach().cpu().double().clone() for k, v in model.state_dict().items()}
    flat = FlatBuffers(model.named_parameters())
    opt = FlatAdamW(None, lr=8e-4, betas=(0.9, 0.999), weight_decay=0.01, max_grad_norm=2.5, flat=flat)
    step = TwoPassStep(model, AdaptiveScalingRoughLossFunction(AdaptiveScalingRoughLossFunctionConifg()),
                       AdaptiveScalingPreciseLossFunction(AdaptiveScalingPreciseLossFunctionConifg()), opt)
    seen = {}
    opt_step = opt.step

    def capturing_step(lr=None):
        seen['grad'] = flat.flat_grad.detach().clone()   # accumulated, un-clipped gradient of both passes
        opt_step(lr=lr)
    opt.step = capturing_step
    rl, pl = step(rough, precise, lr=8e-4)
    # host reference
    ref = {k: torch.nn.Parameter(v.clone()) for k, v in sd0.items()}
    cpu = lambda d: {k: (v.cpu().double() if isinstance(v, torch.Tensor) and v.is_floating_point() else
                         (v.cpu() if isinstance(v, torch.Tensor) else v)) for k, v in d.items()}
    r, p = cpu(rough), cpu(precise)
    box = r['downsampled_core_box']
    cb = (box.up, box.down, box.left, box.right)
    m, h = O.forward_rough(ref, r['image'], 'upernext')
    lr_ = O.rough_loss(m, h, r['downsampled_mask'], r['downsampled_score_map'], cb) / 2
    lr_.backward()
    outs = O.forward_precise(ref, p['image'], 'upernext')
    lp_ = O.precise_loss(*outs, p['downsampled_score_map'], p['downsampled_mask'], cb, p['downsampled_label_point_y'],
                         p['downsampled_label_point_x'], p['up_left_offsets'], p['corner_angles'], p['corner_distances']) / 2
    lp_.backward()
    tol_l = 1e-4 if dtype == torch.float32 else 1e-2
    assert abs(float(rl) - float(lr_)) <= tol_l * abs(float(lr_)) and abs(float(pl) - float(lp_)) <= tol_l * abs(float(lp_))
    from tests import parity_log
    tag = 'train_step_128[%s]' % ('f32' if dtype == torch.float32 else 'bf16')
    g_dev = seen['grad'].double().cpu()
    g_got = torch.cat([g_dev[s0:s0 + n] for s0, n in (flat.offsets[k] for k in flat.names)])
    g_ref = torch.cat([ref[k].grad.reshape(-1) for k in flat.names])
    ge = rel(g_got, g_ref)
    gtol = 1e-3 if dtype == torch.float32 else 1e-2   # north_star: gradient within 1e-2 in bf16
    if dtype != torch.float32:
        # what bf16 storage alone does to this gradient (oracle with rounded stored activations, no kernels)
        refq = {k: v.clone().requires_grad_(True) for k, v in sd0.items()}
        with O.storage_rounding(dtype):
            mq, hq = O.forward_rough(refq, r['image'], 'upernext')
            (O.rough_loss(mq, hq, r['downsampled_mask'], r['downsampled_score_map'], cb) / 2).backward()
            oq = O.forward_precise(refq, p['image'], 'upernext')
            (O.precise_loss(*oq, p['downsampled_score_map'], p['downsampled_mask'], cb, p['downsampled_label_point_y'],
                            p['downsampled_label_point_x'], p['up_left_offsets'], p['corner_angles'],
                            p['corner_distances']) / 2).backward()
        qe = rel(torch.cat([refq[k].grad.reshape(-1) for k in flat.names]), g_ref)
        gtol = max(gtol, 1.1 * qe)
        parity_log.record(tag, 'storage-rounded oracle vs oracle: flat gradient', qe, None, 'format error, no kernels')
    parity_log.record(tag, 'flat gradient before the clip', ge, gtol)
    print('train-step gradient rel err', dtype, ge)
    assert ge < gtol, ge
    params = list(ref.values())
    torch.nn.utils.clip_grad_norm_(params, 2.5)
    torch.optim.AdamW(params, lr=8e-4, betas=(0.9, 0.999), weight_decay=0.01).step()
    upd_ref = torch.cat([(ref[k].detach() - sd0[k]).reshape(-1) for k in sd0])
    upd = torch.cat([(v.detach().cpu().double() - sd0[k]).reshape(-1) for k, v in model.state_dict().items()])
    e = rel(upd, upd_ref)
    print('train-step update rel err', dtype, e)
    # The gradient is what is held to the north-star bound (above).  Adam's FIRST update is lr * sign(g) wherever |g| >> eps
    # (m / sqrt(v) = +-1): an entry whose gradient is near zero flips sign on a 1e-2 perturbation and then contributes
    # 2 lr, so this is a coarse check of clip + AdamW + weight decay on top of that gradient, not a numerics bound:
    # fraction of flipped entries ~ e^2 / 4
    utol = 2e-3 if dtype == torch.float32 else 1.5e-1
    parity_log.record(tag, 'first AdamW update (sign-like, see the test)', e, utol)
    assert e < utol
    assert float(flat.flat_grad.abs().max()) == 0.0  # zero_grad() ran


def test_config3_full_batch_schedules_and_label_point_paths_agree():
    """BASELINE.json configs[2] at its FULL size under assertions (B = 8 rough + B = 8 precise, 1024 x 1024, bf16 - what
    bench.py times): the merged schedule, the reference's two-pass order, the dense evaluation of the regression heads'
    backward and the opt-in label-point forward all give the same losses and the same accumulated gradient (flat buffer,
    norm-wise; the differences are bf16 roundings of dx at the head input, see test_point_sparse_head_backward_matches_dense),
    and one optimizer step leaves finite parameters."""
    from vkit_ocr_model_adaptive_scaling_amd import ops
    from vkit_ocr_model_adaptive_scaling_amd.loss_function import (
        AdaptiveScalingRoughLossFunction, AdaptiveScalingRoughLossFunctionConifg,
        AdaptiveScalingPreciseLossFunction, AdaptiveScalingPreciseLossFunctionConifg)
    from vkit_ocr_model_adaptive_scaling_amd.training import FlatBuffers, FlatAdamW, TwoPassStep
    model = _tiny_upernext(torch.bfloat16, 1024, 8)   # eval mode: no stochastic depth, the runs are comparable
    flat = FlatBuffers(model.named_parameters())
    rough, precise = _batches(8, 1024, 21)
    rl = AdaptiveScalingRoughLossFunction(AdaptiveScalingRoughLossFunctionConifg())
    pl = AdaptiveScalingPreciseLossFunction(AdaptiveScalingPreciseLossFunctionConifg())

    class NoStep:  # gradients only: TwoPassStep's optimizer hook
        def step(self, lr=None):
            pass

        def zero_grad(self):
            pass

    def run(merge, sparse=True, points=False):
        flat.flat_grad.zero_()
        old = ops._POINT_SPARSE
        ops._POINT_SPARSE = sparse
        try:
            step = TwoPassStep(model, rl, pl, NoStep(), merge_backbone=merge, label_point_forward=points)
            lr_, lp_ = step(rough, precise)
            torch.cuda.synchronize()
            return float(lr_), float(lp_), flat.flat_grad.clone()
        finally:
            ops._POINT_SPARSE = old
    base = run(True)
    assert all(map(np.isfinite, base[:2])) and bool(torch.isfinite(base[2]).all()) and float(base[2].norm()) > 0
    for name, other in (('two-pass order', run(False)), ('dense point backward', run(True, sparse=False)),
                        ('label-point forward', run(True, points=True))):
        assert abs(other[0] - base[0]) <= 1e-6 * abs(base[0]), name                # the rough pass is untouched
        assert abs(other[1] - base[1]) <= 2e-3 * abs(base[1]), (name, other[1], base[1])
        err = float(rel(other[2], base[2]))
        print(f'config #3 full batch, {name}: gradient rel err {err:.2e}')
        assert err < 2e-3, (name, err)  # measured: 2e-7 (two-pass), 5e-5 (dense backward), 7e-5 (label-point forward)
    opt = FlatAdamW(None, lr=8e-4, betas=(0.9, 0.999), weight_decay=0.01, max_grad_norm=2.5, flat=flat)
    TwoPassStep(model, rl, pl, opt, merge_backbone=True)(rough, precise)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(flat.flat_param).all())
    ops.check_deferred(wait=True)
