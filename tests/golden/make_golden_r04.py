#!/usr/bin/env python3
"""Round-4 additions to the golden fixtures, generated from the *imported reference* like make_golden.py (same rules: run in
the build container only, inputs regenerated from portable seeds, no reference source stored anywhere):

    python tests/golden/make_golden_r04.py

``ref_width400.npz`` - the channel widths of the reference's own tests that are not multiples of 8:
    ``FpnNeck((96, 192, 384, 768), out_channels=400)`` + ``FpnHead(400, 1, 1 | 2)`` (tests/test_fpn.py:16-50) and a
    ``UperNextNeck`` with inner width 100 + ``UperNextHead(400, 2, 2)``; seeded parameters, features at 16 x 24 / 8 x 12 /
    4 x 6 / 2 x 3 pixels (the fixture pins the oracle on these widths; the GPU test also runs the reference test's own
    80 x 80 ... 10 x 10 sizes against the oracle); outputs in full, gradients as summaries.

``ref_autocast_bf16.npz`` - what the REFERENCE ITSELF loses in bf16: the reference model under
    ``torch.autocast('cpu', dtype=torch.bfloat16)`` on the two ``full_tiny_*_256`` recipes (the fixture's parameter set and the
    reference's initialisation scale), losses + the flat gradient's relative error against its own fp32 run, whole and per
    reduction bucket.  tests/test_gpu_model.py::test_flat_gradient_north_star bounds the HIP bf16 path by
    max(1e-2, these reference-held numbers) - the 16-bit bounds no longer rest on the builder's own model of the format
    alone.  The losses are evaluated in fp32 on the (bf16) head outputs cast up, as the HIP path does (its projections and
    losses are fp32).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, '/root/reference')
sys.path.insert(0, ROOT)

from tests.golden import recipe  # noqa: E402
from tests.golden.make_golden import install_stand_ins, load_params, grad_summary, backprop, save  # noqa: E402
from vkit_ocr_model_adaptive_scaling_amd.utils import portable_rng as prng  # noqa: E402


def bucket_of(name):
    """training/ddp.py::adaptive_scaling_buckets (restated in tests/test_gpu_model.py::_bucket_of)."""
    if name.startswith('rough_'):
        return 'rough'
    if name.startswith('precise_'):
        return 'precise'
    if name.startswith('backbone.blocks.'):
        return 'backbone%d' % int(name.split('.')[2])
    return 'backbone0'


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def main():
    Box = install_stand_ins()
    torch.set_num_threads(8)
    from vkit_open_model.model import (UperNextNeck, UperNextHead, FpnNeck, FpnHead, AdaptiveScaling,
                                       AdaptiveScalingConfig, AdaptiveScalingSize, AdaptiveScalingNeckHeadType)
    from vkit_open_model.loss_function import (AdaptiveScalingRoughLossFunction, AdaptiveScalingRoughLossFunctionConifg,
                                               AdaptiveScalingPreciseLossFunction,
                                               AdaptiveScalingPreciseLossFunctionConifg)

    # ---- widths that are not multiples of 8 ------------------------------------------------------------------
    torch.set_default_dtype(torch.float64)
    w = recipe.WIDTH400
    arrs = {}
    for kind, neck_cls, head_cls, head_cases in (('fpn', FpnNeck, FpnHead, ((1, 1), (1, 2))),
                                                 ('upernext', UperNextNeck, UperNextHead, ((2, 2),))):
        neck = load_params(neck_cls(w['in_channels_group'], w['out_channels']), w['seed'], w['std']).eval()
        fs = [torch.from_numpy(a).requires_grad_(True) for a in recipe.neck_features(w)]
        out = neck(fs)
        assert out.shape[1] == w['out_channels']
        backprop([out], w['seed'])
        arrs[f'{kind}/neck_out'] = out.detach().numpy().astype(np.float32)
        for i, f in enumerate(fs):
            arrs[f'{kind}/gfeat{i}'] = f.grad.numpy().astype(np.float32)
        arrs.update({f'{kind}/neck/' + k: v for k, v in grad_summary(neck).items()})
        for oc, factor in head_cases:
            head = load_params(head_cls(w['out_channels'], oc, factor), w['seed'] + 10 * oc + factor, w['head_std']).eval()
            xin = out.detach().clone().requires_grad_(True)
            y = head(xin)
            backprop([y], w['seed'] + 1)
            tag = f'{kind}/head_oc{oc}_f{factor}'
            arrs[tag + '/out'] = y.detach().numpy().astype(np.float32)
            arrs[tag + '/gx'] = xin.grad.numpy().astype(np.float32)
            arrs.update({tag + '/' + k: v for k, v in grad_summary(head).items()})
    save('ref_width400', **arrs)

    # ---- the reference's own bf16 (autocast) against its own fp32 ------------------------------------------------
    torch.set_default_dtype(torch.float32)
    Fm = recipe.FULL_MODEL
    arrs = {}
    for kind, enum in (('upernext', AdaptiveScalingNeckHeadType.UPERNEXT), ('fpn', AdaptiveScalingNeckHeadType.FPN)):
        for init, (std, bs) in (('golden', (Fm['std'], 1.0)), ('reference_init', (0.02, 1e-6))):
            model = AdaptiveScaling(AdaptiveScalingConfig(AdaptiveScalingSize.TINY, enum))
            load_params(model, Fm['seed'], std, block_scale=bs, dtype=torch.float32)
            model.eval()
            t = {k: torch.from_numpy(v) for k, v in recipe.full_model_inputs(Fm).items()}
            box = Box(*Fm['core_box'])

            def step(autocast):
                model.zero_grad(set_to_none=True)
                import contextlib
                ctx = (lambda: torch.autocast('cpu', dtype=torch.bfloat16)) if autocast else contextlib.nullcontext
                with ctx():
                    mask_feat, height_feat = model.forward_rough(t['image_rough'])
                rl = AdaptiveScalingRoughLossFunction(AdaptiveScalingRoughLossFunctionConifg())(
                    rough_char_mask_feature=mask_feat.float(), rough_char_height_feature=height_feat.float(),
                    downsampled_mask=t['gt_mask'].clone(), downsampled_score_map=t['gt_score_rough'].clone(),
                    downsampled_shape=Fm['down_shape'], downsampled_core_box=box)
                (rl / 2).backward()
                with ctx():
                    prob, offset, angle, dist = model.forward_precise(t['image_precise'])
                pl = AdaptiveScalingPreciseLossFunction(AdaptiveScalingPreciseLossFunctionConifg())(
                    precise_char_mask_feature=None, precise_char_prob_feature=prob.float(),
                    precise_char_up_left_corner_offset_feature=offset.float(), precise_char_corner_angle_feature=angle.float(),
                    precise_char_corner_distance_feature=dist.float(),
                    downsampled_char_prob_score_map=t['gt_score_precise'].clone(), downsampled_char_mask=t['gt_mask'].clone(),
                    downsampled_shape=Fm['down_shape'], downsampled_core_box=box, downsampled_label_point_y=t['py'],
                    downsampled_label_point_x=t['px'], char_up_left_offsets=t['gt_offsets'],
                    char_corner_angles=t['gt_angles'], char_corner_distances=t['gt_dists'])
                (pl / 2).backward()
                grads = {n: p.grad.detach().double().clone() for n, p in model.named_parameters() if p.grad is not None}
                return float(rl.detach()), float(pl.detach()), grads

            rl32, pl32, g32 = step(False)
            rl16, pl16, g16 = step(True)
            names = sorted(g32)
            assert sorted(g16) == names
            cat = lambda g, sel: torch.cat([g[n].reshape(-1) for n in sel])
            key = f'{kind}/{init}/'
            arrs[key + 'rough_loss_fp32'], arrs[key + 'precise_loss_fp32'] = rl32, pl32
            arrs[key + 'rough_loss_bf16'], arrs[key + 'precise_loss_bf16'] = rl16, pl16
            arrs[key + 'flat_grad_rel_err'] = rel(cat(g16, names), cat(g32, names))
            buckets = {}
            for n in names:
                buckets.setdefault(bucket_of(n), []).append(n)
            for b, sel in sorted(buckets.items()):
                arrs[key + 'bucket/' + b] = rel(cat(g16, sel), cat(g32, sel))
            # per parameter: only where the fp32 gradient is not (numerically) zero - a conv bias in front of a LayerNorm has
            # an analytically zero gradient
            gmax = max(float(g32[n].norm()) for n in names)
            live = [n for n in names if float(g32[n].norm()) > 1e-6 * gmax]
            perr = np.array([rel(g16[n], g32[n]) for n in live])
            names = live
            arrs[key + 'worst_param_rel_err'] = float(perr.max())
            arrs[key + 'worst_param'] = np.array(names[int(perr.argmax())])
            arrs[key + 'params_over_1e-2'] = int((perr > 1e-2).sum())
            arrs[key + 'num_params'] = len(names)
            print(key, 'flat', arrs[key + 'flat_grad_rel_err'], 'losses', (rl32, rl16), (pl32, pl16), 'worst',
                  arrs[key + 'worst_param'], arrs[key + 'worst_param_rel_err'], 'over 1e-2:', arrs[key + 'params_over_1e-2'])
    save('ref_autocast_bf16', **arrs)


if __name__ == '__main__':
    main()
