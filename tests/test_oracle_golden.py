"""Pins the oracle (oracle/torch_oracle.py) to the reference: every fixture in tests/golden/ was
produced by the imported reference (tests/golden/make_golden.py); the oracle must reproduce it from
the same portable seeds.  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import torch_oracle as O
from tests.golden import recipe
from tests.helpers import golden, seeded_state_dict, check_grad_summary, rel_err

F64 = torch.float64


def t64(a):
    return torch.from_numpy(np.asarray(a)).to(F64)


def backprop(outs, seed):
    loss = 0.0
    for i, o in enumerate(outs):
        loss = loss + (o * t64(recipe.cotangent(seed, i, tuple(o.shape)))).sum()
    loss.backward()


def convnext_shapes(plan, stem_k):
    shapes = {'stem.0.weight': (plan[0][0], 3, stem_k, stem_k), 'stem.0.bias': (plan[0][0],),
              'stem.2.weight': (plan[0][0],), 'stem.2.bias': (plan[0][0],)}
    for bi, (c, n) in enumerate(plan):
        for l in range(n):
            p = f'blocks.{bi}.layers.{l}.'
            shapes.update({p + 'block_scale': (c, 1, 1), p + 'block.0.weight': (c, 1, 7, 7), p + 'block.0.bias': (c,),
                           p + 'block.2.weight': (c,), p + 'block.2.bias': (c,), p + 'block.3.weight': (4 * c, c),
                           p + 'block.3.bias': (4 * c,), p + 'block.5.weight': (c, 4 * c), p + 'block.5.bias': (c,)})
        shapes.update({f'blocks.{bi}.ln.1.weight': (c,), f'blocks.{bi}.ln.1.bias': (c,)})
        if bi + 1 < len(plan):
            shapes.update({f'blocks.{bi}.pconv2x2.weight': (plan[bi + 1][0], c, 2, 2),
                           f'blocks.{bi}.pconv2x2.bias': (plan[bi + 1][0],)})
    return shapes


def upernext_neck_shapes(group, out, scales=(1, 2, 3, 6)):
    inner = out // len(group)
    s = {}
    for i, c in enumerate(group[:-1]):
        p = f'step1_conv_blocks.{i}.'
        s.update({p + '1.weight': (inner, c), p + '1.bias': (inner,), p + '2.weight': (inner,), p + '2.bias': (inner,)})
    p = f'step1_conv_blocks.{len(group) - 1}.'
    for k in range(len(scales)):
        q = f'{p}ap_conv_blocks.{k}.1.'
        s.update({q + '1.weight': (inner, group[-1]), q + '1.bias': (inner,), q + '2.weight': (inner,), q + '2.bias': (inner,)})
    cin = group[-1] + len(scales) * inner
    s.update({p + 'final_conv_block.0.weight': (inner, cin, 3, 3), p + 'final_conv_block.0.bias': (inner,),
              p + 'final_conv_block.2.weight': (inner,), p + 'final_conv_block.2.bias': (inner,)})
    for i in range(len(group) - 1):
        p = f'step2_conv_blocks.{i}.'
        s.update({p + '0.weight': (inner, inner, 3, 3), p + '0.bias': (inner,), p + '2.weight': (inner,), p + '2.bias': (inner,)})
    return s


def fpn_neck_shapes(group, out):
    inner = out // len(group)
    s = {}
    for i, c in enumerate(group):
        p = f'step1_conv_blocks.{i}.'
        s.update({p + '1.weight': (out, c), p + '1.bias': (out,), p + '2.weight': (out,), p + '2.bias': (out,)})
        p = f'step2_conv_blocks.{i}.'
        s.update({p + '0.weight': (inner, out, 3, 3), p + '0.bias': (inner,), p + '2.weight': (inner,), p + '2.bias': (inner,)})
    return s


def head_shapes(kind, cin, oc, k=3):
    inner = (cin + oc) // 2
    a, b = ('step1_conv3x3.', 'step2_conv1x1.') if kind == 'upernext' else ('step1_conv.', 'step2_conv.')
    return {a + '0.weight': (inner, cin, k, k), a + '0.bias': (inner,), a + '2.weight': (inner,), a + '2.bias': (inner,),
            b + '1.weight': (oc, inner), b + '1.bias': (oc,)}


def test_convnext_toy_eval():
    c = recipe.CONVNEXT_TOY
    g = golden('convnext_toy_eval')
    sd = seeded_state_dict(convnext_shapes(c['plan'], 4), c['seed'], c['std'], dtype=F64, requires_grad=True)
    x = t64(recipe.image(c['seed'], c['shape'])).requires_grad_(True)
    feats = O.convnext_forward(sd, x)
    for i, f in enumerate(feats):
        assert rel_err(f, g[f'out{i}']) < 1e-10
    backprop(feats, c['seed'])
    assert rel_err(x.grad, g['gx']) < 1e-9
    check_grad_summary(sd, g, tol=1e-9)


def test_convnext_toy_train_masks():
    c = recipe.CONVNEXT_TOY
    g = golden('convnext_toy_train')
    sd = seeded_state_dict(convnext_shapes(c['plan'], 4), c['seed'], c['std'], dtype=F64)
    x = t64(recipe.image(c['seed'], c['shape']))
    probs = O.stochastic_depth_probs([n for _, n in c['plan']])
    assert np.allclose(probs, g['prob_bypass'], atol=1e-15)
    masks = [t64(m).view(-1, 1, 1, 1) for m in g['masks']]
    assert any(float(m.min()) == 0.0 for m in masks), 'fixture should contain a dropped sample'
    feats = O.convnext_forward(sd, x, drop_masks=masks)
    for i, f in enumerate(feats):
        assert rel_err(f, g[f'out{i}']) < 1e-10


def test_convnext_toy_pconv2x2():
    c = recipe.CONVNEXT_TOY_P2
    g = golden('convnext_toy_pconv2x2')
    sd = seeded_state_dict(convnext_shapes(c['plan'], 2), c['seed'], c['std'], dtype=F64)
    feats = O.convnext_forward(sd, t64(recipe.image(c['seed'], c['shape'])))
    assert [tuple(f.shape[2:]) for f in feats] == [(16, 32), (8, 16), (4, 8), (2, 4)]
    for i, f in enumerate(feats):
        assert rel_err(f, g[f'out{i}']) < 1e-10


@pytest.mark.parametrize('kind', ['upernext', 'fpn'])
def test_neck_toy(kind):
    n = recipe.NECK_TOY
    g = golden(f'neck_{kind}_toy')
    shapes = (upernext_neck_shapes if kind == 'upernext' else fpn_neck_shapes)(n['in_channels_group'], n['out_channels'])
    sd = seeded_state_dict(shapes, n['seed'], n['std'], dtype=F64, requires_grad=True)
    feats = [t64(a).requires_grad_(True) for a in recipe.neck_features(n)]
    out = (O.upernext_neck_forward if kind == 'upernext' else O.fpn_neck_forward)(sd, feats)
    assert rel_err(out, g['out']) < 1e-10
    backprop([out], n['seed'])
    for i, f in enumerate(feats):
        assert rel_err(f.grad, g[f'gfeat{i}']) < 1e-9
    check_grad_summary(sd, g, tol=1e-9)


@pytest.mark.parametrize('kind', ['upernext', 'fpn'])
@pytest.mark.parametrize('case', recipe.HEAD_CASES)
def test_head_toy(kind, case):
    oc, factor, bias = case
    h = recipe.HEAD_TOY
    g = golden(f'head_{kind}_oc{oc}_f{factor}')
    sd = seeded_state_dict(head_shapes(kind, h['in_channels'], oc), h['seed'] + oc, h['std'], dtype=F64, requires_grad=True)
    x = t64(recipe.head_input(h)).requires_grad_(True)
    out = (O.upernext_head_forward if kind == 'upernext' else O.fpn_head_forward)(sd, x, '', factor)
    assert rel_err(out, g['out']) < 1e-10
    backprop([out], h['seed'])
    assert rel_err(x.grad, g['gx']) < 1e-9
    check_grad_summary(sd, g, tol=1e-9)


def test_op_semantics():
    g = golden('ops')
    for (hi, wi, ho, wo) in recipe.RESIZE_CASES:
        a = t64(recipe.plain_tensor(7, (2, 3, hi, wi)))
        assert rel_err(O.resize_bilinear(a, (ho, wo)), g[f'bilinear_{hi}x{wi}_{ho}x{wo}']) < 1e-12
        assert rel_err(O.resize_nearest(a, (ho, wo)), g[f'nearest_{hi}x{wi}_{ho}x{wo}']) == 0.0
    for (hi, wi, s) in recipe.POOL_CASES:
        a = t64(recipe.plain_tensor(9, (2, 3, hi, wi)))
        assert rel_err(O.adaptive_avg_pool(a, s), g[f'avgpool_{hi}x{wi}_{s}']) < 1e-12
    t = t64(recipe.TAIL_POINTS)
    assert np.allclose(O.gelu(t).numpy(), g['gelu_tail'], rtol=1e-12, atol=1e-300)
    assert np.allclose(O.softplus(t * 6).numpy(), g['softplus_tail'], rtol=1e-12, atol=1e-300)


@pytest.mark.parametrize('variant', ['plain', 'edge'])
def test_losses(variant):
    L = recipe.LOSS_TOY
    g = golden('losses')
    t = {k: torch.from_numpy(v) for k, v in recipe.loss_inputs(L, variant).items()}
    mf = t['mask_feat'].clone().requires_grad_(True)
    hf = t['height_feat'].clone().requires_grad_(True)
    rl = O.rough_loss(mf, hf, t['gt_mask'], t['gt_score_rough'], L['core_box'])
    rl.backward()
    # the reference builds l1_mask with .float() (fp32) even for fp64 inputs (loss_function/adaptive_scaling.py:114),
    # so its (mask.sum() + 1e-6) loses the eps in this fp64 fixture: tolerate 1e-8 here, fp32 runs are unaffected
    assert abs(float(rl.detach()) - float(g[f'{variant}/rough_loss'])) < 1e-8 * max(1.0, abs(float(rl.detach())))
    assert rel_err(mf.grad, g[f'{variant}/g_mask_feat']) < 1e-9
    assert rel_err(hf.grad, g[f'{variant}/g_height_feat']) < 1e-7
    if variant == 'edge':
        assert float((hf.grad == 0).double().mean()) > 0.3, 'edge case must mask out many height gradients'
    p = {k: t[k].clone().requires_grad_(True) for k in ('prob', 'offset', 'angle', 'dist')}
    pl = O.precise_loss(p['prob'], p['offset'], p['angle'], p['dist'], t['gt_score_precise'], t['gt_mask'], L['core_box'],
                        t['py'], t['px'], t['gt_offsets'], t['gt_angles'], t['gt_dists'])
    pl.backward()
    assert abs(float(pl.detach()) - float(g[f'{variant}/precise_loss'])) < 1e-11 * max(1.0, abs(float(pl.detach())))
    for k, v in p.items():
        assert rel_err(v.grad, g[f'{variant}/g_{k}']) < 1e-9


@pytest.mark.parametrize('kind', ['upernext', 'fpn'])
def test_full_model_tiny_256(kind):
    """Config #1 shape (1x3x256x256) through the whole path incl. both losses, fp32, vs the reference's outputs."""
    Fm = recipe.FULL_MODEL
    g = golden(f'full_tiny_{kind}_256')
    shapes = {k: eval(s) for k, s in zip(g['state_dict_keys'], g['state_dict_shapes'])}
    assert len(shapes) == (304 if kind == 'upernext' else 268) or len(shapes) > 250
    sd = seeded_state_dict(shapes, Fm['seed'], Fm['std'], dtype=torch.float32, requires_grad=True)
    t = {k: torch.from_numpy(v) for k, v in recipe.full_model_inputs(Fm).items()}
    m, h = O.forward_rough(sd, t['image_rough'], kind)
    assert rel_err(m, g['rough_mask']) < 2e-5 and rel_err(h, g['rough_height']) < 2e-5
    rl = O.rough_loss(m, h, t['gt_mask'], t['gt_score_rough'], Fm['core_box'])
    assert abs(float(rl.detach()) - float(g['rough_loss'])) < 2e-5 * abs(float(g['rough_loss']))
    (rl / 2).backward()
    check_grad_summary(sd, g, tol=2e-3, prefix='rough/')
    outs = O.forward_precise(sd, t['image_precise'], kind)
    for o, name in zip(outs, ('precise_prob', 'precise_offset', 'precise_angle', 'precise_dist')):
        assert rel_err(o, g[name]) < 2e-5, name
    pl = O.precise_loss(*outs, t['gt_score_precise'], t['gt_mask'], Fm['core_box'], t['py'], t['px'], t['gt_offsets'],
                        t['gt_angles'], t['gt_dists'])
    assert abs(float(pl.detach()) - float(g['precise_loss'])) < 2e-5 * abs(float(g['precise_loss']))
    (pl / 2).backward()
    check_grad_summary(sd, g, tol=2e-3, prefix='both/')
