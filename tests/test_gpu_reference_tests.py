"""The reference's own module tests, restated against the mirror on the MI355X.

/root/reference/tests/test_fpn.py:16-50, tests/test_upernext.py:16-31 and tests/test_convnext.py:42-76 build the necks, heads
and the backbone on their own, call them on NCHW tensors, script them (``torch.jit.script``) and - the backbone - CALL the
scripted module and round-trip its state dict.  Their configurations include channel widths that are not multiples of 8
(``FpnNeck((96, 192, 384, 768), out_channels=400)`` = 4 x 100 channels, whose concatenation puts every level at a 200-byte
offset).  The tests below restate them line for line on GPU tensors and add what the reference's assertions do not check:
the numbers, against the oracle on the same inputs and parameters (fp32 <= 1e-3, 16-bit within the bounds of
tests/test_gpu_model.py) and against a fixture the imported reference generated at these widths
(tests/golden/ref_width400.npz, make_golden_r04.py); a scripted sub-module must RUN and equal the eager module bit for bit.
"""
import numpy as np
import pytest
import torch

from oracle import torch_oracle as O
from tests.golden import recipe
from tests.helpers import golden, rel_err, check_grad_summary
from tests.test_gpu_model import DTYPES, IDS, FWD_TOL, GRAD_TOL, fmt_bound, seed_module, cot, named_params, _rec, _tag

pytestmark = pytest.mark.gpu


def _oracle_sd(module, dtype=torch.float64):
    return {k: v.detach().to(dtype).cpu().requires_grad_(True) for k, v in module.state_dict().items()}


def _fmt(dtype):
    import contextlib
    return O.storage_rounding(dtype) if dtype != torch.float32 else contextlib.nullcontext()


# ------------------------------------------------------------------------------------------- tests/test_fpn.py:16-50
def test_fpn():
    from vkit_ocr_model_adaptive_scaling_amd.model.fpn import FpnNeck, FpnHead
    neck = FpnNeck(
        in_channels_group=(96, 192, 384, 768),
        out_channels=400,
    ).cuda()
    features = [
        torch.rand(1, 96, 80, 80).cuda(),
        torch.rand(1, 192, 40, 40).cuda(),
        torch.rand(1, 384, 20, 20).cuda(),
        torch.rand(1, 768, 10, 10).cuda(),
    ]
    neck_output = neck(features)
    assert neck_output.shape == (1, 400, 80, 80)

    model_jit = torch.jit.script(neck)  # type: ignore
    assert model_jit
    # beyond the reference's assertion: the scripted neck runs, and gives the eager result bit for bit
    assert torch.equal(model_jit(features), neck_output)

    head = FpnHead(
        in_channels=400,
        out_channels=1,
        upsampling_factor=1,
    ).cuda()
    head_output = head(neck_output)
    assert head_output.shape == (1, 1, 80, 80)

    head = FpnHead(
        in_channels=400,
        out_channels=1,
        upsampling_factor=2,
    ).cuda()
    head_output = head(neck_output)
    assert head_output.shape == (1, 1, 160, 160)

    model_jit = torch.jit.script(head)  # type: ignore
    assert model_jit
    assert torch.equal(model_jit(neck_output), head_output)


# ------------------------------------------------------------------------------------------- tests/test_upernext.py:16-31
def test_upernext():
    from vkit_ocr_model_adaptive_scaling_amd.model.upernext import UperNextNeck
    model = UperNextNeck(
        in_channels_group=(96, 192, 384, 768),
        out_channels=384,
    ).cuda()
    features = [
        torch.rand(1, 96, 80, 80).cuda(),
        torch.rand(1, 192, 40, 40).cuda(),
        torch.rand(1, 384, 20, 20).cuda(),
        torch.rand(1, 768, 10, 10).cuda(),
    ]
    output = model(features)
    assert output.shape == (1, 384, 80, 80)

    model_jit = torch.jit.script(model)  # type: ignore
    assert model_jit
    assert torch.equal(model_jit(features), output)


# ------------------------------------------------------------------------------------------- tests/test_convnext.py:42-76
def test_convnext():
    from vkit_ocr_model_adaptive_scaling_amd.model.convnext import ConvNext
    model = ConvNext.create_tiny().cuda()
    x = torch.rand((1, 3, 320, 320)).cuda()
    features = model(x)  # type: ignore
    assert len(features) == 4
    assert features[0].shape == (1, 96, 80, 80)
    assert features[1].shape == (1, 192, 40, 40)
    assert features[2].shape == (1, 384, 20, 20)
    assert features[3].shape == (1, 768, 10, 10)


def test_convnext_jit(tmp_path):
    from vkit_ocr_model_adaptive_scaling_amd.model.convnext import ConvNext
    model = ConvNext.create_tiny(stem_use_pconv2x2=True)
    model_jit = torch.jit.script(model)  # type: ignore  (on the host, as the reference does)
    model_jit = model_jit.cuda()

    x = torch.rand((1, 3, 320, 320)).cuda()
    features = model_jit(x)  # type: ignore
    assert len(features) == 4
    assert features[0].shape == (1, 96, 160, 160)
    assert features[1].shape == (1, 192, 80, 80)
    assert features[2].shape == (1, 384, 40, 40)
    assert features[3].shape == (1, 768, 20, 20)

    out_fd = tmp_path  # the reference writes below $VKIT_OPEN_MODEL_DATA through iolite
    torch.save(
        {'model': model_jit.state_dict()},  # type: ignore
        out_fd / 'torch-save-convnext-state-dict.pt',
    )

    dump = torch.load(out_fd / 'torch-save-convnext-state-dict.pt')
    model.load_state_dict(dump['model'])
    # beyond the reference's assertions: eager == scripted bit for bit (eval mode: no stochastic depth), the scripted module
    # trains (gradients land on its own parameters), and it keeps working once the eager module is gone
    model.eval(), model_jit.eval()
    eager = model(x)
    scripted = model_jit(x)
    assert all(torch.equal(a, b) for a, b in zip(eager, scripted))
    sum(f.float().sum() for f in model_jit(x)).backward()
    assert all(p.grad is not None for p in model_jit.parameters())
    del model
    import gc
    gc.collect()
    assert all(torch.equal(a, b) for a, b in zip(eager, model_jit(x)))


def test_scripted_blocks_run_and_match_eager():
    """Every scriptable class of the mirror (model/scripting.py) scripted on its own: ConvNextBlockLayer, ConvNextBlock,
    PpmBlock, UperNextHead - the remaining ones are covered above."""
    from vkit_ocr_model_adaptive_scaling_amd.model.convnext import ConvNextBlockLayer, ConvNextBlock
    from vkit_ocr_model_adaptive_scaling_amd.model.upernext import PpmBlock, UperNextHead
    torch.manual_seed(3)
    cases = [(ConvNextBlockLayer(40, 0.0), torch.rand(2, 40, 12, 20)),
             (ConvNextBlock(0, 5, 24, 2, 48), torch.rand(1, 24, 16, 24)),
             (PpmBlock((1, 2, 3, 6), 72, 20), torch.rand(2, 72, 7, 9)),
             (UperNextHead(64, 2, 2, 0.5), torch.rand(1, 64, 12, 20))]
    for module, x in cases:
        module = module.cuda().eval()
        x = x.cuda()
        jit = torch.jit.script(module)
        want, got = module(x), jit(x)
        if isinstance(want, torch.Tensor):
            want, got = (want,), (got,)
        assert len(want) == len(got) and all(torch.equal(a, b) for a, b in zip(want, got)), type(module).__name__
        # train / eval travels with the scripted module and does not touch the eager one it was scripted from
        jit.train()
        jit(x)
        assert not module.training


# ------------------------------------------------------------------ the numbers: oracle at the reference tests' sizes
@pytest.mark.parametrize('dtype', DTYPES, ids=IDS)
@pytest.mark.parametrize('kind', ['fpn', 'upernext'])
def test_reference_test_sizes_match_oracle(kind, dtype):
    """The configurations of tests/test_fpn.py / test_upernext.py at their own sizes (80 x 80 ... 10 x 10, rand inputs,
    default initialisation), out_channels = 400 -> inner width 100 (a 104-channel activation with 4 pad channels per level, a
    compact 400-channel concatenation): neck output, head outputs (factor 1 and 2) and all gradients against the oracle."""
    from vkit_ocr_model_adaptive_scaling_amd.model import FpnNeck, FpnHead, UperNextNeck, UperNextHead, set_compute_dtype
    neck_cls, head_cls = (FpnNeck, FpnHead) if kind == 'fpn' else (UperNextNeck, UperNextHead)
    torch.manual_seed(11)
    neck = set_compute_dtype(neck_cls((96, 192, 384, 768), 400).cuda().eval(), dtype)
    heads = [set_compute_dtype(head_cls(400, 1, f).cuda().eval(), dtype) for f in (1, 2)]
    feats = [torch.rand(1, c, 80 >> i, 80 >> i) for i, c in enumerate((96, 192, 384, 768))]
    gfeats = [f.clone().cuda().requires_grad_(True) for f in feats]
    out = neck(gfeats)
    assert out.shape == (1, 400, 80, 80)
    houts = [h(out) for h in heads]
    assert houts[0].shape == (1, 1, 80, 80) and houts[1].shape == (1, 1, 160, 160)
    cots = [torch.randn(o.shape) for o in [out] + houts]
    sum((o.float() * c.cuda()).sum() for o, c in zip([out] + houts, cots)).backward()

    def oracle(store):
        sd_n, sd_h = _oracle_sd(neck), [_oracle_sd(h) for h in heads]
        fs = [f.double().requires_grad_(True) for f in feats]
        with _fmt(store):
            o = (O.fpn_neck_forward if kind == 'fpn' else O.upernext_neck_forward)(sd_n, fs)
            hf = O.fpn_head_forward if kind == 'fpn' else O.upernext_head_forward
            ho = [hf(sd, o, '', f) for sd, f in zip(sd_h, (1, 2))]
            sum((a * c.double()).sum() for a, c in zip([o] + ho, cots)).backward()
        grads = {'neck.' + k: v.grad for k, v in sd_n.items()}
        for i, sd in enumerate(sd_h):
            grads.update({f'head{i}.' + k: v.grad for k, v in sd.items()})
        return [o.detach()] + [h.detach() for h in ho], grads, [f.grad for f in fs]

    ref_outs, ref_grads, ref_gf = oracle(torch.float32)
    q_outs, q_grads, q_gf = oracle(dtype) if dtype != torch.float32 else (ref_outs, ref_grads, ref_gf)
    tag = _tag('reference_test_sizes_w400', kind, dtype)
    names = ['neck output (1,400,80,80)', 'head x1', 'head x2']
    for n, got, want, q in zip(names, [out] + houts, ref_outs, q_outs):
        e, fe = rel_err(got, want), rel_err(q, want)
        _rec(tag, n, e, fmt_bound(FWD_TOL[dtype], fe), 'format alone %.3e' % fe if dtype != torch.float32 else '')
        assert e < fmt_bound(FWD_TOL[dtype], fe), (n, e, fe)
    got = {'neck.' + k: p.grad for k, p in neck.named_parameters()}
    for i, h in enumerate(heads):
        got.update({f'head{i}.' + k: p.grad for k, p in h.named_parameters()})
    gmax = max(float(v.norm()) for v in ref_grads.values())
    worst = (0.0, '')
    for k, g in got.items():
        want = ref_grads[k]
        if float(want.norm()) < 1e-6 * gmax:  # e.g. a conv bias in front of a LayerNorm: analytically zero
            assert float(g.double().norm()) < (1e-5 if dtype == torch.float32 else 3e-3) * gmax, k
            continue
        e, fe = rel_err(g, want), rel_err(q_grads[k], want)
        worst = max(worst, (e, k))
        assert e < fmt_bound(GRAD_TOL[dtype], fe), (k, e, fe)
    _rec(tag, 'worst parameter gradient of %d' % len(got), worst[0], GRAD_TOL[dtype], worst[1])
    for i, (g, want, q) in enumerate(zip(gfeats, ref_gf, q_gf)):
        e, fe = rel_err(g.grad, want), rel_err(q, want)
        _rec(tag, f'gradient of feature {i}', e, fmt_bound(GRAD_TOL[dtype], fe))
        assert e < fmt_bound(GRAD_TOL[dtype], fe), (i, e, fe)


# ------------------------------------------------------------------ the numbers: a fixture of the reference at these widths
@pytest.mark.parametrize('dtype', DTYPES, ids=IDS)
@pytest.mark.parametrize('kind', ['fpn', 'upernext'])
def test_width400_matches_reference_fixture(kind, dtype):
    """tests/golden/ref_width400.npz (generated by the imported reference): FpnNeck / UperNextNeck((96,192,384,768), 400)
    with seeded parameters, then FpnHead(400, 1, 1 | 2) / UperNextHead(400, 2, 2) on the reference's neck output."""
    from vkit_ocr_model_adaptive_scaling_amd.model import FpnNeck, FpnHead, UperNextNeck, UperNextHead, set_compute_dtype
    w = recipe.WIDTH400
    g = golden('ref_width400')
    neck_cls, head_cls = (FpnNeck, FpnHead) if kind == 'fpn' else (UperNextNeck, UperNextHead)
    neck = set_compute_dtype(seed_module(neck_cls(w['in_channels_group'], w['out_channels']), w['seed'], w['std']).cuda().eval(),
                             dtype)
    fs = [torch.from_numpy(a).float().cuda().requires_grad_(True) for a in recipe.neck_features(w)]
    out = neck(fs)
    tag = _tag('width400_fixture', kind, dtype)
    qout = qg = None
    if dtype != torch.float32:
        sd = _oracle_sd(neck)
        qf = [torch.from_numpy(a).double().requires_grad_(True) for a in recipe.neck_features(w)]
        with O.storage_rounding(dtype):
            qout = (O.fpn_neck_forward if kind == 'fpn' else O.upernext_neck_forward)(sd, qf)
            (qout * torch.from_numpy(recipe.cotangent(w['seed'], 0, tuple(qout.shape)))).sum().backward()
        qg = {k: v.grad for k, v in sd.items()}
    e = rel_err(out, g[f'{kind}/neck_out'])
    fe = rel_err(qout.detach(), g[f'{kind}/neck_out']) if qout is not None else 0.0
    _rec(tag, 'neck output', e, fmt_bound(FWD_TOL[dtype], fe))
    assert e < fmt_bound(FWD_TOL[dtype], fe), (e, fe)
    (out.float() * cot(w['seed'], 0, out.shape)).sum().backward()
    n = check_grad_summary(named_params(neck), g, tol=GRAD_TOL[dtype], prefix=f'{kind}/neck/', tag=tag, fmt_grads=qg)
    assert n == len(list(neck.parameters()))
    for i, f in enumerate(fs):
        fe = rel_err(qf[i].grad, g[f'{kind}/gfeat{i}']) if qout is not None else 0.0
        assert rel_err(f.grad, g[f'{kind}/gfeat{i}']) < fmt_bound(GRAD_TOL[dtype], fe), i
    ref_out = torch.from_numpy(g[f'{kind}/neck_out']).float().cuda()
    for oc, factor in (((1, 1), (1, 2)) if kind == 'fpn' else ((2, 2),)):
        head = set_compute_dtype(seed_module(head_cls(w['out_channels'], oc, factor), w['seed'] + 10 * oc + factor,
                                             w['head_std']).cuda().eval(), dtype)
        xin = ref_out.clone().requires_grad_(True)
        y = head(xin)
        key = f'{kind}/head_oc{oc}_f{factor}'
        qy = qhg = qgx = None
        if dtype != torch.float32:
            sd = _oracle_sd(head)
            qx = ref_out.double().cpu().requires_grad_(True)
            with O.storage_rounding(dtype):
                qy = (O.fpn_head_forward if kind == 'fpn' else O.upernext_head_forward)(sd, qx, '', factor)
                (qy * torch.from_numpy(recipe.cotangent(w['seed'] + 1, 0, tuple(qy.shape)))).sum().backward()
            qhg, qgx = {k: v.grad for k, v in sd.items()}, qx.grad
        e = rel_err(y, g[key + '/out'])
        fe = rel_err(qy.detach(), g[key + '/out']) if qy is not None else 0.0
        _rec(tag, f'head oc{oc} x{factor} output', e, fmt_bound(FWD_TOL[dtype], fe))
        assert e < fmt_bound(FWD_TOL[dtype], fe), (key, e, fe)
        (y.float() * cot(w['seed'] + 1, 0, y.shape)).sum().backward()
        check_grad_summary(named_params(head), g, tol=GRAD_TOL[dtype], prefix=key + '/', tag=tag, fmt_grads=qhg)
        fe = rel_err(qgx, g[key + '/gx']) if qgx is not None else 0.0
        assert rel_err(xin.grad, g[key + '/gx']) < fmt_bound(GRAD_TOL[dtype], fe), key


@pytest.mark.parametrize('dtype', DTYPES, ids=IDS)
def test_backbone_widths_not_multiple_of_8(dtype):
    """ConvNext(...) with stage widths 20 / 36 / 52 / 100 (activations padded to 24 / 40 / 56 / 104 channels) against the
    oracle: features and all gradients.  The reference puts no constraint on the widths (convnext.py:154-167)."""
    from vkit_ocr_model_adaptive_scaling_amd.model import ConvNext, set_compute_dtype
    plan = ((20, 1), (36, 1), (52, 2), (100, 1))
    m = set_compute_dtype(seed_module(ConvNext(3, plan, False), 77, 0.12).cuda().eval(), dtype)
    x = torch.from_numpy(recipe.image(77, (2, 3, 64, 96))).float()
    feats = m(x.cuda())
    assert [tuple(f.shape) for f in feats] == [(2, 20, 16, 24), (2, 36, 8, 12), (2, 52, 4, 6), (2, 100, 2, 3)]
    cots = [torch.from_numpy(recipe.cotangent(77, i, tuple(f.shape))) for i, f in enumerate(feats)]
    sum((f.float() * c.float().cuda()).sum() for f, c in zip(feats, cots)).backward()

    def oracle(store):
        sd = _oracle_sd(m)
        with _fmt(store):
            fo = O.convnext_forward(sd, x.double())
            sum((f * c).sum() for f, c in zip(fo, cots)).backward()
        return [f.detach() for f in fo], {k: v.grad for k, v in sd.items()}

    ref_f, ref_g = oracle(torch.float32)
    q_f, q_g = oracle(dtype) if dtype != torch.float32 else (ref_f, ref_g)
    tag = _tag('backbone_odd_widths', dtype)
    for i, (f, want, q) in enumerate(zip(feats, ref_f, q_f)):
        e, fe = rel_err(f, want), rel_err(q, want)
        _rec(tag, f'feature {i}', e, fmt_bound(FWD_TOL[dtype], fe))
        assert e < fmt_bound(FWD_TOL[dtype], fe), (i, e, fe)
    worst = (0.0, '')
    for k, p in m.named_parameters():
        e, fe = rel_err(p.grad, ref_g[k]), rel_err(q_g[k], ref_g[k])
        worst = max(worst, (e, k))
        assert e < fmt_bound(GRAD_TOL[dtype], fe), (k, e, fe)
    _rec(tag, 'worst parameter gradient', worst[0], GRAD_TOL[dtype], worst[1])
