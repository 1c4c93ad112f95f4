"""CPU-side checks: the C-ABI library loads and exports every symbol declared in include/vkas.h (no compute
calls without a GPU), the module mirror reproduces the reference's state-dict schema, host logic (LR rule, flat
buffers, bucket layout) and the product path refuses to run without the GPU."""
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'vkas.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(vkas_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    from vkit_ocr_model_adaptive_scaling_amd import _lib
    syms = declared_symbols()
    assert len(syms) >= 35
    for s in syms:
        assert hasattr(_lib.lib, s), f'{s} declared in include/vkas.h but not exported by libvkas.so'
    assert set(_lib.EXPORTS) == set(syms), set(_lib.EXPORTS) ^ set(syms)
    assert _lib.lib.vkas_abi_version() == 1


def test_argument_validation_without_gpu():
    """Entry points validate shapes before touching the device: bad arguments come back as error codes + message."""
    import ctypes
    from vkit_ocr_model_adaptive_scaling_amd import _lib
    g = _lib.ConvGeom(1, 4, 4, 4, 4, 12, 12, 1, 1, 1, 0)  # Cp not a multiple of 8
    e = _lib.Epilogue()
    rc = _lib.lib.vkas_conv_gemm_fwd(ctypes.c_void_p(16), ctypes.byref(g), ctypes.c_void_p(16), 8, ctypes.byref(e), 0, None)
    assert rc == -1 and b'multiple of 8' in _lib.lib.vkas_last_error()
    with pytest.raises(_lib.VkasError):
        _lib.check(rc, 'conv')
    assert _lib.lib.vkas_layernorm_fwd(None, 8, None, None, None, 8, None, 4, 8, 8, 0, 0, None) == -1


@pytest.mark.parametrize('kind', ['upernext', 'fpn'])
def test_state_dict_schema_matches_reference(kind):
    from vkit_ocr_model_adaptive_scaling_amd.model import (AdaptiveScaling, AdaptiveScalingConfig, AdaptiveScalingSize,
                                                           AdaptiveScalingNeckHeadType)
    g = np.load(os.path.join(ROOT, 'tests', 'golden', f'full_tiny_{kind}_256.npz'))
    enum = AdaptiveScalingNeckHeadType.UPERNEXT if kind == 'upernext' else AdaptiveScalingNeckHeadType.FPN
    sd = AdaptiveScaling(AdaptiveScalingConfig(AdaptiveScalingSize.TINY, enum)).state_dict()
    assert list(sd.keys()) == list(g['state_dict_keys'])
    assert [str(tuple(v.shape)) for v in sd.values()] == list(g['state_dict_shapes'])
    assert all(v.dtype == torch.float32 for v in sd.values())


def test_config_defaults_and_errors():
    from vkit_ocr_model_adaptive_scaling_amd.model import (AdaptiveScalingConfig, AdaptiveScalingSize,
                                                           AdaptiveScalingNeckHeadType, UperNextNeck, FpnHead, ConvNext)
    from vkit_ocr_model_adaptive_scaling_amd.loss_function import (AdaptiveScalingRoughLossFunctionConifg,
                                                                   AdaptiveScalingPreciseLossFunctionConifg,
                                                                   AdaptiveScalingRoughLossFunction)
    c = AdaptiveScalingConfig()
    assert (c.size, c.neck_head_type, c.rough_upsampling_factor, c.rough_init_char_height_output_bias,
            c.precise_upsampling_factor, c.precise_enable_char_mask_head) == (
        AdaptiveScalingSize.SMALL, AdaptiveScalingNeckHeadType.FPN, 2, 8.0, 2, False)
    r, p = AdaptiveScalingRoughLossFunctionConifg(), AdaptiveScalingPreciseLossFunctionConifg()
    assert (r.bce_factor, r.focal_factor, r.dice_factor, r.l1_factor, r.downsampled_score_map_min) == (0.0, 5.0, 1.0, 1.0, 1.1)
    assert (p.char_prob_pos_l2_factor, p.char_prob_neg_l2_factor, p.char_corner_angle_cross_entropy_factor,
            p.loss_factor) == (2.0, 1.0, 5.0, 0.15)
    with pytest.raises(AssertionError):
        UperNextNeck((16, 32, 64), 64)  # 64 % 3 != 0 (upernext.py:144)
    with pytest.raises(NotImplementedError):
        FpnHead(64, 1, upsampling_factor=5)  # fpn.py:176
    with pytest.raises(NotImplementedError):
        AdaptiveScalingRoughLossFunction(AdaptiveScalingRoughLossFunctionConifg(bce_factor=1.0))
    m = ConvNext.create_tiny()
    assert m.in_channels_group == [96, 192, 384, 768]
    probs = [l.prob_bypass for b in m.blocks for l in b.layers]
    assert probs[0] == 0.0 and abs(probs[-1] - 0.1) < 1e-12 and abs(probs[1] - 0.1 / 17) < 1e-12
    # reference initialisation: layer scale 1e-6, head bias constant (convnext.py:38, upernext.py:231)
    assert float(m.blocks[0].layers[0].block_scale.max()) == pytest.approx(1e-6)


def test_product_path_refuses_cpu():
    from vkit_ocr_model_adaptive_scaling_amd.model import ConvNext
    m = ConvNext(3, ((16, 1), (32, 1)), False)
    with pytest.raises(RuntimeError, match='MI355X'):
        m(torch.zeros(1, 3, 32, 32))


def test_product_never_imports_oracle():
    import subprocess, sys
    code = ('import sys; import vkit_ocr_model_adaptive_scaling_amd.model, vkit_ocr_model_adaptive_scaling_amd.loss_function, '
            'vkit_ocr_model_adaptive_scaling_amd.training; '
            'bad=[m for m in sys.modules if m == "oracle" or m.startswith("oracle.")]; sys.exit(1 if bad else 0)')
    assert subprocess.run([sys.executable, '-c', code], cwd=ROOT).returncode == 0
    for dirpath, _, files in os.walk(os.path.join(ROOT, 'vkit_ocr_model_adaptive_scaling_amd')):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dirpath, f)).read()
                assert 'import oracle' not in src and 'from oracle' not in src, f


def test_lr_rule_matches_torch_scheduler():
    import warnings
    from vkit_ocr_model_adaptive_scaling_amd.training import cosine_warm_restarts_lr
    opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=8e-4)
    sch = torch.optim.lr_scheduler.CosineAnnealingWarmRestarts(opt, T_0=10, T_mult=10, eta_min=8e-6)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        for e in [0, 0.25, 3.999, 9.999, 10, 10.5, 64.25, 109.999, 110, 500.125]:
            sch.step(e)
            assert abs(sch.get_last_lr()[0] - cosine_warm_restarts_lr(e, 8e-4, 8e-6, 10, 10)) < 1e-12


def test_flat_buffers_views_and_ranges():
    from vkit_ocr_model_adaptive_scaling_amd.training import FlatBuffers
    m = torch.nn.Sequential(torch.nn.Linear(5, 3), torch.nn.Linear(3, 7))
    before = {n: p.detach().clone() for n, p in m.named_parameters()}
    fb = FlatBuffers(m.named_parameters())
    for n, p in m.named_parameters():
        assert torch.equal(p.detach(), before[n])
        s, k = fb.offsets[n]
        assert s % 4 == 0 and p.data_ptr() == fb.flat_param.data_ptr() + 4 * s
    m(torch.ones(2, 5)).sum().backward()
    m(torch.ones(2, 5)).sum().backward()  # accumulates in place into the flat buffer
    s, k = fb.offsets['1.bias']
    assert torch.equal(fb.flat_grad[s:s + k], torch.full((7,), 4.0))
    assert fb.range_of(('0.',)) == (0, fb.offsets['1.weight'][0])
    with pytest.raises(ValueError):
        fb.range_of(('0.weight', '1.weight'))
    fb.zero_grad()
    assert float(fb.flat_grad.abs().sum()) == 0.0


def test_bucket_layout_for_adaptive_scaling():
    from vkit_ocr_model_adaptive_scaling_amd.model import (AdaptiveScaling, AdaptiveScalingConfig, AdaptiveScalingSize,
                                                           AdaptiveScalingNeckHeadType)
    from vkit_ocr_model_adaptive_scaling_amd.training import FlatBuffers, BucketedGradReducer, adaptive_scaling_buckets
    model = AdaptiveScaling(AdaptiveScalingConfig(AdaptiveScalingSize.TINY, AdaptiveScalingNeckHeadType.UPERNEXT))
    fb = FlatBuffers(model.named_parameters())
    red = BucketedGradReducer(fb, adaptive_scaling_buckets(model))
    assert list(red.buckets) == ['rough', 'precise', 'backbone3', 'backbone2', 'backbone1', 'backbone0']
    spans = sorted((b.start, b.end) for b in red.buckets.values())
    assert spans[0][0] == 0 and spans[-1][1] == fb.numel
    assert all(a[1] == b[0] for a, b in zip(spans, spans[1:])), 'buckets must tile the flat buffer'
    n_rough = sum(p.numel() for n, p in model.named_parameters() if n.startswith('rough_'))
    assert red.buckets['rough'].end - red.buckets['rough'].start >= n_rough


def test_flat_buffers_touched_ranges_and_invalidation():
    """Parameters that never got a gradient are left out of the AdamW launch ranges (torch.optim.AdamW skips
    .grad is None); load_flat() drops the packed-weight cache (writes behind the version counter)."""
    from vkit_ocr_model_adaptive_scaling_amd.training import FlatBuffers
    from vkit_ocr_model_adaptive_scaling_amd import ops
    m = torch.nn.ModuleDict({'a': torch.nn.Linear(5, 3), 'unused': torch.nn.Linear(3, 3), 'b': torch.nn.Linear(3, 7)})
    fb = FlatBuffers(m.named_parameters())
    assert fb.touched_ranges() == []
    m['b'](m['a'](torch.ones(2, 5))).sum().backward()
    a0, b0 = fb.offsets['a.weight'][0], fb.offsets['b.weight'][0]
    u0 = fb.offsets['unused.weight'][0]
    assert fb.touched_ranges() == [(a0, u0), (b0, fb.numel)]
    fb.zero_grad()
    assert fb.touched_ranges() == []
    ops._PACK_CACHE[('sentinel', 0)] = None
    epoch = ops._PACK_EPOCH[0]
    fb.load_flat(torch.arange(fb.numel, dtype=torch.float32))
    assert not ops._PACK_CACHE and ops._PACK_EPOCH[0] == epoch + 1
    assert float(m['a'].weight.reshape(-1)[1]) == 1.0
    with pytest.raises(ValueError):
        fb.load_flat(torch.zeros(3))


def test_loss_exports_and_shape_contract():
    """loss_function/__init__.py:12-24 exports, and the shape contract the fused loss ops enforce before any kernel
    indexes with the numbers (pure host logic: runs on CPU tensors)."""
    from vkit_ocr_model_adaptive_scaling_amd import loss_function as L
    from vkit_ocr_model_adaptive_scaling_amd import ops
    for name in ('WeightedBceWithLogitsLossFunction', 'CrossEntropyWithLogitsLossFunction', 'FocalWithLogitsLossFunction',
                 'L1LossFunction', 'L2LossFunction', 'WeightAdaptiveHeatmapRegressionLossFunction', 'DiceLossFunction',
                 'AdaptiveScalingRoughLossFunctionConifg', 'AdaptiveScalingRoughLossFunction',
                 'AdaptiveScalingPreciseLossFunctionConifg', 'AdaptiveScalingPreciseLossFunction'):
        assert hasattr(L, name), name
    f = L.FocalWithLogitsLossFunction()
    assert (f.alpha, f.gamma, f.eps) == (0.25, 2, 1e-6)
    l1 = L.L1LossFunction()
    assert (l1.eps, l1.smooth, l1.smooth_beta) == (1e-6, False, 1.0)
    assert L.DiceLossFunction().eps == 1e-6 and L.L2LossFunction().eps == 1e-6
    with pytest.raises(NotImplementedError):
        L.WeightedBceWithLogitsLossFunction()
    with pytest.raises(RuntimeError, match='MI355X'):
        L.L2LossFunction()(torch.zeros(4), torch.zeros(4))  # no CPU fallback
    z = torch.zeros
    ops._check_loss_maps('t', (z(2, 1, 8, 9), z(2, 1, 8, 9)), (1, 1), z(2, 4, 5), z(2, 4, 5), 2, 2)
    with pytest.raises(ValueError):
        ops._check_loss_maps('t', (z(2, 1, 8, 9), z(2, 2, 8, 9)), (1, 1), z(2, 4, 5), z(2, 4, 5), 2, 2)
    with pytest.raises(ValueError):
        ops._check_loss_maps('t', (z(2, 1, 8, 9),), (1,), z(3, 4, 5), z(3, 4, 5), 2, 2)  # batch mismatch
    with pytest.raises(ValueError):
        ops._check_loss_maps('t', (z(2, 1, 8, 9),), (1,), z(2, 4, 5), z(2, 4, 6), 2, 2)  # score map != mask
    with pytest.raises(ValueError):
        ops._check_loss_maps('t', (z(2, 1, 8, 9),), (1,), z(2, 7, 5), z(2, 7, 5), 2, 2)  # crop leaves the map


def test_restore_state_roundtrip_and_torch_adamw_compat(tmp_path):
    """RestoreState files (train.py:91-96,599-605): our optimizer export loads into a real torch.optim.AdamW, a real
    AdamW / CosineAnnealingWarmRestarts state dict loads into the flat moment buffers, and the file round-trips."""
    import types
    import warnings
    from vkit_ocr_model_adaptive_scaling_amd.training import (FlatBuffers, save_restore_state, load_restore_state,
                                                              optimizer_state_dict, load_optimizer_state_dict,
                                                              scheduler_state_dict)
    torch.manual_seed(0)
    mk = lambda: torch.nn.Sequential(torch.nn.Linear(5, 3), torch.nn.Linear(3, 2))
    ref = mk()
    topt = torch.optim.AdamW(ref.parameters(), lr=8e-4, weight_decay=0.01)
    sch = torch.optim.lr_scheduler.CosineAnnealingWarmRestarts(topt, T_0=10, T_mult=10, eta_min=8e-6)
    for i in range(3):
        ref(torch.ones(2, 5) * (i + 1)).sum().backward()
        topt.step()
        topt.zero_grad()
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            sch.step(12.25 + i)
    m = mk()
    fb = FlatBuffers(m.named_parameters())
    opt = types.SimpleNamespace(flat=fb, exp_avg=torch.zeros(fb.numel), exp_avg_sq=torch.zeros(fb.numel), step_count=0,
                                lr=1.0, betas=(0.5, 0.5), eps=1.0, weight_decay=0.5)  # stand-in with FlatAdamW's fields
    load_optimizer_state_dict(opt, topt.state_dict())
    assert opt.step_count == 3 and opt.lr == topt.param_groups[0]['lr'] and opt.weight_decay == 0.01
    for i, p in enumerate(ref.parameters()):
        s, k = fb.offsets[fb.names[i]]
        assert torch.equal(opt.exp_avg[s:s + k], topt.state[p]['exp_avg'].reshape(-1))
        assert torch.equal(opt.exp_avg_sq[s:s + k], topt.state[p]['exp_avg_sq'].reshape(-1))
    # export -> a fresh torch AdamW accepts it
    t2 = torch.optim.AdamW(mk().parameters(), lr=1.0)
    t2.load_state_dict(optimizer_state_dict(opt))
    assert t2.param_groups[0]['weight_decay'] == 0.01 and float(list(t2.state.values())[0]['step']) == 3.0
    # scheduler export matches the real scheduler's state after step(14.25)
    ours, real = scheduler_state_dict(14.25, 8e-4, 8e-6, 10, 10), sch.state_dict()
    for k in ('T_0', 'T_i', 'T_mult', 'eta_min', 'base_lrs', 'last_epoch'):
        assert ours[k] == real[k], k
    assert abs(ours['T_cur'] - real['T_cur']) < 1e-12 and abs(ours['_last_lr'][0] - real['_last_lr'][0]) < 1e-12
    # file round trip (weights_only load), parameters restored into the flat views
    path = tmp_path / 'state_dict_7.pt'
    save_restore_state(path, 7, ref, opt, ours)
    rs = load_restore_state(path, m, opt)
    assert rs.epoch_idx == 7 and set(rs.model_jit_state_dict) == set(ref.state_dict())
    for (n, p), (_, q) in zip(m.named_parameters(), ref.named_parameters()):
        assert torch.equal(p, q) and p.data_ptr() == fb.flat_param.data_ptr() + 4 * fb.offsets[n][0]
