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
    # fused MLP with LayerNorm: 16-bit storage only, LayerNorm parameters required, the backward operands travel together
    P = ctypes.c_void_p
    a = lambda: P(256)  # any aligned non-null address: the checks run before anything is dereferenced or launched
    ln = _lib.lib.vkas_mlp_chain_ln_fwd
    assert ln(a(), 96, a(), a(), None, 96, None, a(), a(), a(), 96, a(), None, 64, None, 384, None, 96, a(), 96, 128, 96,
              _lib.F32, None) == -1 and b'16-bit' in _lib.lib.vkas_last_error()
    assert ln(a(), 96, None, a(), None, 96, None, a(), a(), a(), 96, a(), None, 64, None, 384, None, 96, a(), 96, 128, 96,
              _lib.BF16, None) == -1 and b'LayerNorm parameter' in _lib.lib.vkas_last_error()
    assert ln(a(), 96, a(), a(), a(), 96, None, a(), a(), a(), 96, a(), None, 64, None, 384, None, 96, a(), 96, 128, 96,
              _lib.BF16, None) == -1 and b'together' in _lib.lib.vkas_last_error()
    assert ln(a(), 96, a(), a(), None, 96, None, a(), a(), a(), 96, a(), None, 64, None, 384, None, 96, a(), 96, 128, 100,
              _lib.BF16, None) == -1 and b'not covered' in _lib.lib.vkas_last_error()
    assert ln(a(), 96, a(), a(), None, 96, None, a(), a(), a(), 96, a(), None, 64, None, 384, None, 96, a(), 96, 0, 96,
              _lib.BF16, None) == 0  # M = 0: nothing to do, nothing launched


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


def test_metrics_window_mean_matches_reference_update_rule():
    """training/metrics.py:36-55 updates the window mean incrementally (avg + (new - popped) / n once the queue is full);
    the mirror recomputes it from the window.  Same values, and reset() is per tag."""
    from enum import Enum
    from vkit_ocr_model_adaptive_scaling_amd.training import Metrics

    class Tag(Enum):
        A = 'a'
        B = 'b'
    m = Metrics(Tag, avg_num_batches=4)
    rng = np.random.default_rng(0)
    vals = rng.random(11).tolist()
    queue, avg = [], None
    for v in vals:
        if not queue:
            avg = v
        elif len(queue) < 4:
            avg = (avg * len(queue) + v) / (len(queue) + 1)
        else:
            avg = avg + (v - queue.pop(0)) / 4
        queue.append(v)
        got = m.update(Tag.A, v)
        assert abs(got - avg) < 1e-12 and abs(got - sum(queue) / len(queue)) < 1e-15
    assert m.tag_to_avg_value[Tag.B] is None and m.update(Tag.B, 2.0) == 2.0
    m.reset([Tag.A])
    assert m.tag_to_avg_value[Tag.A] is None and m.tag_to_avg_value[Tag.B] == 2.0
    assert m.update(Tag.A, 5.0) == 5.0


def test_collate_schema_and_synthetic_dataset():
    """dataset/adaptive_scaling.py:282-368: keys, dtypes and shapes of the collated batch; the synthetic source is
    index-deterministic (the same sample whatever the worker layout) and feeds a torch DataLoader."""
    from torch.utils.data import DataLoader
    from vkit_ocr_model_adaptive_scaling_amd.dataset import (SyntheticAdaptiveScalingIterableDataset,
                                                             adaptive_scaling_dataset_collate_fn)
    ds = SyntheticAdaptiveScalingIterableDataset(6, (96, 128), num_label_points=7, margin=10, rng_seed=3)
    batches = list(DataLoader(ds, batch_size=3, collate_fn=adaptive_scaling_dataset_collate_fn))
    assert len(batches) == 2
    b = batches[0]
    assert set(b) == {'rough', 'precise'}
    common = {'image': ((3, 3, 96, 128), torch.float32), 'downsampled_mask': ((3, 28, 44), torch.float32),
              'downsampled_score_map': ((3, 28, 44), torch.float32)}
    extra = {'downsampled_label_point_y': ((3, 7), torch.int64), 'downsampled_label_point_x': ((3, 7), torch.int64),
             'up_left_offsets': ((3, 7, 2), torch.int64), 'corner_angles': ((3, 7, 4), torch.float32),
             'corner_distances': ((3, 7, 3), torch.float32)}
    for part, spec in (('rough', common), ('precise', {**common, **extra})):
        assert set(b[part]) == set(spec) | {'downsampled_shape', 'downsampled_core_box', 'rng_states'}
        for k, (shape, dt) in spec.items():
            assert tuple(b[part][k].shape) == shape and b[part][k].dtype == dt, (part, k)
        assert b[part]['downsampled_shape'] == (48, 64) and len(b[part]['rng_states']) == 3
        box = b[part]['downsampled_core_box']
        assert (box.up, box.down, box.left, box.right) == (10, 37, 10, 53)
    p = b['precise']
    assert float(p['image'].max()) <= 255.0 and float(p['image'].min()) >= 0.0
    assert int(p['downsampled_label_point_y'].min()) >= 10 and int(p['downsampled_label_point_y'].max()) <= 37
    assert torch.allclose(p['corner_angles'].sum(-1), torch.ones(3, 7), atol=1e-6)
    r0, p0 = ds.sample(4)
    r1, p1 = SyntheticAdaptiveScalingIterableDataset(6, (96, 128), num_label_points=7, rng_seed=3).sample(4)
    assert np.array_equal(r0.image, r1.image) and np.array_equal(p0.up_left_offsets, p1.up_left_offsets)
    two = list(DataLoader(ds, batch_size=3, num_workers=0, collate_fn=adaptive_scaling_dataset_collate_fn))
    assert torch.equal(two[1]['rough']['image'], batches[1]['rough']['image'])


def test_run_training_loop_rules(tmp_path):
    """The epoch loop's host logic (train.py:340-605) with a stand-in step: learning rate per step = the scheduler value
    set AFTER the previous step, dev means, and the checkpoint rule / file names."""
    import types
    from vkit_ocr_model_adaptive_scaling_amd.training import (EpochConfig, OptimizerConfig, FlatBuffers, run_training,
                                                              cosine_warm_restarts_lr, load_restore_state)

    class Toy(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.lin = torch.nn.Linear(2, 1)

        def forward_rough(self, x):
            return x.mean(), x.mean()

        def forward_precise(self, x):
            return x.mean(), x.mean(), x.mean(), x.mean()
    model = Toy()
    fb = FlatBuffers(model.named_parameters())
    opt = types.SimpleNamespace(flat=fb, exp_avg=torch.zeros(fb.numel), exp_avg_sq=torch.zeros(fb.numel), step_count=0,
                                lr=8e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    lrs, dev_values = [], iter([3.0, 1.0, 2.0, 2.5])  # dev loss per epoch (rough == precise == value / 2 each)

    class Step:
        def __init__(self):
            self.model, self.optimizer = model, opt
            self.cur = None

        def __call__(self, rough, precise, lr=None):
            lrs.append(lr)
            opt.step_count += 1
            return torch.tensor(0.25), torch.tensor(0.75)

        def _rough_loss(self, outs, b, scale):
            return torch.tensor(self.cur * scale)

        def _precise_loss(self, outs, b, scale):
            return torch.tensor(self.cur * scale)
    step = Step()
    batch = {'rough': {'image': torch.zeros(1, 3, 4, 4)}, 'precise': {'image': torch.zeros(1, 3, 4, 4)}}

    def dev():
        step.cur = next(dev_values)
        return [batch, batch]
    ec = EpochConfig(num_epochs=4, train_num_batches=3, dev_num_batches=2, avg_num_batches=2)
    oc = OptimizerConfig()
    res = run_training(step, lambda e: [batch] * 5, dev, ec, oc, str(tmp_path), torch.device('cpu'),
                       dataset_switch_epochs=(3,))
    # 3 batches per epoch even though the loader offers 5; first step at the base rate, then the rate of the previous call
    rule = lambda t: cosine_warm_restarts_lr(t, 8e-4, 8e-6, 10, 10)
    want = [rule(0.0)] + [rule(e + (b - 1) / 3) for e in range(4) for b in range(1, 4)][:-1]
    assert len(lrs) == 12 and all(abs(a - b) < 1e-15 for a, b in zip(lrs, want))
    assert [round(r.dev_loss, 6) for r in res] == [3.0, 1.0, 2.0, 2.5]
    assert [r.dev_rough_loss for r in res] == [1.5, 0.5, 1.0, 1.25]
    names = [None if r.state_dict_path is None else os.path.basename(r.state_dict_path) for r in res]
    # epoch 0 best, epoch 1 best, epoch 2 saved because a dataset switch follows (epoch 3), epoch 3 is the last
    assert names == ['state_dict_0.pt', 'state_dict_1.pt', 'state_dict_2_not_best.pt', 'state_dict_3_not_best.pt']
    rs = load_restore_state(res[1].state_dict_path)
    assert rs.epoch_idx == 1 and set(rs.model_jit_state_dict) == {'lin.weight', 'lin.bias'}
    sd = rs.optimizer_scheduler_state_dict  # the scheduler's state after its last call of epoch 1: step(1 + 2/3)
    assert sd['last_epoch'] == 1 and abs(sd['T_cur'] - (1 + 2 / 3)) < 1e-12


def test_torch_jit_script_owns_the_reference_schema_and_refuses_the_cpu(tmp_path):
    """train.py:277-280 scripts the model on the host, then moves it: torch.jit.script(model) must succeed without a GPU, the
    scripted module must carry the reference's state-dict keys (train.py:599 saves model_jit.state_dict()) on the SAME
    tensors as the eager module, survive torch.jit.save / load, and its forward must fail loudly on a CPU tensor."""
    from vkit_ocr_model_adaptive_scaling_amd.model import (AdaptiveScaling, AdaptiveScalingConfig, AdaptiveScalingSize,
                                                           AdaptiveScalingNeckHeadType, scripting)
    model = AdaptiveScaling(AdaptiveScalingConfig(AdaptiveScalingSize.TINY, AdaptiveScalingNeckHeadType.UPERNEXT))
    jit = torch.jit.script(model)
    assert list(jit.state_dict()) == list(model.state_dict())
    assert all(a.data_ptr() == b.data_ptr() for a, b in zip(jit.parameters(), model.parameters()))
    assert 'vkas::adaptive_scaling_forward' in str(jit.forward_rough.graph) + str(jit.forward_precise.graph)
    with pytest.raises(RuntimeError, match='MI355X only'):
        jit.forward_rough(torch.zeros(1, 3, 32, 32))
    path = str(tmp_path / 'model_jit.pt')
    torch.jit.save(jit, path)
    del jit, model
    loaded = torch.jit.load(path)
    assert len(loaded.state_dict()) == 304 and loaded._script_spec.count('upernext') == 1
    # the operator's kernel binds the loaded tensors into a parameter-less skeleton of the eager module for the duration of a
    # call (no copy) and unbinds them afterwards: nothing keeps the parameters alive once the scripted module is gone
    params = list(loaded.parameters())
    with scripting._Bound('test', params, loaded._script_spec, False) as eager:
        assert all(a is b for a, b in zip(eager.parameters(), params))
        assert eager.compute_dtype == torch.bfloat16 and not eager.training
    assert all(p.is_meta for p in eager.parameters())
    with pytest.raises(RuntimeError, match='parameter tensors'):
        scripting._Bound('test', params[:-1], loaded._script_spec, False)


def test_sub_modules_script_on_the_host_with_the_reference_schema():
    """tests/test_convnext.py:53-63, test_fpn.py:30,49, test_upernext.py:30 of the reference script the backbone, the necks and
    the heads on their own: every scriptable class of the mirror compiles without a GPU, keeps the reference's state-dict
    keys on the eager module's tensors, and its compiled forward is ONE call of vkas::module_forward (which fails loudly on a
    CPU tensor; running it is tests/test_gpu_reference_tests.py)."""
    import json
    from vkit_ocr_model_adaptive_scaling_amd.model import ConvNext, FpnNeck, FpnHead, UperNextNeck, UperNextHead
    from vkit_ocr_model_adaptive_scaling_amd.model.convnext import ConvNextBlock, ConvNextBlockLayer
    from vkit_ocr_model_adaptive_scaling_amd.model.upernext import PpmBlock
    from vkit_ocr_model_adaptive_scaling_amd.model import scripting
    cases = [ConvNext.create_tiny(stem_use_pconv2x2=True), FpnNeck((96, 192, 384, 768), 400), FpnHead(400, 1, 2),
             UperNextNeck((96, 192, 384, 768), 384), UperNextHead(64, 2, 2, 0.5), ConvNextBlock(0, 5, 24, 2, 48),
             ConvNextBlockLayer(40, 0.05), PpmBlock((1, 2, 3, 6), 72, 20)]
    for m in cases:
        jit = torch.jit.script(m)
        name = type(m).__name__
        assert list(jit.state_dict()) == list(m.state_dict()), name
        assert all(a.data_ptr() == b.data_ptr() for a, b in zip(jit.parameters(), m.parameters())), name
        assert 'vkas::module_forward' in str(jit.forward.graph), name
        spec = json.loads(jit._script_spec)
        assert spec['cls'] == name and spec['compute_dtype'] == 'bf16'
        # the recipe rebuilds the same structure (parameter names and shapes) on the meta device
        skeleton, slots = scripting._skeleton(jit._script_spec)
        assert [(n, tuple(p.shape)) for _, _, p, n in slots] == [(n, tuple(p.shape)) for n, p in m.named_parameters()], name
    x = torch.zeros(1, 3, 64, 64)
    with pytest.raises(RuntimeError, match='MI355X only'):
        torch.jit.script(cases[0])(x)
    scripting.clear()


def test_necks_accept_the_reference_tests_widths():
    """tests/test_fpn.py:16-28 builds FpnNeck((96, 192, 384, 768), out_channels=400); the reference only asks for
    out_channels % len(levels) == 0 (fpn.py:75, upernext.py:144)."""
    from vkit_ocr_model_adaptive_scaling_amd.model import FpnNeck, UperNextNeck
    assert FpnNeck((96, 192, 384, 768), 400).inner_channels == 100
    assert UperNextNeck((96, 192, 384, 768), 400).inner_channels == 100
    with pytest.raises(AssertionError):
        FpnNeck((96, 192, 384, 768), 402)
    with pytest.raises(AssertionError):
        UperNextNeck((96, 192, 384, 768), 402)
