"""torch.jit.script of the AdaptiveScaling mirror on the MI355X: the reference's caller scripts the model
before training (train.py:277-280), checkpoints model_jit.state_dict() (train.py:599) and the inference class loads a
TorchScript file (inferencing/adaptive_scaling.py:85-90).  Scripted forward_rough / forward_precise must equal the eager
methods bit for bit on the config #1 recipe (1 x 3 x 256 x 256, golden parameters), train through autograd, and keep working
after `del model` and after a torch.jit.save / load round trip."""
import numpy as np
import pytest
import torch

from tests.golden import recipe
from tests.helpers import golden, rel_err
from vkit_ocr_model_adaptive_scaling_amd.utils import portable_rng as prng

pytestmark = pytest.mark.gpu


def _model(kind, dtype):
    from vkit_ocr_model_adaptive_scaling_amd.model import (AdaptiveScaling, AdaptiveScalingConfig, AdaptiveScalingSize,
                                                           AdaptiveScalingNeckHeadType)
    enum = AdaptiveScalingNeckHeadType.UPERNEXT if kind == 'upernext' else AdaptiveScalingNeckHeadType.FPN
    c = recipe.FULL_MODEL
    m = AdaptiveScaling(AdaptiveScalingConfig(AdaptiveScalingSize.TINY, enum), compute_dtype=dtype)
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    vals = prng.fill_state_dict(shapes, c['seed'], std=c['std'])
    m.load_state_dict({k: torch.from_numpy(v).float() for k, v in vals.items()})
    t = recipe.full_model_inputs(c)
    return m, torch.from_numpy(t['image_rough']).cuda(), torch.from_numpy(t['image_precise']).cuda()


@pytest.mark.parametrize('kind', ['upernext', 'fpn'])
def test_scripted_forward_equals_eager_bit_for_bit_and_the_golden(kind):
    m, xr, xp = _model(kind, torch.float32)
    jit = torch.jit.script(m)          # on the host, as train.py:278 does
    jit = jit.to('cuda')               # train.py:279 (the eager module shares the tensors and moves with it)
    assert next(m.parameters()).is_cuda
    m.eval(), jit.eval()
    g = golden(f'full_tiny_{kind}_256')
    with torch.no_grad():
        er, ep = m.forward_rough(xr), m.forward_precise(xp)
        jr, jp = jit.forward_rough(xr), jit.forward_precise(xp)
    assert len(jr) == 2 and len(jp) == 4
    for a, b in zip(er + ep, jr + jp):
        assert a.shape == b.shape and torch.equal(a, b)
    # ... and hence the reference fixture (fp32 mode: north-star bound 1e-3)
    names = ['rough_mask', 'rough_height', 'precise_prob', 'precise_offset', 'precise_angle', 'precise_dist']
    for n, t in zip(names, jr + jp):
        assert rel_err(t, g[n]) < 1e-3, n


def test_scripted_module_trains_like_the_eager_one_after_del_model(tmp_path):
    """Gradients through the scripted methods land on the scripted module's parameters and equal the eager ones;
    the reference deletes the eager module right after scripting (train.py:280), so the operator's kernel must serve the
    scripted module from its own parameter tensors; the saved file reloads and runs (inferencing/adaptive_scaling.py:85-90)."""
    import gc
    m, x, _ = _model('upernext', torch.bfloat16)
    m.cuda().eval()

    def run(mod):
        mod.zero_grad()
        r = mod.forward_rough(x)
        p = mod.forward_precise(x)
        loss = sum((t.float() * torch.linspace(-1.0, 1.0, t.numel(), device=t.device).view_as(t)).sum() for t in r + p)
        loss.backward()
        return [t.detach().clone() for t in r + p], [q.grad.detach().clone() for q in mod.parameters() if q.grad is not None]

    eo, eg = run(m)
    jit = torch.jit.script(m)
    jo, jg = run(jit)
    assert len(eg) == len(jg) > 250
    # outputs bit for bit; gradients up to the summation order of the weight-gradient kernels' fp32 atomics (two eager runs
    # differ by as much)
    same_grads = lambda a, b: all(rel_err(x, y) < 1e-5 for x, y in zip(a, b))
    assert all(torch.equal(a, b) for a, b in zip(eo, jo)) and same_grads(jg, eg)
    sd = {k: v.detach().clone() for k, v in jit.state_dict().items()}
    path = str(tmp_path / 'model_jit.pt')
    torch.jit.save(jit, path)
    del m
    gc.collect()
    jo2, jg2 = run(jit)   # the eager module is gone: the kernel rebuilt one around jit's tensors
    assert all(torch.equal(a, b) for a, b in zip(eo, jo2)) and same_grads(jg2, eg)
    loaded = torch.jit.load(path, map_location='cuda')
    loaded.eval()
    assert all(torch.equal(sd[k], v) for k, v in loaded.state_dict().items())
    with torch.no_grad():
        lo = list(loaded.forward_rough(x)) + list(loaded.forward_precise(x))
    assert all(torch.equal(a, b) for a, b in zip(eo, lo))
    # train / eval travels with the scripted module (stochastic depth is active in train mode only)
    jit.train()
    with torch.no_grad():
        t1 = jit.forward_rough(torch.cat([x] * 4))
    jit.eval()
    with torch.no_grad():
        t2 = jit.forward_rough(torch.cat([x] * 4))
    assert torch.equal(t2[0][0], eo[0][0])
    assert t1[0].shape == t2[0].shape


def test_train_step_on_the_scripted_module_matches_the_eager_module():
    """train.py:277-478 end to end on model_jit: script, move, delete the eager module, put the SCRIPTED module's parameters
    into the flat buffers, run TwoPassStep (the reference's two-pass order: forward_rough / forward_precise of model_jit) with
    the fused clip + AdamW.  Losses and updated parameters equal those of the same steps on an eager module (the operator's
    kernel is the same eager code, with direct gradient delivery into the flat buffer and all)."""
    import gc
    import bench
    from vkit_ocr_model_adaptive_scaling_amd.loss_function import (
        AdaptiveScalingRoughLossFunction, AdaptiveScalingRoughLossFunctionConifg, AdaptiveScalingPreciseLossFunction,
        AdaptiveScalingPreciseLossFunctionConifg)
    from vkit_ocr_model_adaptive_scaling_amd.training import FlatBuffers, FlatAdamW, TwoPassStep
    dev = torch.device('cuda', 0)
    rough, precise = bench.synthetic_batches(2, (256, 256), dev, 7)

    def run(scripted):
        m, _, _ = _model('upernext', torch.bfloat16)
        if scripted:
            jit = torch.jit.script(m).to(dev)
            del m
            gc.collect()
            m = jit
        else:
            m = m.to(dev)
        m.eval()  # no stochastic depth: the two runs must see the same function
        flat = FlatBuffers(m.named_parameters())
        step = TwoPassStep(m, AdaptiveScalingRoughLossFunction(AdaptiveScalingRoughLossFunctionConifg()),
                           AdaptiveScalingPreciseLossFunction(AdaptiveScalingPreciseLossFunctionConifg()),
                           FlatAdamW(None, flat=flat), None, merge_backbone=False)
        p0 = flat.flat_param.detach().clone()
        losses = [tuple(float(v) for v in step(rough, precise, lr=1e-4)) for _ in range(3)]
        torch.cuda.synchronize()
        assert rel_err(flat.flat_param, p0) > 1e-4   # the optimizer moved the scripted module's own parameters
        return losses, flat.flat_param.detach().clone()

    le, pe = run(False)
    lj, pj = run(True)
    assert le[0] == lj[0]                       # first step: identical forward, bit for bit
    assert all(abs(a - b) <= 1e-4 * abs(a) for x, y in zip(le, lj) for a, b in zip(x, y)), (le, lj)
    assert rel_err(pj, pe) < 1e-5               # (weight-gradient atomics reorder fp32 sums between runs)
