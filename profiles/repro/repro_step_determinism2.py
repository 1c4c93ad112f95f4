"""Which intermediate gradient first differs between two runs of the same train step when another process shares the GPU
(development aid for tests/test_gpu_00_ddp_world2.py; run two copies at once)."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
import bench
from vkit_ocr_model_adaptive_scaling_amd.model import AdaptiveScaling, AdaptiveScalingConfig, AdaptiveScalingSize, AdaptiveScalingNeckHeadType
from vkit_ocr_model_adaptive_scaling_amd.loss_function import (AdaptiveScalingRoughLossFunction, AdaptiveScalingRoughLossFunctionConifg,
    AdaptiveScalingPreciseLossFunction, AdaptiveScalingPreciseLossFunctionConifg)
from vkit_ocr_model_adaptive_scaling_amd.training import FlatBuffers, TwoPassStep
dev = torch.device('cuda', 0)
torch.manual_seed(1000)
model = AdaptiveScaling(AdaptiveScalingConfig(AdaptiveScalingSize.TINY, AdaptiveScalingNeckHeadType.UPERNEXT), compute_dtype=torch.bfloat16).to(dev).eval()
with torch.no_grad():
    for n, p in model.named_parameters():
        if n.endswith('block_scale'): p.fill_(0.5)
flat = FlatBuffers(model.named_parameters())
rough, precise = bench.synthetic_batches(1, (256, 256), dev, 500)
rl = AdaptiveScalingRoughLossFunction(AdaptiveScalingRoughLossFunctionConifg())
pl = AdaptiveScalingPreciseLossFunction(AdaptiveScalingPreciseLossFunctionConifg())
class Keep:
    def step(self, lr=None): pass
    def zero_grad(self): pass
cap = {}
def tap(name, t):
    if isinstance(t, torch.Tensor) and t.requires_grad:
        cnt = sum(1 for k in cap if k.startswith(name + '#'))
        key = '%s#%d' % (name, cnt)
        t.register_hook(lambda g, key=key: cap.__setitem__(key, g.detach().float().clone()))
    return t
def wrap(obj, attr, name):
    f = getattr(obj, attr)
    def g(*a, **k):
        out = f(*a, **k)
        if isinstance(out, (list, tuple)):
            return type(out)(tap('%s[%d]' % (name, i), o) for i, o in enumerate(out))
        return tap(name, out)
    setattr(obj, attr, g)
wrap(model.backbone, 'forward_act', 'feats')
wrap(model.rough_neck, 'forward_act', 'rough_neck_out')
wrap(model.precise_neck, 'forward_act', 'precise_neck_out')
for bi, blk in enumerate(model.backbone.blocks):
    for li, layer in enumerate(blk.layers):
        if hasattr(layer, 'forward_act'):
            wrap(layer, 'forward_act', 'b%d.l%d' % (bi, li))
ref = None
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
for i in range(N):
    cap.clear()
    flat.zero_grad()
    TwoPassStep(model, rl, pl, Keep())(rough, precise)
    torch.cuda.synchronize()
    cur = dict(cap)
    if ref is None:
        ref = cur
        print('taps:', sorted(ref.keys()))
        continue
    bad = []
    for k in ref:
        e = float((cur[k].double() - ref[k].double()).norm() / max(float(ref[k].double().norm()), 1e-30))
        if e > 1e-6: bad.append((k, e))
    if bad:
        print('run', i, 'differing taps:', ['%s %.1e' % b for b in bad])
print('done', N)
