"""Which autograd node of the train step first produces a different gradient between two runs on the same data when another
process shares the GPU (development aid for tests/test_gpu_00_ddp_world2.py; run two copies at once).  Every ops.* Function is
wrapped: the gradient arriving at each of its outputs and leaving for each of its inputs is kept, in the order backward
produces them, and so are the forward outputs.  A node whose output gradients agree with the first run while an input gradient
does not is where the difference is made."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
import bench
from vkit_ocr_model_adaptive_scaling_amd import ops
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
ONLY = os.environ.get('TAP_ONLY', 'HeadsFused,HeadsAtPoints,Resize,ResizeAdd,ToNchw,RoughLoss,PreciseLoss,Softplus,SplitBatch,ResizeCat,Cat,Conv,LayerNorm,MultiLayerNorm,AdaptiveAvgPools,AdaptiveAvgPool').split(',')
fwd, bwd, order, count = {}, {}, [], {}
def keep(store, key, t):
    store[key] = t.detach().float().clone()
    if store is bwd: order.append(key)
def wrap(cls):
    name = cls.__name__
    inner = cls.apply
    def apply(*a):
        k = count.get(name, 0); count[name] = k + 1
        tag = '%s.%d' % (name, k)
        for i, t in enumerate(a):
            if isinstance(t, torch.Tensor) and t.requires_grad and not t.is_leaf and t.numel() > 64:
                t.register_hook(lambda g, key='%s.in%d' % (tag, i): keep(bwd, key, g))
        out = inner(*a)
        outs = out if isinstance(out, (tuple, list)) else (out,)
        for i, t in enumerate(outs):
            if isinstance(t, torch.Tensor):
                keep(fwd, '%s.out%d' % (tag, i), t)
                if t.requires_grad:
                    t.register_hook(lambda g, key='%s.gout%d' % (tag, i): keep(bwd, key, g))
        return out
    cls.apply = staticmethod(apply)
for n in ONLY:
    if hasattr(ops, n): wrap(getattr(ops, n))
# internals of HeadsFused.backward: its saved tensors, and the operands / results of every GEMM it launches
inside = [None]
_bw = ops.HeadsFused.backward
def hf_backward(ctx, *dprojs):
    k = count.get('HFb', 0); count['HFb'] = k + 1
    inside[0] = 'HFb%d' % k
    for j, t in enumerate(ctx.saved_tensors[:4]):
        keep(bwd, '%s.saved%d' % (inside[0], j), t)
    for j, t in enumerate(dprojs):
        if t is not None: keep(bwd, '%s.dproj%d' % (inside[0], j), t)
    try:
        return _bw(ctx, *dprojs)
    finally:
        inside[0] = None
ops.HeadsFused.backward = staticmethod(hf_backward)
def wrap_fn(name, in_idx, out_of):
    inner = getattr(ops, name)
    def f(*a, **k):
        if inside[0] is None: return inner(*a, **k)
        c = count.get(inside[0] + name, 0); count[inside[0] + name] = c + 1
        tag = '%s.%s%d' % (inside[0], name, c)
        for i in in_idx:
            keep(bwd, '%s.arg%d' % (tag, i), a[i])
        out = inner(*a, **k)
        o = out_of(a, k, out)
        if o is not None: keep(bwd, tag + '.result', o)
        return out
    setattr(ops, name, f)
wrap_fn('conv_gemm', (0, 2), lambda a, k, out: a[4])             # x, Bw -> out
wrap_fn('conv_wgrad', (0, 2), lambda a, k, out: None)            # x, dy
ref = None
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
def rel(a, b):
    return float((a.double() - b.double()).norm() / max(float(b.double().norm()), 1e-30))
for i in range(N):
    fwd.clear(); bwd.clear(); order.clear(); count.clear()
    flat.zero_grad()
    TwoPassStep(model, rl, pl, Keep())(rough, precise)
    torch.cuda.synchronize()
    cur = (dict(fwd), dict(bwd), list(order))
    if ref is None:
        ref = cur
        print('forward taps', len(ref[0]), 'backward taps', len(ref[1]))
        continue
    badf = [(k, rel(cur[0][k], ref[0][k])) for k in ref[0] if not torch.equal(cur[0][k], ref[0][k])]
    badb = [(k, rel(cur[1][k], ref[1][k])) for k in ref[2] if not torch.equal(cur[1][k], ref[1][k])]
    big = [b for b in badb if b[1] > 1e-6]
    if badf or big:
        print('run', i, 'forward differing (first 6):', ['%s %.1e' % b for b in badf[:6]])
        print('   backward, in the order produced: first 10 differing (any size):', ['%s %.1e' % b for b in badb[:10]])
        print('   first 6 above 1e-6:', ['%s %.1e' % b for b in big[:6]])
        d = big[0][0] if big else None
        if d is not None:
            a, b = cur[1][d], ref[1][d]
            diff = (a - b).abs()
            nz = (diff > 0).nonzero()
            print('   ', d, 'shape', tuple(a.shape), 'elements differing', int((diff > 0).sum()), 'max abs', float(diff.max()),
                  'ref max abs', float(b.abs().max()), 'first idx', nz[0].tolist(), 'last idx', nz[-1].tolist())
            if a.dim() == 4:
                rows = (diff.amax(dim=3) > 0)
                ys = rows.any(dim=2).nonzero()[:, 1]; xs = rows.any(dim=1).nonzero()[:, 1]
                ch = (diff.amax(dim=(0, 1, 2)) > 0).nonzero().flatten()
                print('    y range', int(ys.min()), int(ys.max()), 'x range', int(xs.min()), int(xs.max()), 'channels', ch[:8].tolist(), '...', ch[-4:].tolist(), 'n', ch.numel())
print('done', N)
