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
ref = None; worst = 0.0
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
for i in range(N):
    flat.zero_grad()
    out = TwoPassStep(model, rl, pl, Keep())(rough, precise)
    torch.cuda.synchronize()
    g = flat.flat_grad.clone()
    losses = [float(v) for v in out] if isinstance(out, (tuple, list)) else out
    if ref is None: ref = g; ref_losses = losses
    else:
        e = float((g.double() - ref.double()).norm() / ref.double().norm())
        worst = max(worst, e)
        if e > 1e-5:
            print('run', i, 'differs', e, 'losses', losses, 'ref', ref_losses)
            rows = []
            for n in flat.names:
                st, sz = flat.offsets[n]
                d = float((g[st:st + sz].double() - ref[st:st + sz].double()).norm()); r = float(ref[st:st + sz].double().norm())
                rows.append((d / max(r, 1e-30), n))
            rows.sort(reverse=True)
            nz = sum(1 for e2, _ in rows if e2 > 1e-5)
            print('   params differing:', nz, 'of', len(rows), '; top:', ['%s %.1e' % (n, e2) for e2, n in rows[:8]])
            print('   smallest differing:', ['%s %.1e' % (n, e2) for e2, n in rows[max(nz - 4, 0):nz]])
print('single process,', N, 'runs: worst rel diff', worst)
