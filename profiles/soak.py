#!/usr/bin/env python3
"""Soak run: N merged train steps of config #3 on one fixed synthetic batch (the model should overfit it), printing the
losses, the allocator's peak / current memory and the step time every 50 steps - a leak, a drift into NaN or a slow-down
would show here.  usage: soak.py [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
from vkit_ocr_model_adaptive_scaling_amd.model import (AdaptiveScaling, AdaptiveScalingConfig, AdaptiveScalingSize,  # noqa: E402
                                                       AdaptiveScalingNeckHeadType)
from vkit_ocr_model_adaptive_scaling_amd.loss_function import (  # noqa: E402
    AdaptiveScalingRoughLossFunction, AdaptiveScalingRoughLossFunctionConifg,
    AdaptiveScalingPreciseLossFunction, AdaptiveScalingPreciseLossFunctionConifg)
from vkit_ocr_model_adaptive_scaling_amd.training import FlatBuffers, FlatAdamW, TwoPassStep, cosine_warm_restarts_lr  # noqa: E402
from vkit_ocr_model_adaptive_scaling_amd import ops  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device('cuda')
torch.manual_seed(1234)
model = AdaptiveScaling(AdaptiveScalingConfig(AdaptiveScalingSize.TINY, AdaptiveScalingNeckHeadType.UPERNEXT)).to(dev).train()
flat = FlatBuffers(model.named_parameters())
opt = FlatAdamW(None, lr=8e-4, betas=(0.9, 0.999), weight_decay=0.01, max_grad_norm=2.5, flat=flat)
step = TwoPassStep(model, AdaptiveScalingRoughLossFunction(AdaptiveScalingRoughLossFunctionConifg()),
                   AdaptiveScalingPreciseLossFunction(AdaptiveScalingPreciseLossFunctionConifg()), opt, merge_backbone=True)
rough, precise = bench.synthetic_batches(8, (1024, 1024), dev, 1337)
t0 = time.perf_counter()
for i in range(1, steps + 1):
    losses = step(rough, precise, lr=cosine_warm_restarts_lr(i / 1000.0, 8e-4, 8e-6, 10, 10))
    if i % 50 == 0 or i == 1:
        rl, pl = float(losses[0]), float(losses[1])
        dt = (time.perf_counter() - t0) / (50 if i > 1 else 1)
        t0 = time.perf_counter()
        print(f'step {i:4d}: L_rough {rl:.5f}  L_precise {pl:.5f}  {dt * 1e3:6.1f} ms/step  '
              f'mem now {torch.cuda.memory_allocated() / 2**30:.2f} GiB  peak {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB', flush=True)
        assert rl == rl and pl == pl, 'NaN'
ops.check_deferred(wait=True)
print('soak OK')
