#!/usr/bin/env python3
"""Which Python lines of this package still launch torch-native kernels / device copies inside one train step (the ~65 ATen
launches and ~80 rocclr copies of profiles/r02_v6_kernel_stats.csv): torch.profiler with stacks over 2 steps at a small size,
grouped by (op, innermost frame inside the package)."""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from torch.profiler import profile, ProfilerActivity  # noqa: E402
import bench  # noqa: E402
from vkit_ocr_model_adaptive_scaling_amd.model import (AdaptiveScaling, AdaptiveScalingConfig, AdaptiveScalingSize,  # noqa: E402
                                                       AdaptiveScalingNeckHeadType)
from vkit_ocr_model_adaptive_scaling_amd.loss_function import (AdaptiveScalingRoughLossFunction,  # noqa: E402
                                                               AdaptiveScalingRoughLossFunctionConifg,
                                                               AdaptiveScalingPreciseLossFunction,
                                                               AdaptiveScalingPreciseLossFunctionConifg)
from vkit_ocr_model_adaptive_scaling_amd.training import FlatBuffers, FlatAdamW, TwoPassStep  # noqa: E402


def main():
    size = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    model = AdaptiveScaling(AdaptiveScalingConfig(AdaptiveScalingSize.TINY, AdaptiveScalingNeckHeadType.UPERNEXT)).to(dev).train()
    flat = FlatBuffers(model.named_parameters())
    opt = FlatAdamW(None, flat=flat)
    step = TwoPassStep(model, AdaptiveScalingRoughLossFunction(AdaptiveScalingRoughLossFunctionConifg()),
                       AdaptiveScalingPreciseLossFunction(AdaptiveScalingPreciseLossFunctionConifg()), opt, None,
                       merge_backbone=True)
    rough, precise = bench.synthetic_batches(2, (size, size), dev, 1)
    for _ in range(3):
        step(rough, precise, lr=1e-4)
    torch.cuda.synchronize()
    n = 2
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        for _ in range(n):
            step(rough, precise, lr=1e-4)
        torch.cuda.synchronize()
    pkg = 'vkit_ocr_model_adaptive_scaling_amd'
    groups = collections.Counter()
    for ev in prof.events():
        if not ev.name.startswith('aten::') or ev.cpu_parent is not None and ev.cpu_parent.name.startswith('aten::'):
            continue
        # only ops that reach the device
        if not (ev.device_time_total > 0 or any(k.name for k in ev.kernels)):
            continue
        frame = next((f for f in (ev.stack or []) if pkg in f or 'bench.py' in f), '?')
        groups[(ev.name, frame.split(pkg)[-1])] += 1
    for (name, frame), c in groups.most_common(60):
        print('%6.1f / step  %-28s %s' % (c / n, name, frame))
    # runtime copies / memsets and who called them (outermost enclosing CPU op)
    rt = collections.Counter()
    for ev in prof.events():
        if not any(k in ev.name for k in ('Memcpy', 'Memset', 'memcpy', 'memset')):
            continue
        par, top = ev.cpu_parent, None
        while par is not None:
            top = par
            par = par.cpu_parent
        rt[(ev.name[:40], top.name[:60] if top is not None else '-')] += 1
    for (name, parent), c in rt.most_common(30):
        print('%6.1f / step  %-40s under %s' % (c / n, name, parent))


if __name__ == '__main__':
    main()
