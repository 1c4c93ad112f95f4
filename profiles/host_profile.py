#!/usr/bin/env python3
"""Host-side cost of one train step: cProfile of bench-like steps at a small size (the GPU work is then negligible, so the
wall time per step is the Python / ctypes / allocator / launch cost of the ~1000 launches)."""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from vkit_ocr_model_adaptive_scaling_amd.model import (AdaptiveScaling, AdaptiveScalingConfig, AdaptiveScalingSize,  # noqa: E402
                                                       AdaptiveScalingNeckHeadType)
from vkit_ocr_model_adaptive_scaling_amd.loss_function import (AdaptiveScalingRoughLossFunction,  # noqa: E402
                                                               AdaptiveScalingRoughLossFunctionConifg,
                                                               AdaptiveScalingPreciseLossFunction,
                                                               AdaptiveScalingPreciseLossFunctionConifg)
from vkit_ocr_model_adaptive_scaling_amd.training import FlatBuffers, FlatAdamW, TwoPassStep  # noqa: E402


def main():
    size = int(sys.argv[1]) if len(sys.argv) > 1 else 256
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
    n = 10
    t0 = time.perf_counter()
    for _ in range(n):
        step(rough, precise, lr=1e-4)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f'host enqueue {1e3 * (t1 - t0) / n:.1f} ms/step, wall {1e3 * (t2 - t0) / n:.1f} ms/step at {size}x{size}')
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(5):
        step(rough, precise, lr=1e-4)
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats('tottime').print_stats(28)


if __name__ == '__main__':
    main()
