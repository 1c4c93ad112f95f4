#!/usr/bin/env python3
"""Run only the head-shaped implicit GEMMs (precise pass: 8x512x512x384 -> 792 channels, 3x3) a few times: used under
rocprofv3 --pmc to read SQ / TCC counters of the dominant kernels without the other ~1500 dispatches of a step."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from vkit_ocr_model_adaptive_scaling_amd import ops  # noqa: E402

g = torch.Generator(device='cuda').manual_seed(0)
x = torch.randn((8, 512, 512, 384), generator=g, device='cuda').bfloat16().requires_grad_(True)
w = (torch.randn((792, 384, 3, 3), generator=g, device='cuda') * 0.02).requires_grad_(True)
b = torch.zeros(792, device='cuda', requires_grad=True)
for _ in range(int(os.environ.get('PROBE_ITERS', '3'))):
    y = ops.Conv.apply(x, w, b, 1, 1)
    y.backward(torch.randn_like(y))
torch.cuda.synchronize()
print('done')
