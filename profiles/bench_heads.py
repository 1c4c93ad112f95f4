#!/usr/bin/env python3
"""Times the fused head forward (conv3x3_slab_mfma_kernel<.,1>: 3x3 conv + LayerNorm + GELU + projection in the epilogue)
of the precise (4 heads) and rough (2 heads) pass at 8 x 512 x 512 x 384.  Run against timing-only ablation builds
(profiles/build_ablations.sh 256: no head tail) through VKAS_LIB_PATH to see what the epilogue costs."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from vkit_ocr_model_adaptive_scaling_amd import ops  # noqa: E402

g = torch.Generator(device='cuda').manual_seed(0)
x = torch.randn((8, 512, 512, 384), generator=g, device='cuda').bfloat16()
for name, chans, ocs in (('precise', (192, 193, 194, 194), (1, 2, 4, 4)), ('rough', (192, 192), (1, 1))):
    params = []
    for c, oc in zip(chans, ocs):
        params += [torch.randn((c, 384, 3, 3), generator=g, device='cuda') * 0.02, torch.zeros(c, device='cuda'),
                   torch.ones(c, device='cuda'), torch.zeros(c, device='cuda'),
                   torch.randn((oc, c), generator=g, device='cuda') * 0.05, torch.zeros(oc, device='cuda')]
    for keep in (True, False):
        for _ in range(2):
            ops.HeadsFused.apply(x, keep, *params)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(3):
            ops.HeadsFused.apply(x, keep, *params)
        e.record()
        torch.cuda.synchronize()
        print(f'{name:8s} keep={keep}: {s.elapsed_time(e) / 3:.3f} ms')
