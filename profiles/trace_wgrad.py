#!/usr/bin/env python3
"""Phase times of conv3x3_wgrad_slab_kernel (-DVKAS_TRACE build, profiles/build_trace.sh): one head-shaped weight-gradient
launch (8 x 512 x 512 x 384 -> N channels), median s_memtime cycles per 64-pixel chunk and wave group.
usage: VKAS_LIB_PATH=build_variants/libvkas_trace.so trace_wgrad.py [N]"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from vkit_ocr_model_adaptive_scaling_amd import ops  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 384
g = torch.Generator(device='cuda').manual_seed(0)
x = torch.randn((8, 512, 512, 384), generator=g, device='cuda').bfloat16()
dy = torch.randn((8, 512, 512, N), generator=g, device='cuda').bfloat16()
geom = ops._geom(8, 512, 512, 512, 512, 384, 384, 3, 3, 1, 1)
for _ in range(2):
    ops.conv_wgrad(x, geom, dy, N, with_bias=True)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
ops.conv_wgrad(x, geom, dy, N, with_bias=True)
e.record()
torch.cuda.synchronize()
raw = ctypes.CDLL(os.environ['VKAS_LIB_PATH'])
buf = np.zeros(65536 * 8, dtype=np.uint64)
assert raw.vkas_trace_read(ctypes.c_void_p(buf.ctypes.data), ctypes.c_size_t(buf.nbytes)) == 0
t = buf.reshape(-1, 8).astype(np.float64)
print(f'N = {N}: launch {s.elapsed_time(e):.3f} ms')
names = ('fragment reads (+ lgkmcnt)', 'DMA issue', 'retire in READ (group 1) + lgkmcnt', 'barrier after READ', 'MFMA phase',
         'retire after MFMA (group 0)', 'barrier after MFMA')
for grp in (0, 1):
    r = t[grp::2]
    r = r[r[:, 7] > 0]
    per = r[:, :7] / r[:, 7:8]
    print(f' wave group {grp}: {len(r)} workgroups, {np.median(r[:, 7]):.0f} chunks each; median cycles per chunk')
    for i, n in enumerate(names):
        print(f'    {n:38s} {np.median(per[:, i]):7.0f}')
    print(f'    {"sum":38s} {np.median(per.sum(1)):7.0f}')
