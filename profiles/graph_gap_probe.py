#!/usr/bin/env python3
"""How much of the ~5 us between two dependent kernels of a stream a HIP graph removes: the same chain of 300 small
dependent launches (LayerNorm forward through the C ABI, ~60 us each: long enough that the host stays ahead) timed as plain stream launches with the
host far ahead, and replayed from a captured graph."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from vkit_ocr_model_adaptive_scaling_amd import ops  # noqa: E402

x = torch.randn((16, 128, 128, 384), device='cuda').bfloat16()
g = torch.ones(384, device='cuda')
b = torch.zeros(384, device='cuda')
N = 300


def chain(t):
    for _ in range(N):
        t, _ = ops.layernorm_fwd(t, g, b, 384, False)
    return t


def timed(fn, iters=5):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3 / N


print(f'stream launches: {timed(lambda: chain(x)):.2f} us per kernel (kernel + gap)')
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    chain(x)
torch.cuda.current_stream().wait_stream(side)
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    y = chain(x)
print(f'graph replay:    {timed(graph.replay):.2f} us per kernel')
